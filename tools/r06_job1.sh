# Round-6 job 1 (GPU box): changed GPU tests, then the new bench paths on the tiny workload (registry-built units, live PMC traffic, widened parity)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6a}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_parakeet_progressive_gpu.py tests/test_pipeline_gpu.py tests/test_server_gpu.py -m gpu -x -q > $O/tests_changed.log 2>&1; tail -5 $O/tests_changed.log
timeout 400 python bench.py --workload tiny --steps 2 --warmup 1 --boundary-seconds 6 --max-probes 2 --confirm-probes 1 > $O/bench_tiny.json 2> $O/bench_tiny.err; echo "tiny rc $?"; tail -5 $O/bench_tiny.err
python - <<PY
import json
try:
    d = json.load(open("$O/bench_tiny.json"))
    r = d.get("roofline") or {}
    print("tiny value", d["value"], "roof", r.get("frac"), "traffic", r.get("traffic"), r.get("traffic_source"), r.get("traffic_live_error"))
    print("parity", {k: v for k, v in (d.get("parity") or {}).items() if k != "what"})
    b = d.get("boundary") or {}
    print("boundary err", b.get("error"), "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), (b.get("single_session") or {}).get("pipeline_units"))
except Exception as e:
    print("ERR", e)
PY
