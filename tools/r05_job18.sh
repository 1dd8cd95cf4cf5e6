# Round 5, GPU job 18: f3 - cached (incremental) Parakeet log-mel: bit-identity tests, the a6 tests with incremental windows, C5 open loop with it
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5r}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_parakeet_progressive_gpu.py tests/test_logmel_gpu.py tests/test_parakeet_gpu.py tests/test_abi.py -x -q -m gpu -s > $O/tests_f3.log 2>&1; grep -v amdgpu $O/tests_f3.log | tail -25
timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --probe-only --probe-sessions 988 --boundary-seconds 20 > $O/bench_c5.json 2> $O/bench_c5.err
python - <<PY
import json
d = json.load(open("$O/bench_c5.json")); b = d.get("boundary") or {}
for r in b.get("probes", []):
    print("   probe", r["sessions"], "p50", r["p50_ttfa_ms"], "p95", r["p95_ttfa_ms"], "done", r["turns_completed"], "/", r["turns_arrived"], "lag", r["max_playback_lag_ms"], "stt", r.get("p50_stt_ms"), r.get("p95_stt_ms"), r["pool"])
PY
