# GPU box: the check of HEAD before a round ends - full GPU suite, smoke(), the default bench line.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3q
mkdir -p $O
cd $R
timeout 500 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 700 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -2 $O/bench.err; cat $O/bench.time
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3q/bench.json"))
b = d.get("boundary", {})
print("value", d["value"], d.get("value_source"), "xrt", d["xrt"], "ms/step", d["ms_per_step"], "p50", d.get("p50_ttfa_ms"), "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), b.get("error"))
for r in b.get("probes", []):
    print({k: r.get(k) for k in ("sessions", "sustained", "turns_completed", "turns_arrived", "min_blocks_per_turn", "p50_stt_ms", "p50_ttfa_ms", "p95_ttfa_ms")})
print("roof", {k: d["roofline"].get(k) for k in ("frac", "avg_kernel_us", "traffic", "frac_predictor_weights_once_per_frame")}, "cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "cores", "ttfa_ms")})
PY
