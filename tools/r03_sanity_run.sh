set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3q
mkdir -p $O
cd $R
timeout 500 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --max-probes 3 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3q/bench.json"))
b = d.get("boundary", {})
print("value", d["value"], d.get("value_source"), "xrt", d["xrt"], "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), b.get("error"))
for r in b.get("probes", []):
    print({k: r.get(k) for k in ("sessions", "sustained", "turns_completed", "turns_arrived", "min_blocks_per_turn", "p50_ttfa_ms", "p95_ttfa_ms")})
PY
