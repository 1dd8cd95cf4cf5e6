# Round-6 job 8: codec k = 7 convolutions against their floors; TDT step variants test; full C5 line with the 64-window batch cap
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6h}
mkdir -p $O
cd $R
timeout 300 python tools/bench_conv.py 512 2>&1 | grep -v amdgpu.ids > $O/bench_conv.txt; cat $O/bench_conv.txt
timeout 600 python -m pytest tests/test_parakeet_gpu.py -m gpu -x -q -k variants -s > $O/tests_tdt_variants.log 2>&1; tail -4 $O/tests_tdt_variants.log
T0=$SECONDS; timeout 1500 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 bench took $((SECONDS - T0)) s rc $?"
python - <<PY
import json
d = json.load(open("$O/bench_c5.json"))
b = d.get("boundary") or {}
r = d.get("roofline") or {}
print("c5 value", d["value"], d.get("value_source"), "xrt", d["xrt"], "ms/step", d["ms_per_step"], "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), "spread", b.get("spread_at_sustained"), "err", b.get("error"))
for k, f in (r.get("families") or {}).items():
    print("  fam", k, f["kernel_ms_per_step"], "ms", f["launches_per_step"], "launches", f["frac"])
for r_ in b.get("probes", []) + b.get("repeats", []):
    print("   probe", r_["sessions"], r_.get("sustained"), "p50", r_["p50_ttfa_ms"], "p95", r_["p95_ttfa_ms"], "done", r_["turns_completed"], "/", r_["turns_arrived"], "partials", r_.get("partials_answered_frac"), r_.get("p50_partial_ms"), r_.get("p95_partial_ms"))
PY
