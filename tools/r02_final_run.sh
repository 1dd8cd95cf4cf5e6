# Round-2 canonical run (GPU box): full GPU test suite, smoke, the unprofiled bench lines that go into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fin
mkdir -p $O
cd $R
timeout 240 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err
timeout 120 python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/fin/bench_c3_default.json"))
b = d.get("boundary", {})
print("default", d["value"], d["ms_per_step"], "sustained", b.get("sustained_sessions"), "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
d = json.load(open("gpurun_out/fin/bench_c3_serial.json"))
print("serial", d["value"], d["config"]["stages_ms_per_step"], {k: (v["frac"], v["avg_kernel_us"]) for k, v in d["roofline"]["families"].items()})
PY
