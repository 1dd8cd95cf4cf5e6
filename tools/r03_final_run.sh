# Round-3 canonical run (GPU box): full GPU suite, smoke, bench lines and profiles that go into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3fin
mkdir -p $O
cd $R
timeout 500 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
# 1. serial schedule under rocprofv3 kernel trace: the CSV the roofline fractions are recomputed from; then the unprofiled serial line
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o c3s -- python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_c3_serial_prof.json 2> $O/bench_c3_serial_prof.err
timeout 200 python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-boundary --breakdown > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
# 2. PMC traffic on a reduced step (separate passes)
RED="--serial --steps 1 --warmup 0 --no-cpu-baseline --no-boundary --decode-tokens 4 --reply-frames 8"
timeout 100 python bench.py $RED > $O/bench_reduced.json 2> $O/bench_reduced.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py $RED > /dev/null 2> $O/pmc_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py $RED > /dev/null 2> $O/pmc_write.err
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/bench_reduced.json $O/pmc_traffic.json "python bench.py $RED" > $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json 2>/dev/null
# 3. the default line (what the driver runs), other workloads
timeout 600 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err
timeout 150 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --max-probes 2 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 150 python bench.py --workload c1 --steps 5 --warmup 2 --no-cpu-baseline --max-probes 2 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 150 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_c5.json 2> $O/bench_c5.err
# keep only the small summaries
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
python tools/prof_families.py $(find $O/prof_serial -name "*kernel_stats.csv" | head -1) 5 $O/bench_c3_serial.json
tail -3 $O/pmc_traffic.log
python - <<'PY'
import json
for n in ("bench_c3_default", "bench_c3_serial", "bench_c2", "bench_c1", "bench_c5"):
    try:
        d = json.load(open(f"gpurun_out/r3fin/{n}.json"))
    except Exception as e:
        print(n, "ERR", e); continue
    b = d.get("boundary") or {}
    print(n, "value", d["value"], d.get("value_source"), "xrt", d["xrt"], "ms/step", d["ms_per_step"], "p50", d.get("p50_ttfa_ms"), "sustained", b.get("sustained_sessions"),
          "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), "roof", (d.get("roofline") or {}).get("frac"), "traffic", (d.get("roofline") or {}).get("traffic"),
          "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
