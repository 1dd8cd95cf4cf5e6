# Round-4 canonical run (GPU box): GPU suite, smoke, bench lines and the profiles that go into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r4fin}
mkdir -p $O
cd $R
if [ "${2:-all}" != "bench" ]; then
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
fi
# 1. the default line (what the driver runs)
if [ "${2:-all}" != "prof" ]; then
T0=$SECONDS; timeout 900 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "default bench took $((SECONDS - T0)) s"
fi
if [ "${2:-all}" != "bench" ]; then
# 2. serial schedule (one stream, 8 turn-batches -> ONE 512-row TTS run) under rocprofv3 kernel trace: the CSV the roofline fractions are recomputed from; then unprofiled
SER="--serial --tts-merge 8 --steps 8 --warmup 8 --no-cpu-baseline --no-boundary"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o c3s -- python bench.py $SER > $O/bench_c3_serial_prof.json 2> $O/bench_c3_serial_prof.err
timeout 300 python bench.py $SER --breakdown > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
# 3. PMC traffic on a reduced step (separate passes)
RED="--serial --tts-merge 2 --steps 2 --warmup 0 --no-cpu-baseline --no-boundary --decode-tokens 4 --reply-frames 8"
timeout 100 python bench.py $RED > $O/bench_reduced.json 2> $O/bench_reduced.err
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py $RED > /dev/null 2> $O/pmc_fetch.err
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py $RED > /dev/null 2> $O/pmc_write.err
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/bench_reduced.json $O/pmc_traffic.json "python bench.py $RED" > $O/pmc_traffic.log 2>&1
# 4. other workloads
timeout 200 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --max-probes 2 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 200 python bench.py --workload c1 --steps 5 --warmup 2 --no-cpu-baseline --max-probes 2 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 300 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --probe-only --probe-sessions 300 --boundary-seconds 15 > $O/bench_c5.json 2> $O/bench_c5.err
cp $(find $O/prof_serial -name "*kernel_stats.csv" | head -1) $O/c3_serial_kernel_stats.csv 2>/dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
python tools/prof_families.py $O/c3_serial_kernel_stats.csv 24 $O/bench_c3_serial.json    # 8 warm-up + 8 timed + 8 (one merged roofline-sample step) turn-batches
tail -3 $O/pmc_traffic.log
fi
python - <<PY
import json
for n in ("bench_c3_default", "bench_c3_serial", "bench_c2", "bench_c1", "bench_c5"):
    try:
        d = json.load(open("$O/" + n + ".json"))
    except Exception as e:
        print(n, "ERR", e); continue
    b = d.get("boundary") or {}
    print(n, "value", d["value"], d.get("value_source"), "xrt", d["xrt"], "ms/step", d["ms_per_step"], "p50", d.get("p50_ttfa_ms"), "sustained", b.get("sustained_sessions"),
          "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), "roof", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("kernel", "")[:30], "traffic", (d.get("roofline") or {}).get("traffic"),
          "cpu", (d.get("cpu_baseline") or {}).get("value"))
    for r in b.get("probes", []) + b.get("repeats", []):
        print("   probe", r["sessions"], r.get("sustained"), "p50", r["p50_ttfa_ms"], "p95", r["p95_ttfa_ms"], "done", r["turns_completed"], "/", r["turns_arrived"], "lag", r["max_playback_lag_ms"])
PY
