# Round-5 canonical run (GPU box): GPU suite, smoke, serial profile (kernel families), default bench line
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5fin}
mkdir -p $O
cd $R
if [ "${2:-all}" != "bench" ]; then timeout 700 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log; fi
if [ "${2:-all}" != "bench" ]; then timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2; fi
SER="--serial --tts-merge 8 --steps 8 --warmup 8 --no-cpu-baseline --no-boundary --parity-sessions 0"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o c3s -- python bench.py $SER > $O/bench_c3_serial_prof.json 2> $O/bench_c3_serial_prof.err
timeout 300 python bench.py $SER --breakdown > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
cp $(find $O/prof_serial -name "*kernel_stats.csv" | head -1) $O/c3_serial_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_serial
python tools/prof_families.py $O/c3_serial_kernel_stats.csv 24 $O/bench_c3_serial.json
if [ "${2:-all}" != "prof" ]; then
T0=$SECONDS; timeout 1000 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "default bench took $((SECONDS - T0)) s"
fi
python - <<PY
import json
for n in ("bench_c3_default", "bench_c3_serial"):
    try:
        d = json.load(open("$O/" + n + ".json"))
    except Exception as e:
        print(n, "ERR", e); continue
    b = d.get("boundary") or {}
    print(n, "value", d["value"], d.get("value_source"), "xrt", d["xrt"], "ms/step", d["ms_per_step"], "p50", d.get("p50_ttfa_ms"), "sustained", b.get("sustained_sessions"),
          "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), "roof", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("kernel", "")[:30], "traffic", (d.get("roofline") or {}).get("traffic"),
          "cpu", (d.get("cpu_baseline") or {}).get("value"), "spread", b.get("spread_at_sustained"), "parity", {k: v for k, v in (d.get("parity") or {}).items() if k != "what"})
    for r in b.get("probes", []) + b.get("repeats", []):
        print("   probe", r["sessions"], r.get("sustained"), "p50", r["p50_ttfa_ms"], "p95", r["p95_ttfa_ms"], "done", r["turns_completed"], "/", r["turns_arrived"], "lag", r["max_playback_lag_ms"], "stt", r.get("p50_stt_ms"), r.get("p95_stt_ms"))
PY
