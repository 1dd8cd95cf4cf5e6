# Round-6 canonical run (GPU box): GPU suite, smoke, serial profile of the default schedule's launches (rocprofv3), default bench line, C5, C2, C1
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6fin}
mkdir -p $O
cd $R
if [ "${2:-all}" = "all" ] || [ "${2:-all}" = "tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
  timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
fi
if [ "${2:-all}" = "all" ] || [ "${2:-all}" = "prof" ]; then
  SER="--serial --tts-merge 4 --steps 8 --warmup 4 --no-cpu-baseline --no-boundary --no-pmc --parity-sessions 0"
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o c3s -- python bench.py $SER > $O/bench_c3_serial_prof.json 2> $O/bench_c3_serial_prof.err
  cp $(find $O/prof_serial -name "*kernel_stats.csv" | head -1) $O/c3_serial_kernel_stats.csv 2>/dev/null
  rm -rf $O/prof_serial
  timeout 400 python bench.py $SER --breakdown > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
  python tools/prof_families.py $O/c3_serial_kernel_stats.csv 16 $O/bench_c3_serial.json
fi
if [ "${2:-all}" = "all" ] || [ "${2:-all}" = "bench" ]; then
  T0=$SECONDS; timeout 1500 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "default bench took $((SECONDS - T0)) s rc $?"
fi
if [ "${2:-all}" = "all" ] || [ "${2:-all}" = "others" ]; then
  T0=$SECONDS; timeout 1500 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 bench took $((SECONDS - T0)) s rc $?"
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --max-probes 1 --confirm-probes 0 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc $?"
  timeout 600 python bench.py --workload c1 --no-cpu-baseline --max-probes 1 --confirm-probes 0 > $O/bench_c1.json 2> $O/bench_c1.err; echo "c1 rc $?"
fi
python - <<PY
import json, os
for n in ("bench_c3_default", "bench_c3_serial", "bench_c5", "bench_c2", "bench_c1"):
    f = "$O/" + n + ".json"
    if not os.path.exists(f):
        continue
    try:
        d = json.load(open(f))
    except Exception as e:
        print(n, "ERR", e); continue
    b = d.get("boundary") or {}
    r = d.get("roofline") or {}
    print(n, "value", d["value"], d.get("value_source"), "xrt", d["xrt"], "ms/step", d["ms_per_step"], "p50", d.get("p50_ttfa_ms"), "single", (b.get("single_session") or {}).get("p50_ttfa_ms"),
          "roof", r.get("frac"), (r.get("kernel") or "")[:24], "traffic", r.get("traffic"), (r.get("traffic_source") or "")[:12], r.get("traffic_live_error"), "cpu", (d.get("cpu_baseline") or {}).get("value"),
          "spread", b.get("spread_at_sustained"), "err", b.get("error"), "wall", d.get("wall_s"))
    p = d.get("parity") or {}
    print("  parity", {k: v for k, v in p.items() if k not in ("what", "tts")}, "tts", {k: v for k, v in (p.get("tts") or {}).items() if k != "what"})
    for k, f_ in (r.get("families") or {}).items():
        print("  fam", k, f_["kernel_ms_per_step"], "ms", f_["launches_per_step"], "launches", f_["frac"])
    for r_ in b.get("probes", []) + b.get("repeats", []):
        print("   probe", r_["sessions"], r_.get("sustained"), "p50", r_["p50_ttfa_ms"], "p95", r_["p95_ttfa_ms"], "done", r_["turns_completed"], "/", r_["turns_arrived"], "lag", r_["max_playback_lag_ms"],
              "partials", r_.get("partials_answered_frac"), r_.get("p50_partial_ms"), r_.get("p95_partial_ms"))
PY
