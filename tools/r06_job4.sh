# Round-6 job 4: default C3 line (registry-built units, live PMC traffic, widened parity) and the C5 line under the >= 90 % partials criterion
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6d}
mkdir -p $O
cd $R
T0=$SECONDS; timeout 1500 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "default bench took $((SECONDS - T0)) s rc $?"
T0=$SECONDS; timeout 1500 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 bench took $((SECONDS - T0)) s rc $?"
python - <<PY
import json
for n in ("bench_c3_default", "bench_c5"):
    try:
        d = json.load(open("$O/" + n + ".json"))
    except Exception as e:
        print(n, "ERR", e); continue
    b = d.get("boundary") or {}
    r = d.get("roofline") or {}
    print(n, "value", d["value"], d.get("value_source"), "xrt", d["xrt"], "ms/step", d["ms_per_step"], "p50", d.get("p50_ttfa_ms"), "sustained", b.get("sustained_sessions"),
          "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), "roof", r.get("frac"), r.get("kernel", "")[:30], "traffic", r.get("traffic"), r.get("traffic_source"), r.get("traffic_live_error"),
          "cpu", (d.get("cpu_baseline") or {}).get("value"), "spread", b.get("spread_at_sustained"), "err", b.get("error"))
    print("  parity", {k: (v if k != "tts" else {kk: vv for kk, vv in v.items() if kk != "what"}) for k, v in (d.get("parity") or {}).items() if k != "what"})
    for k, f in (r.get("families") or {}).items():
        print("  fam", k, f["kernel_ms_per_step"], "ms", f["launches_per_step"], "launches", f["frac"])
    for r_ in b.get("probes", []) + b.get("repeats", []):
        print("   probe", r_["sessions"], r_.get("sustained"), "p50", r_["p50_ttfa_ms"], "p95", r_["p95_ttfa_ms"], "done", r_["turns_completed"], "/", r_["turns_arrived"], "lag", r_["max_playback_lag_ms"], "stt", r_.get("p50_stt_ms"), r_.get("p95_stt_ms"),
              "partials", r_.get("partials_answered"), "/", r_.get("partials_sent"), r_.get("p50_partial_ms"), r_.get("p95_partial_ms"), "units", (r_.get("pipeline_units") or {}).get("build_s"))
PY
