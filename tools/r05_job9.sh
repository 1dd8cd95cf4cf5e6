# Round 5, GPU job 9: f32 full-depth bit-exact tests, PMC traffic of the tile family, the default bench line
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5i}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_full_depth_gpu.py -x -q -m gpu -s > $O/tests_full_depth.log 2>&1; grep -E "f32|bf16|passed|failed|Error|assert" $O/tests_full_depth.log | tail -30
RED="--serial --tts-merge 2 --steps 2 --warmup 0 --no-cpu-baseline --no-boundary --decode-tokens 4 --reply-frames 8"
timeout 100 python bench.py $RED > $O/bench_reduced.json 2> $O/bench_reduced.err
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py $RED > /dev/null 2> $O/pmc_fetch.err
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py $RED > /dev/null 2> $O/pmc_write.err
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/bench_reduced.json $O/pmc_traffic.json "python bench.py $RED" > $O/pmc_traffic.log 2>&1
tail -8 $O/pmc_traffic.log
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
T0=$SECONDS; timeout 1000 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "default bench took $((SECONDS - T0)) s"
python - <<PY
import json
for n in ("bench_c3_default",):
    try:
        d = json.load(open("$O/" + n + ".json"))
    except Exception as e:
        print(n, "ERR", e); continue
    b = d.get("boundary") or {}
    print(n, "value", d["value"], d.get("value_source"), "xrt", d["xrt"], "ms/step", d["ms_per_step"], "p50", d.get("p50_ttfa_ms"), "sustained", b.get("sustained_sessions"),
          "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), "roof", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("kernel", "")[:30], "traffic", (d.get("roofline") or {}).get("traffic"),
          "cpu", (d.get("cpu_baseline") or {}).get("value"), "spread", b.get("spread_at_sustained"))
    for r in b.get("probes", []) + b.get("repeats", []):
        print("   probe", r["sessions"], r.get("sustained"), "p50", r["p50_ttfa_ms"], "p95", r["p95_ttfa_ms"], "done", r["turns_completed"], "/", r["turns_arrived"], "lag", r["max_playback_lag_ms"], "stt", r.get("p50_stt_ms"), r.get("p95_stt_ms"))
PY
