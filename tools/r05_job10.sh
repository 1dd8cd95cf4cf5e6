# Round 5, GPU job 10: a6 on the GPU, PMC traffic (tool now counts gemm_tile5), the other workloads' lines (C2, C1, C5)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5j}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_parakeet_progressive_gpu.py -x -q -m gpu -s > $O/tests_progressive.log 2>&1; tail -12 $O/tests_progressive.log
RED="--serial --tts-merge 2 --steps 2 --warmup 0 --no-cpu-baseline --no-boundary --decode-tokens 4 --reply-frames 8"
timeout 100 python bench.py $RED > $O/bench_reduced.json 2> $O/bench_reduced.err
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py $RED > /dev/null 2> $O/pmc_fetch.err
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py $RED > /dev/null 2> $O/pmc_write.err
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/bench_reduced.json $O/pmc_traffic.json "python bench.py $RED" > $O/pmc_traffic.log 2>&1
python -c "
import json; d=json.load(open('$O/pmc_traffic.json'))
for k,v in d.items(): print(k, v['dispatches'], 'hbm/alg', round(v['hbm_over_algorithmic'],3))"
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
timeout 200 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --max-probes 2 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 200 python bench.py --workload c1 --steps 5 --warmup 2 --no-cpu-baseline --max-probes 2 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --max-probes 4 --confirm-probes 1 --boundary-seconds 20 > $O/bench_c5.json 2> $O/bench_c5.err
python - <<PY
import json
for n in ("bench_c2", "bench_c1", "bench_c5"):
    try:
        d = json.load(open("$O/" + n + ".json"))
    except Exception as e:
        print(n, "ERR", e); continue
    b = d.get("boundary") or {}
    print(n, "value", d["value"], d.get("value_source"), "xrt", d["xrt"], "ms/step", d["ms_per_step"], "p50", d.get("p50_ttfa_ms"), "sustained", b.get("sustained_sessions"),
          "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), "single stt", (b.get("single_session") or {}).get("p50_stt_ms"), "roof", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("kernel", "")[:30])
    for r in b.get("probes", []) + b.get("repeats", []):
        print("   probe", r["sessions"], r.get("sustained"), "p50", r["p50_ttfa_ms"], "p95", r["p95_ttfa_ms"], "done", r["turns_completed"], "/", r["turns_arrived"], "lag", r["max_playback_lag_ms"], "stt", r.get("p50_stt_ms"), r.get("p95_stt_ms"))
PY
