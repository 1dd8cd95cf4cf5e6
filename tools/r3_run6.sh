set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_tts_continuous_gpu.py -x -q > $O/tests.log 2>&1; tail -6 $O/tests.log
for l in 1 2 3; do
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe-sessions 889,1000 --stt-workers 1 --pool-lanes $l --prof-steps 0 > $O/bench_l$l.json 2> $O/bench_l$l.err; tail -3 $O/bench_l$l.err
done
python - <<'PY'
import json
for w in ("1", "2", "3"):
    d = json.load(open(f"gpurun_out/r3e/bench_l{w}.json"))
    b = d.get("boundary", {})
    print("tts lanes", w, "xrt", d["xrt"], "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), (b.get("single_session") or {}).get("p50_stt_ms"))
    for r in b.get("probes", []):
        print({k: r.get(k) for k in ("sessions", "turns_arrived", "turns_completed", "min_blocks_per_turn", "p50_stt_ms", "p95_stt_ms", "p50_ttfa_ms", "p95_ttfa_ms", "max_playback_lag_ms", "pool")})
    print(b.get("error"))
PY
