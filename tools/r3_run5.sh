set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_tts_continuous_gpu.py tests/test_tts_gpu.py tests/test_sampling_gpu.py tests/test_pipeline_gpu.py tests/test_full_depth_gpu.py -x -q > $O/tests.log 2>&1; tail -15 $O/tests.log
for w in 1 2; do
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe-sessions 889,1000 --stt-workers $w --prof-steps 0 > $O/bench_w$w.json 2> $O/bench_w$w.err; tail -3 $O/bench_w$w.err
done
python - <<'PY'
import json
for w in ("1", "2"):
    d = json.load(open(f"gpurun_out/r3d/bench_w{w}.json"))
    b = d.get("boundary", {})
    print("stt workers", w, "xrt", d["xrt"], "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), (b.get("single_session") or {}).get("p50_stt_ms"))
    for r in b.get("probes", []):
        print({k: r.get(k) for k in ("sessions", "turns_arrived", "turns_completed", "min_blocks_per_turn", "p50_stt_ms", "p95_stt_ms", "p50_ttfa_ms", "p95_ttfa_ms", "max_playback_lag_ms", "pool")})
    print(b.get("error"))
PY
