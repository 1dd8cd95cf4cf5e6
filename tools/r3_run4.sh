set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_tts_continuous_gpu.py tests/test_tts_gpu.py tests/test_sampling_gpu.py tests/test_pipeline_gpu.py -x -q > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --max-probes 4 > $O/bench_c3.json 2> $O/bench_c3.err; tail -5 $O/bench_c3.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3c/bench_c3.json"))
b = d.get("boundary", {})
print("value", d["value"], d.get("value_source"), "xrt", d["xrt"], "single", (b.get("single_session") or {}).get("p50_ttfa_ms"))
for r in b.get("probes", []):
    print({k: r.get(k) for k in ("sessions", "sustained", "turns_arrived", "turns_completed", "min_blocks_per_turn", "p50_ttfa_ms", "p95_ttfa_ms", "max_playback_lag_ms", "pool")})
print(b.get("error"))
PY
