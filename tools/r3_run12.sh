set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_whisper_gpu.py tests/test_true_dims_gpu.py tests/test_full_depth_gpu.py tests/test_pipeline_gpu.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.err; cat $O/bench_default.time
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3k/bench_default.json"))
b = d.get("boundary", {})
print("value", d["value"], d.get("value_source"), "xrt", d["xrt"], "ms_per_step", d["ms_per_step"], "p50", d.get("p50_ttfa_ms"), "single", (b.get("single_session") or {}).get("p50_ttfa_ms"))
for r in b.get("probes", []):
    print({k: r.get(k) for k in ("sessions", "sustained", "turns_arrived", "turns_completed", "min_blocks_per_turn", "p50_stt_ms", "p50_ttfa_ms", "p95_ttfa_ms", "max_playback_lag_ms")})
print("roofline", {k: d["roofline"][k] for k in ("bound", "achieved", "frac", "avg_kernel_us")} if d.get("roofline") else None)
print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k != "sample"})
PY
