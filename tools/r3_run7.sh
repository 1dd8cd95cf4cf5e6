set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe-sessions 600,850 --stt-workers 1 --pool-lanes 3 --prof-steps 0 > $O/bench_a.json 2> $O/bench_a.err; tail -3 $O/bench_a.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3f/bench_a.json"))
b = d.get("boundary", {})
print("xrt", d["xrt"], "single", (b.get("single_session") or {}))
for r in b.get("probes", []):
    print({k: r.get(k) for k in ("sessions", "turns_arrived", "turns_completed", "p50_stt_ms", "p95_stt_ms", "p50_ttfa_ms", "p95_ttfa_ms", "max_playback_lag_ms", "pool")})
print(b.get("error"))
PY
