set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
for cfg in "1e9 3" "2.0 3" "1.0 3" "2.0 4"; do
set -- $cfg
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe-sessions 700,900 --tts-lead-s $1 --pool-lanes $2 --prof-steps 0 > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; tail -2 $O/bench_$1_$2.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3g/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e); continue
    b = d.get("boundary", {})
    print(f, "xrt", d["xrt"], "single", (b.get("single_session") or {}).get("p50_ttfa_ms"))
    for r in b.get("probes", []):
        print({k: r.get(k) for k in ("sessions", "turns_arrived", "turns_completed", "p50_stt_ms", "p95_stt_ms", "p50_ttfa_ms", "p95_ttfa_ms", "max_playback_lag_ms", "pool")})
    print(b.get("error"))
PY
