set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3o
mkdir -p $O
cd $R
for f in 0 48; do
timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe-sessions 800,950,1050 --tts-fill-first $f --prof-steps 0 > $O/bench_f$f.json 2> $O/bench_f$f.err; tail -2 $O/bench_f$f.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3o/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e); continue
    b = d.get("boundary", {})
    print(f, "xrt", d["xrt"])
    for r in b.get("probes", []):
        p = r.get("pool", {})
        print("   ", {k: r.get(k) for k in ("sessions", "p50_stt_ms", "p95_stt_ms", "p50_ttfa_ms", "p95_ttfa_ms", "max_playback_lag_ms")}, "stt batches", p.get("stt_batches"), "busy", p.get("stt_busy_s"), "rounds", p.get("tts_rounds"), p.get("tts_row_rounds"), "cores", p.get("host_cores_busy"))
PY
