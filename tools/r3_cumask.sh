set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p
mkdir -p $O
cd $R
for c in 8 4; do
timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe-sessions 700,900 --tts-lane-cus $c --prof-steps 0 > $O/bench_c$c.json 2> $O/bench_c$c.err; tail -2 $O/bench_c$c.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3p/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e); continue
    b = d.get("boundary", {})
    print(f, "xrt", d["xrt"], b.get("error"))
    for r in b.get("probes", []):
        p = r.get("pool", {})
        print("   ", {k: r.get(k) for k in ("sessions", "p50_stt_ms", "p95_stt_ms", "p50_ttfa_ms", "p95_ttfa_ms", "max_playback_lag_ms")}, "stt batches", p.get("stt_batches"), "busy", p.get("stt_busy_s"), "rounds", p.get("tts_rounds"), p.get("tts_row_rounds"), "cores", p.get("host_cores_busy"))
PY
