# Round 5, GPU job 12: open loop at 1400 sessions, the pool's final-STT batch cap (32 = round-4 default, 48, 64, 0 = engine max 128); one 20 s window each, same box
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5l}
mkdir -p $O
cd $R
for cap in 32 48 64 0; do
  timeout 240 python bench.py --probe-only --probe-sessions 1400 --boundary-seconds 20 --no-cpu-baseline --steps 2 --warmup 1 --stt-max-batch $cap > $O/ab_cap$cap.json 2> $O/ab_cap$cap.err
done
python - <<PY
import glob, json
for f in sorted(glob.glob("$O/ab_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e); continue
    for r in d["boundary"]["probes"]:
        print(f.split("/")[-1], r["sessions"], "done", r["turns_completed"], "/", r["turns_arrived"], "ttfa", r["p50_ttfa_ms"], r["p95_ttfa_ms"], "lag", r["max_playback_lag_ms"],
              "stt", r.get("p50_stt_ms"), r.get("p95_stt_ms"), {k: r["pool"][k] for k in ("stt_batches", "max_stt_batch", "stt_busy_s", "tts_rounds")})
PY
