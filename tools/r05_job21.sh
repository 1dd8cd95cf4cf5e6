# Round 5, GPU job 21: open loop at 1450 sessions: one STT worker (default) vs the 2-stage STT pipeline vs two STT workers (the encoder is 1.5 x faster than when round 4 measured these)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5u}
mkdir -p $O
cd $R
for cfg in "" "--stt-stages 2" "--stt-workers 2"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 240 python bench.py --probe-only --probe-sessions 1450 --boundary-seconds 20 --no-cpu-baseline --steps 2 --warmup 1 $cfg > "$O/ab_$tag.json" 2> "$O/ab_$tag.err"
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
              "stt", r.get("p50_stt_ms"), r.get("p95_stt_ms"), {k: r["pool"].get(k) for k in ("stt_batches", "max_stt_batch", "stt_busy_s", "tts_rounds", "tts_round_s")})
PY
