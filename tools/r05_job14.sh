# Round 5, GPU job 14: open loop at 1400 sessions, persistent (variant 4) vs one-tile-per-workgroup (variant 0) tile GEMM: a persistent launch holds every CU for its
# whole duration, the TTS lane's latency-bound kernels on the other stream wait for it; same box, 20 s windows, two arrival seeds... (one each here)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5n}
mkdir -p $O
cd $R
for var in 4 0 4 0; do
  n=$(ls $O | grep -c "ab_var$var")
  S2S_TILE5_VAR=$var timeout 240 python bench.py --probe-only --probe-sessions 1400 --boundary-seconds 20 --no-cpu-baseline --steps 2 --warmup 1 > $O/ab_var${var}_$n.json 2> $O/ab_var${var}_$n.err
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
              "stt", r.get("p50_stt_ms"), r.get("p95_stt_ms"), {k: r["pool"][k] for k in ("stt_batches", "max_stt_batch", "stt_busy_s", "tts_rounds", "tts_round_s")})
PY
