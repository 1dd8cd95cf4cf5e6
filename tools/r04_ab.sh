cd $GRAFT_REPO_ROOT; O=gpurun_out/r4j; mkdir -p $O
run() { tag=$1; shift; timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe-only --boundary-seconds 20 "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r4j/$tag.json"))
    for r in d["boundary"]["probes"]:
        p = r["pool"]
        print("$tag", r["sessions"], "done", r["turns_completed"], "/", r["turns_arrived"], "p50", r["p50_ttfa_ms"], "p95", r["p95_ttfa_ms"], "lag", r["max_playback_lag_ms"], "stt", r.get("p50_stt_ms"), r.get("p95_stt_ms"),
              "stt_b", p["stt_batches"], "busy", p["stt_busy_s"], "tts_rounds", p["tts_rounds"], "round_s", p["tts_round_s"], p["tts_row_rounds"])
except Exception as e:
    print("$tag ERR", e)
PY
tail -2 $O/$tag.err | grep -v amdgpu.ids
}
run S2 --probe-sessions 1000,1100 --tts-lead-s 0.5 --tts-admit-wait-s 0.25 --stt-stages 2
run S1 --probe-sessions 1000,1100 --tts-lead-s 0.5 --tts-admit-wait-s 0.25 --stt-stages 1
