cd $GRAFT_REPO_ROOT; O=gpurun_out/r4p; mkdir -p $O
run() { tag=$1; shift; timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe-only --boundary-seconds 20 "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r4p/$tag.json"))
    for r in d["boundary"]["probes"]:
        p = r["pool"]
        print("$tag", r["sessions"], "done", r["turns_completed"], "/", r["turns_arrived"], "p50", r["p50_ttfa_ms"], "p95", r["p95_ttfa_ms"], "lag", r["max_playback_lag_ms"], "stt", r.get("p50_stt_ms"), r.get("p95_stt_ms"),
              "stt_b", p["stt_batches"], "busy", p["stt_busy_s"], "tts_rounds", p["tts_rounds"], "round_s", p["tts_round_s"], p["tts_row_rounds"], "nursery", p.get("tts_nursery_rounds"), p.get("tts_nursery_s"))
    if d["boundary"].get("error"): print("ERR", d["boundary"]["error"])
except Exception as e:
    print("$tag ERR", e)
PY
tail -3 $O/$tag.err | grep -v amdgpu.ids
}
run Q1 --probe-sessions 1300 --stt-priority -1 --tts-nursery-slots 32 --tts-nursery-priority -1
run Q2 --probe-sessions 1300 --stt-priority -1 --tts-nursery-slots 32 --tts-nursery-priority -1 --tts-lead-s 0.7
run Q3 --probe-sessions 1300 --stt-priority -1 --tts-nursery-slots 64 --tts-nursery-priority -1 --stt-max-batch 24
