#!/bin/bash
# C3 open loop at HEAD: first chunks in a nursery loop + STT worker on a high-priority stream (the round-4 experiment again)
mkdir -p gpurun_out/r6w
show() { python - <<P
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
print("$2")
for p in d["boundary"]["probes"]: print({k:p[k] for k in p if k in ("sessions","sustained","p50_ttfa_ms","p95_ttfa_ms","max_playback_lag_ms","p50_stt_ms","p95_stt_ms")})
P
}
timeout 800 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 1500,1650,1800 --tts-nursery-slots 32 --stt-priority -1 --tts-nursery-priority -1 > gpurun_out/r6w/c3_nursery.json 2> gpurun_out/r6w/c3_nursery.err
show gpurun_out/r6w/c3_nursery.json "c3 nursery 32 + stt priority"
timeout 800 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 1500,1650 --stt-priority -1 > gpurun_out/r6w/c3_sttprio.json 2> gpurun_out/r6w/c3_sttprio.err
show gpurun_out/r6w/c3_sttprio.json "c3 stt priority only"
