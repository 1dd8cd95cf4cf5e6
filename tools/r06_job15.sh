#!/bin/bash
# open loop: two wide paced TTS lanes instead of one (C3 and C5)
mkdir -p gpurun_out/r6w
show() { python - <<P
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
print("$2")
for p in d["boundary"]["probes"]: print({k:p[k] for k in p if k in ("sessions","sustained","p50_ttfa_ms","p95_ttfa_ms","partials_answered_frac","p95_partial_ms","min_xrt","max_lag_ms","p50_stt_ms","p95_stt_ms")}, p.get("pool",{}).get("tts_row_rounds"))
P
}
timeout 800 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 1500,1650,1800 --pool-lanes 2 > gpurun_out/r6w/c3_l2.json 2> gpurun_out/r6w/c3_l2.err
show gpurun_out/r6w/c3_l2.json "c3 pool-lanes 2"
timeout 700 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 1500,1700 --pool-lanes 2 > gpurun_out/r6w/c5_l2.json 2> gpurun_out/r6w/c5_l2.err
show gpurun_out/r6w/c5_l2.json "c5 pool-lanes 2 (workers 3)"
