#!/bin/bash
# open loop: C5 with 4 STT workers; C3 with 1 / 2 STT workers around the sustained count
mkdir -p gpurun_out/r6w
show() { python - <<P
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
print("$2")
for p in d["boundary"]["probes"]: print({k:p[k] for k in p if k in ("sessions","sustained","p50_ttfa_ms","p95_ttfa_ms","partials_answered_frac","p50_partial_ms","p95_partial_ms","min_xrt","max_lag_ms","p50_stt_ms","p95_stt_ms")})
P
}
timeout 700 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 1300,1400,1500 --stt-workers 4 > gpurun_out/r6w/c5_w4.json 2> gpurun_out/r6w/c5_w4.err
show gpurun_out/r6w/c5_w4.json "c5 workers 4"
timeout 900 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 1500,1650,1800 --stt-workers 2 > gpurun_out/r6w/c3_w2.json 2> gpurun_out/r6w/c3_w2.err
show gpurun_out/r6w/c3_w2.json "c3 workers 2"
timeout 700 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 1500,1650 --stt-workers 1 > gpurun_out/r6w/c3_w1.json 2> gpurun_out/r6w/c3_w1.err
show gpurun_out/r6w/c3_w1.json "c3 workers 1"
