#!/bin/bash
# C5 open loop after the one-call tick features: STT workers 3 and 2
mkdir -p gpurun_out/r6w
show() { python - <<P
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
print("$2")
for p in d["boundary"]["probes"]: print({k:p[k] for k in p if k in ("sessions","sustained","p50_ttfa_ms","p95_ttfa_ms","partials_answered_frac","p50_partial_ms","p95_partial_ms","min_xrt","max_lag_ms","p50_stt_ms","p95_stt_ms")})
P
}
timeout 700 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 1300,1500,1700 --stt-workers 3 > gpurun_out/r6w/c5_rows_w3.json 2> gpurun_out/r6w/c5_rows_w3.err
show gpurun_out/r6w/c5_rows_w3.json "c5 workers 3 (rows)"
timeout 700 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 1300,1500 --stt-workers 2 > gpurun_out/r6w/c5_rows_w2.json 2> gpurun_out/r6w/c5_rows_w2.err
show gpurun_out/r6w/c5_rows_w2.json "c5 workers 2 (rows)"
