#!/bin/bash
# C5 open loop: STT workers (replicas, own streams) 2 and 3 at higher session counts
mkdir -p gpurun_out/r6w
for w in 2 3; do
  timeout 900 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 1200,1400,1600 --stt-workers $w > gpurun_out/r6w/c5_w${w}_hi.json 2> gpurun_out/r6w/c5_w${w}_hi.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r6w/c5_w${w}_hi.json").read().strip().splitlines()[-1])
print("workers $w")
for p in d["boundary"]["probes"]: print({k:p[k] for k in p if k in ("sessions","sustained","p50_ttfa_ms","p95_ttfa_ms","partials_answered_frac","p50_partial_ms","p95_partial_ms","min_xrt","max_lag_ms")})
P
done
