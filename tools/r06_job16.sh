#!/bin/bash
# final bench lines at HEAD: C5 (new search start), C3 default
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6fin3; mkdir -p $O; cd $R
T0=$SECONDS; timeout 1300 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 took $((SECONDS - T0)) s"
T0=$SECONDS; timeout 1300 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "c3 took $((SECONDS - T0)) s"
python - <<P
import json
for n in ("bench_c5","bench_c3_default"):
    d=json.load(open("$O/"+n+".json"))
    print(n, "value", d["value"], "xrt", d["xrt"], "ms", d["ms_per_step"], d["wall_s"]["total"], "roof", d["roofline"]["frac"], d["roofline"].get("traffic"), (d["roofline"].get("traffic_source") or "")[:10])
    b=d["boundary"]
    for r in b["probes"]+b["repeats"]: print("  ", r["sessions"], r["sustained"], r["p50_ttfa_ms"], r["p95_ttfa_ms"], r["max_playback_lag_ms"], r.get("partials_answered_frac"), r.get("p95_partial_ms"), r.get("retry_of_first_failure"))
P
