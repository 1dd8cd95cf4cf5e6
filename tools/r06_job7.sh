# Round-6 job 7: C5 open-loop A/B at fixed session counts - STT batch cap 32 (default) vs 64 (engine rows 64), at HEAD (fused TDT step, direct-V attention)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6g}
mkdir -p $O
cd $R
COMMON="--workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --parity-sessions 0 --probe-only --probe-sessions 800,900"
timeout 600 python bench.py $COMMON > $O/c5_cap32.json 2> $O/c5_cap32.err; echo "cap32 rc $?"
timeout 600 python bench.py $COMMON --pool-max-batch 64 --stt-max-batch 64 > $O/c5_cap64.json 2> $O/c5_cap64.err; echo "cap64 rc $?"
timeout 600 python bench.py $COMMON --pool-max-batch 64 --stt-max-batch 48 > $O/c5_cap48.json 2> $O/c5_cap48.err; echo "cap48 rc $?"
python - <<PY
import json
for n in ("c5_cap32", "c5_cap64", "c5_cap48"):
    try:
        d = json.load(open("$O/" + n + ".json"))
    except Exception as e:
        print(n, "ERR", e); continue
    for r_ in (d.get("boundary") or {}).get("probes", []):
        print(n, "probe", r_["sessions"], r_.get("sustained"), "p50", r_["p50_ttfa_ms"], "p95", r_["p95_ttfa_ms"], "done", r_["turns_completed"], "/", r_["turns_arrived"], "stt", r_.get("p50_stt_ms"), r_.get("p95_stt_ms"),
              "partials", r_.get("partials_answered"), "/", r_.get("partials_sent"), r_.get("partials_answered_frac"), r_.get("p50_partial_ms"), r_.get("p95_partial_ms"), "pool", r_["pool"].get("stt_batches"), r_["pool"].get("max_stt_batch"), r_["pool"].get("stt_busy_s"))
PY
