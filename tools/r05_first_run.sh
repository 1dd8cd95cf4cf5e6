# First GPU job of round 5: everything that was written at the end of round 4 without GPU time left, in the order of what it can break.
#   bash tools/r05_first_run.sh [out_dir]
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5a}
mkdir -p $O
cd $R
# 1. the suite as the driver runs it (ArenaPlan refactor, ABI 3, conftest guard fixture: all on the default path)
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
# 2. the Whisper slot loop: parity first
S2S_WHISPER_SLOTS=1 timeout 200 python -m pytest tests/test_whisper_slots_gpu.py -x -q -s > $O/tests_slots.log 2>&1; tail -15 $O/tests_slots.log
# 3. ... then what it does to the open loop (same-box block: batch worker vs slot loop, 20 s windows)
for cfg in "" "--stt-slots 128" "--stt-slots 128 --stt-slot-steps 8" "--stt-slots 128 --stt-priority -1"; do
  timeout 400 python bench.py --probe-only --probe-sessions 1000,1200 --boundary-seconds 20 --no-cpu-baseline $cfg > "$O/ab_$(echo $cfg | tr -d ' -').json" 2> "$O/ab_$(echo $cfg | tr -d ' -').err"
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
              "stt", r.get("p50_stt_ms"), r.get("p95_stt_ms"), r["pool"])
PY
# 4. the suite under the poisoned-gap allocator (S2S_GUARD=1); mode 2 and the bench starts: tools/guard_run.sh
S2S_GUARD=1 timeout 900 python -m pytest tests -m gpu -x -q > $O/tests_guard1.log 2>&1; tail -3 $O/tests_guard1.log
