# Round 5, GPU job 1 (re-entry): r05_run1.sh + r05_run2.sh merged into one box lease, kernels first.
#   bash tools/r05_job1.sh [out_dir]
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5a}
mkdir -p $O
cd $R
ls -d ${HF_HOME:-/root/.cache/huggingface} /root/.cache/huggingface/hub/* 2>&1 | head > $O/hf_home.txt      # any real checkpoint on the box?
nproc > $O/nproc.txt
# 1. the suite as the driver runs it
timeout 500 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
# 2. staggered 8-phase 256x256 GEMM: parity (forced on every shape), timing of the arms against gemm_tile4
T5_REPS=2 S2S_TILE5=2 S2S_TILE5_VAR=0 timeout 200 python tools/proto/tile5_check.py > $O/tile5_var0.log 2>&1; echo "== var0 rc=$?"; grep -c OK $O/tile5_var0.log; grep -v 'OK$' $O/tile5_var0.log | tail -12
for v in 1 2; do T5_REPS=1 S2S_TILE5=2 S2S_TILE5_VAR=$v timeout 150 python tools/proto/tile5_check.py > $O/tile5_var$v.log 2>&1; echo "== var$v rc=$?"; grep -v 'OK$' $O/tile5_var$v.log | tail -8; done
S2S_TILE5=0 timeout 150 python tools/proto/tile5_check.py > $O/tile5_off.log 2>&1; echo "== tile4 rc=$?"; tail -5 $O/tile5_off.log
# 3. software-pipelined tile4 main loops
for v in 1 2; do S2S_TILE4_PIPE=$v timeout 120 python tools/proto/tile4_check.py > $O/tile4_pipe$v.log 2>&1; echo "== S2S_TILE4_PIPE=$v"; tail -6 $O/tile4_pipe$v.log; done
# 4. the Whisper slot loop: parity
S2S_WHISPER_SLOTS=1 timeout 200 python -m pytest tests/test_whisper_slots_gpu.py -x -q -s > $O/tests_slots.log 2>&1; tail -15 $O/tests_slots.log
# 5. the captured TDT step (S2S_TDT_GRAPH=1): parity, then C2 with and without it
S2S_TDT_GRAPH=1 timeout 200 python -m pytest tests/test_parakeet_gpu.py tests/test_full_depth_gpu.py -x -q -m gpu > $O/tests_tdt_graph.log 2>&1; tail -3 $O/tests_tdt_graph.log
timeout 120 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --max-probes 2 > $O/c2_plain.json 2> $O/c2_plain.err
S2S_TDT_GRAPH=1 timeout 120 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --max-probes 2 > $O/c2_tdt_graph.json 2> $O/c2_tdt_graph.err
python - <<PY
import json
for n in ("c2_plain", "c2_tdt_graph"):
    try:
        d = json.load(open("$O/" + n + ".json")); s = (d.get("boundary") or {}).get("single_session") or {}
        print(n, "xrt", d["xrt"], "ms/step", d["ms_per_step"], "single TTFA", s.get("p50_ttfa_ms"), "stt", s.get("p50_stt_ms"))
    except Exception as e:
        print(n, "ERR", e)
PY
# 6. the slot loop in the open loop (same-box block: batch worker vs slot loop, one 20 s window each)
for cfg in "" "--stt-slots 128"; do
  timeout 240 python bench.py --probe-only --probe-sessions 1100 --boundary-seconds 20 --no-cpu-baseline --steps 2 --warmup 1 $cfg > "$O/ab_$(echo $cfg | tr -d ' -').json" 2> "$O/ab_$(echo $cfg | tr -d ' -').err"
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
