# Round-6 job 5: fused TDT step (8 launches instead of 14) - parity, C2 A/B; tiny bench with the teacher-forced TTS parity statistics
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6e}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_parakeet_gpu.py tests/test_parakeet_progressive_gpu.py tests/test_full_depth_gpu.py tests/test_true_dims_gpu.py -m gpu -x -q -k "parakeet or conformer or tdt or progressive or growing or cached or timestamped" > $O/tests_tdt.log 2>&1; tail -5 $O/tests_tdt.log
timeout 300 python bench.py --workload tiny --steps 2 --warmup 1 --no-boundary --no-pmc --no-cpu-baseline > $O/bench_tiny.json 2> $O/bench_tiny.err; echo "tiny rc $?"; tail -3 $O/bench_tiny.err
C2="--workload c2 --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --parity-sessions 0 --max-probes 1 --confirm-probes 0 --boundary-seconds 12"
timeout 500 python bench.py $C2 > $O/bench_c2_fused.json 2> $O/bench_c2_fused.err; echo "c2 fused rc $?"
S2S_TDT_FUSED=0 timeout 500 python bench.py $C2 > $O/bench_c2_unfused.json 2> $O/bench_c2_unfused.err; echo "c2 unfused rc $?"
python - <<PY
import json
d = json.load(open("$O/bench_tiny.json"))
print("tiny parity", json.dumps({k: v for k, v in (d.get("parity") or {}).get("tts", {}).items() if k != "what"}))
for n in ("bench_c2_fused", "bench_c2_unfused"):
    try:
        d = json.load(open("$O/" + n + ".json"))
    except Exception as e:
        print(n, "ERR", e); continue
    b = d.get("boundary") or {}
    print(n, "ms/step", d["ms_per_step"], "stages", d["config"]["stages_ms_per_step"], "single", (b.get("single_session") or {}).get("p50_ttfa_ms"), (b.get("single_session") or {}).get("p50_stt_ms"), "err", b.get("error"))
PY
