# Round 5, GPU job 20: unrolled vocabulary argmax (Whisper / TTS picks): parity tests, the pick kernel's time in the serial schedule
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5t}
mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_whisper_gpu.py tests/test_whisper_slots_gpu.py tests/test_tts_gpu.py tests/test_tts_continuous_gpu.py tests/test_sampling_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
SER="--serial --tts-merge 8 --steps 4 --warmup 4 --no-cpu-baseline --no-boundary --parity-sessions 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py $SER > $O/serial.json 2> $O/serial.err
python - <<PY
import csv, glob
f = glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "pick" in r["Name"] or "argmax" in r["Name"]:
        print(r["Name"][:60], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
PY
rm -rf $O/prof
