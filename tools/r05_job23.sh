# Round 5, GPU job 23: vocabulary argmax of the Whisper picks by 8 workgroups per row (ticketed combine): parity + the kernel's time
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5w}
mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_whisper_gpu.py tests/test_whisper_slots_gpu.py tests/test_true_dims_gpu.py tests/test_pipeline_gpu.py tests/test_server_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python -m pytest tests/test_full_depth_gpu.py -x -q -m gpu -k whisper 2>&1 | tail -2
SER="--serial --tts-merge 8 --steps 4 --warmup 4 --no-cpu-baseline --no-boundary --parity-sessions 8"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py $SER > $O/serial.json 2> $O/serial.err
python - <<PY
import csv, glob, json
f = glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "pick" in r["Name"]:
        print(r["Name"][:60], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
d = json.load(open("$O/serial.json")); print("ms/step", d["ms_per_step"], "parity", {k: v for k, v in (d.get("parity") or {}).items() if k != "what"})
PY
rm -rf $O/prof
