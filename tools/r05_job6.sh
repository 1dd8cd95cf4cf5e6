# Round 5, GPU job 6: tile5 as the default tile GEMM: the GPU suite, the serial C3 profile (kernel families), the default bench line
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5f}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -5 $O/tests_gpu.log
SER="--serial --tts-merge 8 --steps 8 --warmup 8 --no-cpu-baseline --no-boundary --parity-sessions 0"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o c3s -- python bench.py $SER > $O/bench_c3_serial_prof.json 2> $O/bench_c3_serial_prof.err
timeout 300 python bench.py $SER --breakdown > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
cp $(find $O/prof_serial -name "*kernel_stats.csv" | head -1) $O/c3_serial_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_serial
python tools/prof_families.py $O/c3_serial_kernel_stats.csv 24 $O/bench_c3_serial.json
S2S_TILE5=0 timeout 300 python bench.py $SER --breakdown > $O/bench_c3_serial_tile4.json 2> $O/bench_c3_serial_tile4.err
python - <<PY
import json
for n in ("bench_c3_serial", "bench_c3_serial_tile4"):
    try:
        d = json.load(open("$O/" + n + ".json")); print(n, "ms/step", d["ms_per_step"], "xrt", d.get("xrt"), "frac", d["roofline"]["frac"], {k: (v["kernel_ms_per_step"], v["frac"]) for k, v in d["roofline"].get("families", {}).items()})
    except Exception as e:
        print(n, "ERR", e)
PY
