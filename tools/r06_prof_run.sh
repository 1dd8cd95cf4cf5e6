# Round-6 serial profile (GPU box): the kernel families of the DEFAULT schedule's launches (--tts-merge 4, what the default line samples) under rocprofv3
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6p}
mkdir -p $O
cd $R
timeout 200 python tools/bench_attn_step.py 512 165 > $O/bench_attn_step.txt 2>&1; cat $O/bench_attn_step.txt
SER="--serial --tts-merge 4 --steps 8 --warmup 4 --no-cpu-baseline --no-boundary --no-pmc --parity-sessions 0"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o c3s -- python bench.py $SER > $O/bench_c3_serial_prof.json 2> $O/bench_c3_serial_prof.err
cp $(find $O/prof_serial -name "*kernel_stats.csv" | head -1) $O/c3_serial_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_serial
timeout 400 python bench.py $SER --breakdown > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
grep breakdown $O/bench_c3_serial.err
python tools/prof_families.py $O/c3_serial_kernel_stats.csv 16 $O/bench_c3_serial.json
