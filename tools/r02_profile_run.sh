set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2p
mkdir -p $O
cd $R
# 1. serial schedule under rocprofv3 kernel trace: the CSV the roofline fractions are recomputed from
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o c3s -- python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
# 2. default (pipelined) schedule under rocprofv3
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o c3d -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_c3_default_prof.json 2> $O/bench_c3_default_prof.err
# 3. other workloads (bench lines)
timeout 150 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 150 python bench.py --workload c1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err
timeout 150 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_c5.json 2> $O/bench_c5.err
# 4. PMC traffic on a reduced step (separate passes)
RED="--serial --steps 1 --warmup 0 --no-cpu-baseline --no-boundary --decode-tokens 4 --reply-frames 8"
timeout 100 python bench.py $RED > $O/bench_reduced.json 2> $O/bench_reduced.err
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py $RED > /dev/null 2> $O/pmc_fetch.err
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py $RED > /dev/null 2> $O/pmc_write.err
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/bench_reduced.json $O/pmc_traffic.json "python bench.py $RED" > $O/pmc_traffic.log 2>&1
# 5. ordered kernel sequence of one streamed codec chunk
timeout 100 rocprofv3 --kernel-trace --output-format csv -d $O/codect -o c -- python tools/bench_codec.py 64 4 > /dev/null 2>&1
python tools/trace_seq.py $(find $O/codect -name "*kernel_trace.csv" | head -1) 140 > $O/codec_kernel_sequence.txt
rm -rf $O/codect
# keep only the small summaries
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls -la $O $O/prof_serial/* | head -40
for f in $O/prof_serial $O/prof_default; do find $f -name "*kernel_stats.csv" | head -2; done
python tools/prof_families.py $(find $O/prof_serial -name "*kernel_stats.csv" | head -1) 5 $O/bench_c3_serial.json
tail -3 $O/pmc_traffic.log
