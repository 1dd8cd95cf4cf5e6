# Round-3 baseline on the GPU box: full GPU suite, then the serial bench line with the per-stage breakdown.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -5 $O/tests_gpu.log
timeout 200 python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-boundary --breakdown > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
timeout 200 python bench.py --serial --sessions 128 --steps 2 --warmup 1 --no-cpu-baseline --no-boundary --breakdown > $O/bench_c3_serial128.json 2> $O/bench_c3_serial128.err
tail -40 $O/bench_c3_serial.err
