set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_whisper_gpu.py tests/test_true_dims_gpu.py tests/test_server_gpu.py -x -q > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 200 python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-boundary --breakdown > $O/bench_serial.json 2> $O/bench_serial.err; grep breakdown $O/bench_serial.err
S2S_TILE4=0 timeout 200 python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-boundary --breakdown > $O/bench_serial_t3.json 2> $O/bench_serial_t3.err; grep breakdown $O/bench_serial_t3.err
timeout 400 python tools/fault_hunt.py 12 > $O/fault_hunt.log 2>&1; tail -15 $O/fault_hunt.log
