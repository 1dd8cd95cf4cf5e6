set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gemm_dec_gpu.py tests/test_whisper_gpu.py tests/test_true_dims_gpu.py tests/test_full_depth_gpu.py tests/test_server_gpu.py tests/test_pipeline_gpu.py -x -q > $O/tests.log 2>&1; tail -12 $O/tests.log
grep -h "LayerNorm-free" $O/tests.log | head
timeout 200 python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-boundary --breakdown > $O/bench_serial.json 2> $O/bench_serial.err; grep breakdown $O/bench_serial.err
S2S_NO_LN_FOLD=1 timeout 200 python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-boundary --breakdown > $O/bench_serial_noln.json 2> $O/bench_serial_noln.err; grep breakdown $O/bench_serial_noln.err
