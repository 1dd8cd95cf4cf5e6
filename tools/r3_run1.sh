set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_dec_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r3_t1.log
timeout 300 python tools/bench_dec.py sweep 64 > gpurun_out/r3_sweep64.log 2>&1
timeout 300 python tools/bench_dec.py sweep 128 > gpurun_out/r3_sweep128.log 2>&1
timeout 200 python tools/bench_dec.py layer 64 > gpurun_out/r3_layer64.log 2>&1
timeout 200 python tools/bench_dec.py layer 128 > gpurun_out/r3_layer128.log 2>&1
tail -30 gpurun_out/r3_t1.log
