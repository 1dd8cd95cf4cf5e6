set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gemm_dec_gpu.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 200 python tools/bench_dec.py gemm 64 > $O/gemm64.log 2>&1
timeout 200 python tools/bench_dec.py gemm 128 > $O/gemm128.log 2>&1
timeout 200 python tools/bench_dec.py layer 64 > $O/layer64.log 2>&1
timeout 200 python tools/bench_dec.py layer 128 > $O/layer128.log 2>&1
cat $O/gemm64.log $O/gemm128.log $O/layer64.log $O/layer128.log | grep -v amdgpu.ids | grep -E "t.gu|t.down|w.fc|w.logits|p.down|chain"
