# Round-6 job 10: wide decode projections with the residual rows requested before the main loop - parity + chain A/B
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6j}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gemm_dec_gpu.py -m gpu -x -q > $O/tests_gemm_dec.log 2>&1; tail -4 $O/tests_gemm_dec.log
(echo "== residual rows requested before the main loop (default)"; timeout 300 python tools/bench_decw.py chain; echo "== S2S_DECW_XPF=0 (rounds 4 - 5: in the epilogue)"; S2S_DECW_XPF=0 timeout 300 python tools/bench_decw.py chain; echo "== default again"; timeout 300 python tools/bench_decw.py chain) 2>&1 | grep -v amdgpu.ids > $O/bench_decw_chain.txt; cat $O/bench_decw_chain.txt
