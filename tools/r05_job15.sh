# Round 5, GPU job 15: flash attention with direct global->LDS tiles in a double buffer (one barrier per tile) vs the register-staged loop
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5o}
mkdir -p $O
cd $R
for a in 1 0; do S2S_ATTN_DMA=$a timeout 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu | tee -a $O/bench_attn.txt; done
timeout 400 python -m pytest tests/test_attention_gpu.py tests/test_whisper_gpu.py tests/test_parakeet_gpu.py tests/test_true_dims_gpu.py -x -q -m gpu 2>&1 | tail -3
