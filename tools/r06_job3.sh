# Round-6 job 3: flash attention with V transposed in the LDS read (no transpose_v pass) - parity + A/B
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6c}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_attention_gpu.py tests/test_whisper_gpu.py tests/test_parakeet_gpu.py tests/test_true_dims_gpu.py -m gpu -x -q > $O/tests_attn.log 2>&1; tail -6 $O/tests_attn.log
(echo "== direct V (ds_read_b64_tr_b16), no transpose pass"; timeout 200 python tools/bench_attn.py; echo "== S2S_ATTN_VTR=0: transpose_v + V^T kernel (round 5)"; S2S_ATTN_VTR=0 timeout 200 python tools/bench_attn.py; echo "== direct V again"; timeout 200 python tools/bench_attn.py) > $O/bench_attn.txt 2>&1; cat $O/bench_attn.txt
