# Round-6 job 2: the GQA-shared decode-step attention - parity, microbenchmark, the engine tests that run through it
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6b}
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_attention_gpu.py -m gpu -x -q -k attn_step -s > $O/tests_attn_step.log 2>&1; tail -8 $O/tests_attn_step.log
timeout 200 python tools/bench_attn_step.py 512 165 > $O/bench_attn_step.txt 2>&1; cat $O/bench_attn_step.txt
timeout 200 python tools/bench_attn_step.py 64 165 >> $O/bench_attn_step.txt 2>&1; tail -8 $O/bench_attn_step.txt
timeout 900 python -m pytest tests/test_tts_gpu.py tests/test_tts_continuous_gpu.py tests/test_full_depth_gpu.py tests/test_true_dims_gpu.py tests/test_whisper_gpu.py -m gpu -x -q > $O/tests_tts.log 2>&1; tail -5 $O/tests_tts.log
