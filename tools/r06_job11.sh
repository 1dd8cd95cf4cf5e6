# Round-6 job 11: one-wave-per-item attention step for short caches - parity, microbenchmark, TTS engine tests, serial profile
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6k}
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_attention_gpu.py -m gpu -x -q -k attn_step -s > $O/tests_attn_step.log 2>&1; tail -12 $O/tests_attn_step.log
timeout 200 python tools/bench_attn_step.py 512 165 2>&1 | grep -v amdgpu.ids > $O/bench_attn_step.txt; timeout 200 python tools/bench_attn_step.py 256 165 2>&1 | grep -v amdgpu.ids >> $O/bench_attn_step.txt; cat $O/bench_attn_step.txt
timeout 900 python -m pytest tests/test_tts_gpu.py tests/test_tts_continuous_gpu.py tests/test_full_depth_gpu.py tests/test_true_dims_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > $O/tests_tts.log 2>&1; tail -4 $O/tests_tts.log
bash tools/r06_prof_run.sh ${1:-r6k}_prof 2>&1 | grep -v "^+" | tail -8
