# Round-6 job 9: tap-reuse convolution kernel - parity, A/B at the codec's shapes, codec / TTS tests
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6i}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_attention_gpu.py -m gpu -x -q -k "conv" -s > $O/tests_conv.log 2>&1; tail -14 $O/tests_conv.log
timeout 300 python tools/bench_conv.py 512 2>&1 | grep -v amdgpu.ids > $O/bench_conv.txt; cat $O/bench_conv.txt
timeout 900 python -m pytest tests/test_tts_gpu.py tests/test_true_dims_gpu.py tests/test_tts_continuous_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > $O/tests_tts.log 2>&1; tail -4 $O/tests_tts.log
(timeout 200 python tools/bench_codec.py 64 16; S2S_CONV7=0 timeout 200 python tools/bench_codec.py 64 16; timeout 200 python tools/bench_codec.py 512 4; S2S_CONV7=0 timeout 200 python tools/bench_codec.py 512 4) 2>&1 | grep -v amdgpu.ids > $O/bench_codec.txt; cat $O/bench_codec.txt
