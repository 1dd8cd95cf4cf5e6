# Round 5, GPU job 7: packed polynomial GELU in the tile5 epilogue (parity, timing); which GEMM shapes stay on the older tile kernels
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5g}
mkdir -p $O
cd $R
T5_REPS=2 S2S_TILE5=2 timeout 200 python tools/proto/tile5_check.py > $O/tile5_var4.log 2>&1; echo "== var4 rc=$?"; grep -c OK $O/tile5_var4.log; grep -v 'OK$' $O/tile5_var4.log | tail -8
S2S_TILE5=2 timeout 120 python tools/tile5_phases.py fc1 > $O/phases_fc1.txt 2>&1; grep -v amdgpu.ids $O/phases_fc1.txt
S2S_GEMM_TRACE=1 timeout 300 python bench.py --serial --tts-merge 8 --steps 8 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_trace.json 2> $O/bench_trace.err; grep "s2s. gemm" $O/bench_trace.err | sort | uniq -c | sort -k3 > $O/gemm_shapes.txt; cat $O/gemm_shapes.txt
timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_whisper_gpu.py tests/test_true_dims_gpu.py tests/test_full_depth_gpu.py -x -q -m gpu 2>&1 | tail -3
