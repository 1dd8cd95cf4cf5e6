# Round 5, GPU job 19: the GPU suite at HEAD (after the f3 change); serial C3 with the early-prologue arm (variant 12) vs the default (4), same box
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5s}
mkdir -p $O
cd $R
timeout 700 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
SER="--serial --tts-merge 8 --steps 8 --warmup 8 --no-cpu-baseline --no-boundary --parity-sessions 0 --breakdown"
for v in 4 12 4 12; do
  S2S_TILE5_VAR=$v timeout 300 python bench.py $SER > $O/serial_var${v}.json 2> $O/serial_var${v}.err
  python -c "
import json; d=json.load(open('$O/serial_var${v}.json')); print('variant $v', 'ms/step', d['ms_per_step'], 'gemm_tile', d['roofline']['families']['gemm_tile']['kernel_ms_per_step'], d['roofline']['families']['gemm_tile']['frac'])"
done
