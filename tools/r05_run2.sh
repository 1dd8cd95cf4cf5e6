# Round 5, GPU job 2: the staggered 8-phase 256x256 GEMM (gemm_tile5_kernel): parity (forced on every shape, repeated: race screen),
# timing of the three arms against gemm_tile4 on the same box, then the whole GPU suite with the kernel forced on.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5b}
mkdir -p $O
cd $R
T5_REPS=3 S2S_TILE5=2 S2S_TILE5_VAR=0 timeout 300 python tools/proto/tile5_check.py > $O/tile5_var0.log 2>&1; echo "== var0 rc=$?"; grep -c OK $O/tile5_var0.log; grep -v 'OK$' $O/tile5_var0.log | tail -12
for v in 1 2; do T5_REPS=1 S2S_TILE5=2 S2S_TILE5_VAR=$v timeout 200 python tools/proto/tile5_check.py > $O/tile5_var$v.log 2>&1; echo "== var$v rc=$?"; grep -v 'OK$' $O/tile5_var$v.log | tail -8; done
S2S_TILE5=0 timeout 200 python tools/proto/tile5_check.py > $O/tile5_off.log 2>&1; echo "== tile4 rc=$?"; tail -5 $O/tile5_off.log
S2S_TILE5=2 timeout 500 python -m pytest tests -m gpu -x -q > $O/tests_gpu_tile5.log 2>&1; tail -5 $O/tests_gpu_tile5.log
