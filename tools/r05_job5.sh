# Round 5, GPU job 5: tile5 with the straight-line epilogue (plain loads, asm stores): parity on every arm, timing, phases
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5e}
mkdir -p $O
cd $R
for v in 0 4 12 28; do
  T5_REPS=2 S2S_TILE5=2 S2S_TILE5_VAR=$v S2S_TILE5_SPREAD=100 timeout 200 python tools/proto/tile5_check.py > $O/tile5_var$v.log 2>&1; echo "== var$v rc=$?"; grep -c OK $O/tile5_var$v.log; grep -v 'OK$' $O/tile5_var$v.log | tail -8
done
S2S_TILE5=2 S2S_TILE5_VAR=28 S2S_TILE5_SPREAD=50 timeout 200 python tools/proto/tile5_check.py 2>&1 | grep -v 'OK$' | tail -5
for v in 4 12 28; do
  S2S_TILE5=2 S2S_TILE5_VAR=$v timeout 120 python tools/tile5_phases.py qkv o fc1 fc2 > $O/phases_var$v.txt 2>&1; echo "== phases var$v rc=$?"; grep -v amdgpu.ids $O/phases_var$v.txt
done
