# Round 5, GPU job 4: straight-line tile5 epilogue (compile-time activation, hand-counted waits) - parity on every arm, phases, timing
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5d}
mkdir -p $O
cd $R
for v in 0 4 12; do
  T5_REPS=2 S2S_TILE5=2 S2S_TILE5_VAR=$v timeout 200 python tools/proto/tile5_check.py > $O/tile5_var$v.log 2>&1; echo "== var$v rc=$?"; grep -c OK $O/tile5_var$v.log; grep -v 'OK$' $O/tile5_var$v.log | tail -8
done
for v in 0 12; do
  S2S_TILE5=2 S2S_TILE5_VAR=$v timeout 120 python tools/tile5_phases.py qkv o fc1 fc2 > $O/phases_var$v.txt 2>&1; echo "== phases var$v rc=$?"; grep -v amdgpu.ids $O/phases_var$v.txt
done
