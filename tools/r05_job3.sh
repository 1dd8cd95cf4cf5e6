# Round 5, GPU job 3: persistent workgroups spread over one tile period (tile5 variant 28)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5c}
mkdir -p $O
cd $R
for sp in 100 50 25; do
  S2S_TILE5=2 S2S_TILE5_VAR=28 S2S_TILE5_SPREAD=$sp timeout 120 python tools/tile5_phases.py qkv o fc2 > $O/phases_var28_s$sp.txt 2>&1; echo "== phases var28 spread $sp rc=$?"; grep -v amdgpu.ids $O/phases_var28_s$sp.txt
  T5_REPS=1 S2S_TILE5=2 S2S_TILE5_VAR=28 S2S_TILE5_SPREAD=$sp timeout 200 python tools/proto/tile5_check.py > $O/tile5_var28_s$sp.log 2>&1; echo "== var28 spread $sp rc=$?"; grep -c OK $O/tile5_var28_s$sp.log; grep -v 'OK$' $O/tile5_var28_s$sp.log | tail -8
done
