set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
S2S_TILE4_OFF=1 timeout 200 python tools/proto/tile4_check.py > gpurun_out/r3i/tile3.log 2>&1
S2S_TILE4=1 timeout 200 python tools/proto/tile4_check.py > gpurun_out/r3i/tile4_1.log 2>&1
S2S_TILE4=2 timeout 200 python tools/proto/tile4_check.py > gpurun_out/r3i/tile4_2.log 2>&1
tail -8 gpurun_out/r3i/tile3.log; tail -14 gpurun_out/r3i/tile4_1.log; tail -14 gpurun_out/r3i/tile4_2.log
