set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 300 python tools/dec2_phases.py 64 > gpurun_out/r3b/phases64.log 2>&1
timeout 300 python tools/dec2_phases.py 128 > gpurun_out/r3b/phases128.log 2>&1
cat gpurun_out/r3b/phases64.log gpurun_out/r3b/phases128.log
