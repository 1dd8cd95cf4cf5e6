# Round-6 job 6: TDT loop in isolation (fused / unfused / graph), then the whole GPU suite + smoke at HEAD
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r6f}
mkdir -p $O
cd $R
(timeout 200 python tools/bench_tdt.py; S2S_TDT_FUSED=0 timeout 200 python tools/bench_tdt.py; timeout 200 python tools/bench_tdt.py; S2S_TDT_GRAPH=1 timeout 200 python tools/bench_tdt.py; S2S_TDT_GRAPH=1 S2S_TDT_FUSED=0 timeout 200 python tools/bench_tdt.py) 2>&1 | grep -v amdgpu.ids > $O/bench_tdt.txt; cat $O/bench_tdt.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -4 $O/tests_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
