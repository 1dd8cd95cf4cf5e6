# Round 5, GPU job 22: MFMA-pipe utilisation and LDS bank conflicts of the tile GEMM launches (separate --pmc passes, no trace domains)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5v}
mkdir -p $O
cd $R
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o a -- python tools/tile5_pmc.py > /dev/null 2> $O/p1.err; tail -2 $O/p1.err
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -o b -- python tools/tile5_pmc.py > /dev/null 2> $O/p2.err; tail -2 $O/p2.err
python tools/pmc_dump.py $O/p1 gemm_tile enc.qkv enc.o enc.fc1 enc.fc2 cross.kv | tee $O/tile_mfma_busy.txt
python tools/pmc_dump.py $O/p2 gemm_tile enc.qkv enc.o enc.fc1 enc.fc2 cross.kv | tee $O/tile_lds_conflicts.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
