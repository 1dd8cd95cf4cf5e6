# Round 5, GPU job 11: HBM counters of single tile-GEMM launches (is FETCH_SIZE x 2 right for global_load_lds?); TTS loop at 1024 rows
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5k}
mkdir -p $O
cd $R
for arm in 1 0; do
  S2S_TILE5=$arm timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf$arm -o f -- python tools/tile5_pmc.py > /dev/null 2> $O/pf$arm.err
  S2S_TILE5=$arm timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw$arm -o w -- python tools/tile5_pmc.py > /dev/null 2> $O/pw$arm.err
  echo "== S2S_TILE5=$arm"; python tools/tile5_pmc.py --read $O/pf$arm $O/pw$arm | tee $O/tile_pmc_arm$arm.txt
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
for cfg in "--tts-merge 8 --tts-slots 512" "--tts-merge 16 --tts-slots 1024"; do
  timeout 300 python bench.py --serial $cfg --steps 16 --warmup 16 --no-cpu-baseline --no-boundary --breakdown > "$O/serial_$(echo $cfg | tr -d ' -').json" 2> "$O/serial_$(echo $cfg | tr -d ' -').err"
  grep breakdown "$O/serial_$(echo $cfg | tr -d ' -').err" | head -12
  python -c "
import json,sys; d=json.load(open('$O/serial_$(echo $cfg | tr -d ' -').json')); print('$cfg', 'ms/step', d['ms_per_step'], 'xrt', d['xrt'])"
done
