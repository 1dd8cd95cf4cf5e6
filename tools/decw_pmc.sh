# counters of one wide projection (GPU box): HBM-side fetch bytes and L2 hit rate per launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4pmc; mkdir -p $O; cd $R
for F in 0 1 2 3; do S2S_DECW_FLAGS=$F python tools/decw_one.py 512 4096 2048 4 1; S2S_DECW_FLAGS=$F python tools/decw_one.py 512 4096 2048 8 1; done 2>&1 | grep -v amdgpu.ids
for F in 0 1; do S2S_DECW_FLAGS=$F python tools/decw_one.py 512 2048 1024 4 1; S2S_DECW_FLAGS=$F python tools/decw_one.py 512 6144 2048 4 1;  done 2>&1 | grep -v amdgpu.ids
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python tools/decw_one.py 512 4096 2048 4 1 > /dev/null 2> $O/fetch.err
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/hit -o h -- python tools/decw_one.py 512 4096 2048 4 1 > /dev/null 2> $O/hit.err
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/ea -o e -- python tools/decw_one.py 512 4096 2048 4 1 > /dev/null 2> $O/ea.err
python - <<'PY'
import csv, glob, collections
for d in ("fetch", "hit", "ea"):
    acc = collections.defaultdict(list)
    for path in glob.glob(f"gpurun_out/r4pmc/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if "gemm_decw" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, "launches", len(v), "mean per launch", sum(v) / len(v), "min", min(v), "max", max(v))
PY
tail -3 $O/*.err
find $O -name "*.csv" -delete
