set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n
mkdir -p $O
cd $R
nproc > $O/host.log; lscpu | grep -E "Model name|MHz" >> $O/host.log
for g in 5 0.5; do
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe-sessions 600,800 --gil-switch-ms $g --prof-steps 0 > $O/bench_g$g.json 2> $O/bench_g$g.err; tail -2 $O/bench_g$g.err
done
cat $O/host.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3n/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e); continue
    b = d.get("boundary", {})
    print(f, "xrt", d["xrt"])
    for r in b.get("probes", []):
        p = r.get("pool", {})
        print("   ", {k: r.get(k) for k in ("sessions", "p50_stt_ms", "p95_stt_ms", "p50_ttfa_ms", "p95_ttfa_ms")}, "stt batches", p.get("stt_batches"), "busy", p.get("stt_busy_s"), "rounds", p.get("tts_rounds"), p.get("tts_row_rounds"), "cores", p.get("host_cores_busy"))
PY
