set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" > $O/prio.log 2>&1
for cfg in "3 0" "2 0" "1 0" "3 -1"; do
set -- $cfg
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe-sessions 800,950 --pool-lanes $1 --stt-priority $2 --prof-steps 0 > $O/bench_l$1_p$2.json 2> $O/bench_l$1_p$2.err; tail -2 $O/bench_l$1_p$2.err
done
cat $O/prio.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3l/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e); continue
    b = d.get("boundary", {})
    print(f, "xrt", d["xrt"])
    for r in b.get("probes", []):
        p = r.get("pool", {})
        print("   ", {k: r.get(k) for k in ("sessions", "p50_stt_ms", "p95_stt_ms", "p50_ttfa_ms", "p95_ttfa_ms")}, "stt batches", p.get("stt_batches"), "busy", p.get("stt_busy_s"), "rounds", p.get("tts_rounds"), p.get("tts_row_rounds"))
PY
