cd $GRAFT_REPO_ROOT; O=gpurun_out/r4e; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -q --maxfail=20 > $O/tests_gpu.log 2>&1; tail -12 $O/tests_gpu.log
timeout 200 python bench.py --serial --tts-merge 1 --steps 2 --warmup 1 --no-boundary --no-cpu-baseline --breakdown > $O/serial64.json 2> $O/serial64.err; grep breakdown $O/serial64.err
timeout 300 python bench.py --serial --tts-merge 4 --steps 4 --warmup 4 --no-boundary --no-cpu-baseline --breakdown > $O/serial256.json 2> $O/serial256.err; grep breakdown $O/serial256.err
timeout 300 python bench.py --serial --tts-merge 8 --steps 8 --warmup 8 --no-boundary --no-cpu-baseline --breakdown > $O/serial512.json 2> $O/serial512.err; grep breakdown $O/serial512.err
python - <<'PY'
import json
for n in ("serial64", "serial256", "serial512"):
    try:
        d = json.load(open(f"gpurun_out/r4e/{n}.json")); print(n, "xrt", d["xrt"], "ms/step", d["ms_per_step"], d["config"]["stages_ms_per_step"])
    except Exception as e:
        print(n, "ERR", e)
PY
tail -5 $O/serial256.err
