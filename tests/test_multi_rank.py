"""world_size-2 gloo run of the bench's cross-rank logic on CPU: sessions are sharded across ranks with no data-path
collective; the only exchange is barrier + MAX-reduce of the step time, and the whole-job value sums per-rank work."""

import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo_aggregation(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from bench import shard_sessions, reduce_step_time, sum_over_ranks
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        rank, world = dist.get_rank(), dist.get_world_size()
        ids = shard_sessions(per_gpu=5, rank=rank)
        t = reduce_step_time(0.1 * (rank + 1))
        total = sum_over_ranks(7 + rank)
        gathered = [None] * world
        dist.all_gather_object(gathered, ids)
        if rank == 0:
            print(json.dumps({{"t": t, "ids": gathered, "total": total}}))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json

    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert abs(r["t"] - 0.2) < 1e-9                       # MAX over ranks
    assert r["total"] == 15                               # per-rank sustained counts are summed
    flat = [i for ids in r["ids"] for i in ids]
    assert sorted(flat) == list(range(10)) and len(set(flat)) == 10   # disjoint session shards, weak scaling


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` (the form the driver uses for N = 1) must start one process per GPU itself, rendezvous over gloo
    and print ONE line with n_gpus = 2.  Dry run: the ranks do the whole rendezvous / barrier / MAX-reduce, no GPU work."""
    import json

    env = dict(os.environ, S2S_BENCH_DRY_RUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sessions", "3"], capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and abs(r["max_step_s"] - 0.2) < 1e-9 and r["sessions"] == [[0, 1, 2], [3, 4, 5]]
    assert r["sustained_sum"] == 300                      # the whole-job sustained figure is the SUM of the ranks' open-loop figures
