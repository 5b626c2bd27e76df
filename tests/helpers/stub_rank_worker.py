"""Stand-in for one bench.py rank in the CPU test of bench.launch_ranks: no GPU, no HIP library.  It does what a rank
does around the hot path - reads the launcher's environment, joins a gloo group, takes its SURVEY 8(e) shard of the
sequences, gathers (fake) trajectory rows, agrees on the max-over-ranks clock - and rank 0 prints ONE JSON line."""
import argparse
import datetime
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def fake_rows(seed, n):
    rng = np.random.default_rng(seed)
    rows = np.zeros((n, 8))
    rows[:, 0] = 1000.0 + 0.1 * np.arange(n)
    rows[:, 1:4] = np.cumsum(rng.normal(0, 0.1, (n, 3)), axis=0)
    q = rng.normal(0, 1, (n, 4))
    rows[:, 4:8] = q / np.linalg.norm(q, axis=1, keepdims=True)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seqs-per-gpu", type=int, default=1)
    ap.add_argument("--seed-base", type=int, default=1000)
    ap.add_argument("--fail-rank", type=int, default=-1, help="this rank exits 7 before the gather")
    a = ap.parse_args()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        assert k in os.environ, f"launcher did not set {k}"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == a.gpus and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    if rank == a.fail_rank:
        sys.exit(7)
    import torch
    import torch.distributed as dist
    import ptudes_lab_amd  # noqa: F401
    from ptudes_lab_amd import parallel
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
    S, T = a.seqs_per_gpu, a.steps + a.warmup
    mine = parallel.shard_sequences(world * S, rank, world)
    assert mine == [rank + world * j for j in range(S)]
    rows = torch.zeros((S, T, 8), dtype=torch.float64)
    for j, s in enumerate(mine):
        rows[j] = torch.from_numpy(fake_rows(parallel.sequence_seed(s, a.seed_base), T))
    got = parallel.gather_trajectories(rows, [T] * S, dist)
    dt = parallel.max_over_ranks(0.01 * (rank + 1), dist)
    ok = all(np.array_equal(got[(r, j)], fake_rows(a.seed_base + r + world * j, T)) for r in range(world) for j in range(S))
    print(f"rank {rank}: not the JSON line", file=sys.stderr)
    if rank != 0:
        print("a non-zero rank's stdout must not reach the launcher's stdout")
    else:
        print(json.dumps({"metric": "stub", "value": a.steps * S * world / dt, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "gathered_trajectories": {"sequences": len(got), "ok": bool(ok)}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
