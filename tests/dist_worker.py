"""Worker for tests/test_dist_cpu.py: world-size-2 gloo run of the multi-GPU host path (sharding + survivor gather).
No GPU and no scoring here: survivors are deterministic stand-ins keyed by (region, position, strand)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from mipgen_amd import capi, dist as mdist  # noqa: E402


def fake_survivors(region: int, n_pos: int) -> np.ndarray:
    out = np.zeros(2 * n_pos, dtype=capi.SURVIVOR_DTYPE)
    k = np.arange(2 * n_pos)
    out["cand_index"] = region * 1_000_000 + k * 7
    out["score"] = np.sin(region * 0.37 + k * 0.011)
    out["record"] = (region << 32) + k
    return out


def main() -> None:
    out_path = sys.argv[1]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(5)
    n_pos = rng.integers(50, 400, size=23).tolist()              # ragged regions
    weights = [p * 1026 for p in n_pos]
    shards = mdist.shard_regions(weights, world)
    lo, hi = shards[rank]
    local = np.concatenate([fake_survivors(r, n_pos[r]) for r in range(lo, hi)]) if hi > lo else np.zeros(0, dtype=capi.SURVIVOR_DTYPE)
    emitted_local = sum(weights[lo:hi])
    off, total = mdist.exclusive_offsets(emitted_local)
    gathered = mdist.gather_to_rank0(local)
    if rank == 0:
        expect = np.concatenate([fake_survivors(r, n_pos[r]) for r in range(len(n_pos))])
        ok = gathered.shape == expect.shape and all(np.array_equal(gathered[f], expect[f]) for f in expect.dtype.names)
        json.dump({"ok": bool(ok), "shards": shards, "total": total, "expected_total": sum(weights), "n": int(gathered.shape[0])}, open(out_path, "w"))
    else:
        assert off == sum(weights[:lo])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
