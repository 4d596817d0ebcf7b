"""Multi-GPU plumbing for the hot path: one process per GPU, regions sharded by dense-grid size, one gather of the
per-position survivors to rank 0.

Scoring needs no data-path collective: everything the path reads is per-region or read-only
(/root/reference/mipgen.cpp:522-524 clears the per-region maps).  Picking is sequential in region order - the rand()
stream and the used-arm sets persist across regions (/root/reference/mipgen.cpp:97,1863) - so rank 0 receives every
rank's condensed survivors (2 per scan-start position) and runs the pick stage itself.

`torch.distributed` is used as transport only: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def region_cost(dense_candidates: Sequence[int], n_pos: Sequence[int], n_sizes: Sequence[int], n_e: int, n_l: int, inc: int, sum_range: int,
                svr: bool) -> np.ndarray:
    """Relative device time per region = the shard weights (the same rule as mipgen_amd/host/design.cpp: region_cost): the dense-grid
    candidates and, for the dense SVR scorer, the factor-table entries it builds per support vector at the kernel's instruction budget (~47 VALU
    per table entry against ~2.7 per candidate: mipgen_amd/csrc/accel_tiles.hip: build_svr_tiles) - exons with few capture sizes cost more per
    candidate than their dense-grid size says.  sum_range = max arm sum - min arm sum."""
    cand = np.asarray(dense_candidates, dtype=np.float64)
    if not svr:
        return cand
    K = np.asarray(n_sizes, dtype=np.float64)
    P = np.asarray(n_pos, dtype=np.float64)
    ssr = (np.minimum(K, 9) - 1) * inc + sum_range + 1
    runs = np.ceil(K / 9.0)
    ent = 2.0 * runs * (P * (n_e + n_l) + P * ssr)
    return 2.7 * cand + 47.0 * ent


def shard_regions(weights: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous region ranges [lo, hi) per rank, balanced by the given weights (dense-grid candidate counts or region_cost).
    Contiguity keeps the reference's region order, which the sequential pick stage on rank 0 needs."""
    n = len(weights)
    total = float(sum(weights))
    out: List[Tuple[int, int]] = []
    lo = 0
    acc = 0.0
    for r in range(world):
        target = total * (r + 1) / world
        hi = lo
        while hi < n and (acc + weights[hi] <= target or hi == lo) and (n - hi) > (world - 1 - r):
            acc += weights[hi]
            hi += 1
        if r == world - 1:
            hi = n
        out.append((lo, hi))
        lo = hi
    return out


def gather_to_rank0(local: np.ndarray, device: Optional[str] = None):
    """Gather a 1-D structured/plain numpy array from every rank to rank 0, preserving rank order.
    Returns the concatenation on rank 0, None elsewhere.  Uses one all_gather of sizes and one gather of padded
    byte buffers (direct peer->root transfers over xGMI with RCCL; no ring is needed for MB-sized payloads)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device(device) if device else torch.device("cpu")
    raw = np.ascontiguousarray(local).view(np.uint8).reshape(-1)
    t = torch.from_numpy(raw.copy()).to(dev)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes_i = [int(s.item()) for s in sizes]
    mx = max(max(sizes_i), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=dev)
    pad[: t.numel()] = t
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    parts = [bufs[r][: sizes_i[r]].cpu().numpy() for r in range(world)]
    return np.concatenate(parts).view(local.dtype)


def exclusive_offsets(local_count: int, device: Optional[str] = None) -> Tuple[int, int]:
    """(offset of this rank, global total) of a per-rank count - the prefix sums that number emitted candidates
    globally (mip_name = running all_mip_counter, /root/reference/mipgen.cpp:474,488,792)."""
    import torch
    import torch.distributed as dist

    dev = torch.device(device) if device else torch.device("cpu")
    world = dist.get_world_size()
    xs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(xs, torch.tensor([local_count], dtype=torch.int64, device=dev))
    vals = [int(x.item()) for x in xs]
    r = dist.get_rank()
    return sum(vals[:r]), sum(vals)
