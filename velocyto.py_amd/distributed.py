"""Cell-sharded execution over the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference is single-process: there is nothing to mirror, so the
partitioning follows the data: a rank owns a contiguous block of CELLS (one slab of the
cells-major matrices).  Exchange steps of the hot path:

  A  pooling     none   -- S_sz/U_sz are replicated inputs (any cell can be a neighbour)
  B  fit_slope   all-reduce(sum) of the per-gene moments, 3*G fp64 (720 KB at 30k genes)
  C  velocity    none   -- row-local
  D  colDeltaCor all-gather of the Sx_sz shards (every rank needs all of `e`), then the
                 all-gather of the compact correlation rows (C x nrndm) for whoever wants them whole

xGMI is point-to-point (7 links per GPU): the big all-gather moves 1/world of the matrix over
each link once, so it is issued as ONE collective on the full shard (largest possible message).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


import os

# VCY_FORCE_COLLECTIVES=1 issues the collectives even at world size 1 (single-GPU smoke test of the RCCL path)
FORCE = os.environ.get("VCY_FORCE_COLLECTIVES", "0") == "1"


def active() -> bool:
    """True when the exchange steps must be issued (more than one rank, or forced for testing)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced split of range(n): the first n % world ranks get one extra row."""
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_shard_bounds(n: int, world_size: int) -> List[Tuple[int, int]]:
    return [shard_bounds(n, world_size, r) for r in range(world_size)]


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_gather_rows(local: torch.Tensor, n_total: int, out: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """Reassemble a row-sharded (shard_bounds) tensor: local (n_loc, ...) -> (n_total, ...).
    Equal shards use one all_gather_into_tensor straight into `out`; ragged shards pad to the
    largest shard and trim."""
    rank, ws = world()
    if not active():
        if out is None:
            return local
        out.copy_(local)
        return out
    bounds = all_shard_bounds(n_total, ws)
    assert local.shape[0] == bounds[rank][1] - bounds[rank][0], "local shard does not match shard_bounds"
    tail = tuple(local.shape[1:])
    if out is None:
        out = torch.empty((n_total,) + tail, dtype=local.dtype, device=local.device)
    sizes = [b - a for a, b in bounds]
    if len(set(sizes)) == 1 and local.is_contiguous() and out.is_contiguous() and dist.get_backend(group) != "gloo":
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(sizes)
    buf = torch.zeros((mx,) + tail, dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    pieces = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(pieces, buf, group=group)
    for (a, b), p in zip(bounds, pieces):
        out[a:b] = p[: b - a]
    return out
