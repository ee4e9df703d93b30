"""Spatial sharding of the hot path over the GPUs of one node.

Every stage is per-cell along time except the ``ws x ws`` histogram pooling, which couples a cell to
its ``ws//2`` neighbours (lon periodic, lat truncated).  Gridded fields are therefore cut into
contiguous LATITUDE BANDS (lon periodicity stays inside a shard) that are ingested with ``ws//2``
overlap rows per interior side; thresholds of the overlap rows are computed but discarded, so no
GPU<->GPU halo exchange exists.  Unstructured fields are cut into contiguous cell ranges.

The only collectives are all-reduces of a handful of int64 scalars (validation verdict, warning
counters, number of extreme events) through ``torch.distributed`` -- backend ``nccl`` (= RCCL over
xGMI) on GPUs, ``gloo`` in the CPU tests.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    ny_global: int  # 0 for unstructured
    nx: int  # cells per row (or total cells when unstructured)
    own0: int  # first owned row (or cell)
    own1: int  # one past the last owned row (or cell)
    in0: int  # first ingested row (>= own0 - halo)
    in1: int  # one past the last ingested row

    @property
    def gridded(self) -> bool:
        return self.ny_global > 0

    @property
    def ny_in(self) -> int:
        return self.in1 - self.in0 if self.gridded else 0

    @property
    def cells_in(self) -> int:
        return (self.in1 - self.in0) * self.nx if self.gridded else self.in1 - self.in0

    @property
    def cells_own(self) -> int:
        return (self.own1 - self.own0) * self.nx if self.gridded else self.own1 - self.own0

    @property
    def cell_base(self) -> int:
        """Global id of the first ingested cell."""
        return self.in0 * self.nx if self.gridded else self.in0

    def own_cell_slice(self) -> slice:
        """Slice of the owned cells inside the ingested band (cells are the last, contiguous axis)."""
        if self.gridded:
            return slice((self.own0 - self.in0) * self.nx, (self.own1 - self.in0) * self.nx)
        return slice(self.own0 - self.in0, self.own1 - self.in0)


def _split(n: int, parts: int) -> List[int]:
    base, rem = divmod(n, parts)
    edges = [0]
    for r in range(parts):
        edges.append(edges[-1] + base + (1 if r < rem else 0))
    return edges


def plan_shards(ny: int, nx: int, world: int, halo: int) -> List[Shard]:
    """Latitude bands (``ny > 0``) or cell ranges (``ny == 0``, ``nx`` = number of cells, halo ignored)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    if ny > 0:
        if world > ny:
            raise ValueError("more shards than latitude rows")
        e = _split(ny, world)
        return [
            Shard(r, world, ny, nx, e[r], e[r + 1], max(0, e[r] - halo), min(ny, e[r + 1] + halo)) for r in range(world)
        ]
    e = _split(nx, world)
    return [Shard(r, world, 0, nx, e[r], e[r + 1], e[r], e[r + 1]) for r in range(world)]


def stitch_cells(parts: List[np.ndarray], shards: List[Shard]) -> np.ndarray:
    """Concatenate the owned part of per-shard arrays whose LAST axis is the (ingested) cell axis."""
    return np.concatenate([p[..., s.own_cell_slice()] for p, s in zip(parts, shards)], axis=-1)


def allreduce_summary(local: Dict[str, int], device=None) -> Dict[str, int]:
    """Combine the per-shard scalars.  Keys ending in ``_max`` are max-reduced, everything else summed.

    Without an initialised process group (single process) the input is returned unchanged.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(local)
    keys = sorted(local)
    sums = [k for k in keys if not k.endswith("_max")]
    maxs = [k for k in keys if k.endswith("_max")]
    out: Dict[str, int] = {}
    for group, op in ((sums, dist.ReduceOp.SUM), (maxs, dist.ReduceOp.MAX)):
        if not group:
            continue
        t = torch.tensor([int(local[k]) for k in group], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=op)
        for k, v in zip(group, t.tolist()):
            out[k] = int(v)
    return out
