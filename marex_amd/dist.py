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


# --------------------------------------------------------------------------------------
# one pass of the hot path over a rank's shards + the scalar collectives (bench.py and the tests share this code)
# --------------------------------------------------------------------------------------
SUMMARY_KEYS = ("n_ocean", "invalid_total", "invalid_cells", "n_extreme", "thr_too_low", "thr_too_high")


def shard_step(hot, shards, xs, dcal, *, W: int = 15, S: int = 21, bins, q: float, wd: int, ws: int, nx: int, workspace=None,
               detrend=None):
    """validation + anomaly + thresholds + mask for every shard of this rank (``xs[i]`` = resident ``[T, cells_in]`` input
    of ``shards[i]``).  ``detrend=(model, pmodel)`` switches the anomaly stage from ``shifting_baseline`` to
    ``detrend_fixed_baseline`` (detect.py:2400-2462).  Returns ``(result of the last shard, local, mx)``: ``local`` int64[6] in
    ``SUMMARY_KEYS`` order and ``mx`` int64[1] (largest per-cell invalid count) on the engine's device, not yet reduced
    over ranks."""
    import torch

    local = torch.zeros(6, dtype=torch.int64, device=hot.device)
    mx = torch.zeros(1, dtype=torch.int64, device=hot.device)
    r = None
    for sh, x in zip(shards, xs):
        own = sh.own_cell_slice()
        rows = (sh.own0 - sh.in0, sh.own1 - sh.in0) if sh.gridded else None
        ny_s, nx_s = (sh.ny_in, nx) if sh.gridded else (0, sh.cells_in)
        if detrend is None:
            r = hot.shifting_hobday(x, dcal, W=W, S=S, bins=bins, q=q, wd=wd, ws=ws, ny=ny_s, nx=nx_s, own_rows=rows,
                                    workspace=workspace)
        else:
            f = hot.detrend_fixed_baseline(x, detrend[0], detrend[1], True, dcal, None, wsp=workspace)
            h = hot.hobday_approx(f["out"], dcal, bins, q, wd, ws, ny_s, nx_s, rows=rows, cells=(own.start, own.stop), wsp=workspace)
            r = {"dat_anomaly": f["out"], "mask": f["mask"], "invalid_count": f["invalid_count"], "thr_doy_major": h["thr_doy_major"],
                 "stats_dev": h["stats_dev"], "extreme_events": h["extreme"], "n_true": h["n_true"], "path": h["path"]}
        vs = hot.validation_summary(r["mask"], r["invalid_count"], (own.start, own.stop), workspace)  # a3 verdict
        st = r["stats_dev"]
        local[0:3] += vs[0:3]
        local[3:4] += r["n_true"]
        local[4:6] += st[2:4]
        mx = torch.maximum(mx, vs[3:4])
    return r, local, mx


def allreduce_step(local, mx, host_collectives: bool = False):
    """Sum / max the per-rank scalars over the process group (RCCL on GPUs; ``host_collectives``: gloo reduces host tensors)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        if host_collectives:
            local, mx = local.cpu(), mx.cpu()
        dist.all_reduce(local, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    return local, mx


def gather_owned_cells(t_local, all_shards: List[Shard], rank: int, host_collectives: bool = False):
    """All-gather a per-shard array whose LAST axis is the rank's OWNED cells (thresholds ``[366, own]``, ``mask [own]``)
    into the global array on every rank (optional gather of ``thresholds`` / ``mask``).  Shards may own
    different numbers of cells: parts are padded to the largest and trimmed after the collective."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return t_local
    world = dist.get_world_size()
    counts = [s.cells_own for s in all_shards]
    m = max(counts)
    pad = torch.zeros(t_local.shape[:-1] + (m,), dtype=t_local.dtype, device=t_local.device)
    pad[..., : counts[rank]] = t_local
    if host_collectives:
        pad = pad.cpu()
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous())
    return torch.cat([p[..., : counts[r]] for r, p in enumerate(parts)], dim=-1)
