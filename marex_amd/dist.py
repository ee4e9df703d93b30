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
#: thr_unresolved: outputs the list threshold kernel gave up on (marex_thr_stats.n_unresolved) -- must be 0, callers raise otherwise
SUMMARY_KEYS = ("n_ocean", "invalid_total", "invalid_cells", "n_extreme", "thr_too_low", "thr_too_high", "thr_unresolved")


class EngineSet:
    """``n`` engines on ONE device, each with a HIP stream, a calendar copy and an output workspace of its own: the shards of
    a rank go round-robin over them, so the kernels of neighbouring shards overlap (the HBM-heavy anomaly kernel of one band
    runs beside the latency-bound threshold kernel of another; measured on the 100-yr field, round 3: 141.0 ms with one stream,
    129.5-133.5 with two, 128.8-129.8 with three).
    ``shard_step`` takes an ``EngineSet`` wherever it takes a single engine; results are those of one engine, bit for bit
    (``tests/test_gpu_sharding.py``)."""

    def __init__(self, device, n: int = 2, factory=None):
        if n < 1:
            raise ValueError("EngineSet needs at least one engine")
        if factory is None:
            from .engine import HotPath

            factory = lambda: HotPath(device, own_stream=True)  # noqa: E731
        self.engines = [factory() for _ in range(int(n))]
        self.device = self.engines[0].device
        self.workspaces = [{} for _ in self.engines]
        self._dcals = {}

    def __len__(self) -> int:
        return len(self.engines)

    @property
    def ctx(self):
        """The FIRST engine's context (timers of that engine, debug counters).  Options set through it reach only that engine:
        use :meth:`set_option` / :meth:`options`, which fan out to all of them."""
        return self.engines[0].ctx

    def set_option(self, name: str, value) -> None:
        """``ctx.set_option`` on every engine (``None`` clears it): a forced kernel path holds for every band of the set."""
        for e in self.engines:
            e.ctx.set_option(name, value)

    def options(self, **opts):
        """``with es.options(NAME=v, ...):`` -- the options on every engine, previous values restored on exit."""
        from contextlib import ExitStack

        stack = ExitStack()
        for e in self.engines:
            stack.enter_context(e.ctx.options(**opts))
        return stack

    def set_hobday_path(self, path) -> None:
        for e in self.engines:
            e.hobday_path = path

    def calendars(self, cal):
        """Device copies of a calendar plan, one per engine (uploaded once per plan)."""
        held = self._dcals.get("plan")
        if held is None or held[0] is not cal:  # the plan object is kept alive with its copies: an id() alone can be reused
            self._dcals = {"plan": (cal, [e.upload_calendar(cal) for e in self.engines])}
        return self._dcals["plan"][1]

    def sync(self) -> None:
        for e in self.engines:
            if e.stream is not None:
                e.stream.synchronize()
            e.sync()

    def timing_enable(self, on: bool = True) -> None:
        for e in self.engines:
            e.ctx.timing_enable(on)

    def timing_reset(self) -> None:
        for e in self.engines:
            e.ctx.timing_reset()

    def timing_get(self, kernel: str):
        """(total ms, launches) of a kernel family summed over the engines (HIP events on each engine's own stream)."""
        tot, n = 0.0, 0
        for e in self.engines:
            a, b = e.ctx.timing_get(kernel)
            tot, n = tot + a, n + b
        return tot, n


def _one_shard(hot, sh, x, dcal, *, W, S, bins, q, wd, ws, nx, workspace, detrend):
    """The four stages of one shard on one engine; returns (result dict, int64[7] partial summary, int64[1] max invalid)."""
    import torch

    own = sh.own_cell_slice()
    rows = (sh.own0 - sh.in0, sh.own1 - sh.in0) if sh.gridded else None
    ny_s, nx_s = (sh.ny_in, nx) if sh.gridded else (0, sh.cells_in)
    if detrend is None:
        r = hot.shifting_hobday(x, dcal, W=W, S=S, bins=bins, q=q, wd=wd, ws=ws, ny=ny_s, nx=nx_s, own_rows=rows,
                                workspace=workspace)
    else:
        tails_bins = bins if hot.tails_plan(dcal, bins, q, wd, ws, x.shape[1]) is not None else None
        f = hot.detrend_fixed_baseline(x, detrend[0], detrend[1], True, dcal, None, wsp=workspace, tails_bins=tails_bins)
        h = hot.hobday_approx(f["out"], dcal, bins, q, wd, ws, ny_s, nx_s, rows=rows, cells=(own.start, own.stop), wsp=workspace,
                              tails=f.get("tails"))
        r = {"dat_anomaly": f["out"], "mask": f["mask"], "invalid_count": f["invalid_count"], "thr_doy_major": h["thr_doy_major"],
             "stats_dev": h["stats_dev"], "extreme_events": h["extreme"], "n_true": h["n_true"], "path": h["path"]}
    vs = hot.validation_summary(r["mask"], r["invalid_count"], (own.start, own.stop), workspace)  # a3 verdict
    part = torch.cat([vs[0:3], r["n_true"].to(torch.int64).reshape(1), r["stats_dev"][2:5].to(torch.int64)])
    return r, part, vs[3:4].clone()


def shard_step(hot, shards, xs, dcal, *, W: int = 15, S: int = 21, bins, q: float, wd: int, ws: int, nx: int, workspace=None,
               detrend=None):
    """validation + anomaly + thresholds + mask for every shard of this rank (``xs[i]`` = resident ``[T, cells_in]`` input
    of ``shards[i]``).  ``detrend=(model, pmodel)`` switches the anomaly stage from ``shifting_baseline`` to
    ``detrend_fixed_baseline`` (detect.py:2400-2462).  ``hot``: one engine (shards one after the other on the current stream)
    or an :class:`EngineSet` (shards round-robin over its engines and streams; ``dcal`` may then be the host ``CalendarPlan``
    and ``workspace`` is ignored: every engine writes into its own).  Returns ``(result of the last shard, local, mx)``:
    ``local`` int64[7] in ``SUMMARY_KEYS`` order and ``mx`` int64[1] (largest per-cell invalid count) on the device, not yet
    reduced over ranks and valid on the CALLER's current stream."""
    import torch

    kw = dict(W=W, S=S, bins=bins, q=q, wd=wd, ws=ws, nx=nx, detrend=detrend)
    if not isinstance(hot, EngineSet):
        local = torch.zeros(len(SUMMARY_KEYS), dtype=torch.int64, device=hot.device)
        mx = torch.zeros(1, dtype=torch.int64, device=hot.device)
        r = None
        for sh, x in zip(shards, xs):
            r, part, m = _one_shard(hot, sh, x, dcal, workspace=workspace, **kw)
            local += part
            mx = torch.maximum(mx, m)
        return r, local, mx

    es = hot
    dcals = es.calendars(dcal.plan if hasattr(dcal, "plan") else dcal)
    main = torch.cuda.current_stream(es.device)
    fork = torch.cuda.Event()
    fork.record(main)  # the engines' streams start after whatever the caller queued (inputs, the previous step)
    parts = [[] for _ in es.engines]
    r = None
    for i, (sh, x) in enumerate(zip(shards, xs)):
        k = i % len(es)
        e = es.engines[k]
        with torch.cuda.stream(e.stream):
            if i < len(es):
                e.stream.wait_event(fork)
            r, part, m = _one_shard(e, sh, x, dcals[k], workspace=es.workspaces[k], **kw)
            part.record_stream(main)  # allocated on the engine's stream, read on the caller's
            m.record_stream(main)
            parts[k].append((part, m))
    local = torch.zeros(len(SUMMARY_KEYS), dtype=torch.int64, device=es.device)
    mx = torch.zeros(1, dtype=torch.int64, device=es.device)
    for k, e in enumerate(es.engines):
        if not parts[k]:
            continue
        done = torch.cuda.Event()
        done.record(e.stream)
        main.wait_event(done)  # join: the caller's stream sees every engine's partial sums
        for part, m in parts[k]:
            local += part
            mx = torch.maximum(mx, m)
    return r, local, mx


def broadcast_tables(tables: Optional[Dict[str, object]], src: int = 0, device=None, host_collectives: bool = False,
                     force: bool = False) -> Dict[str, object]:
    """The host-built tables of a run -- calendar plan arrays, bin edges / centres, detrend model and pseudo-inverse -- from rank
    ``src`` to every rank ("ncclBroadcast of calendar tables + bin edges/centres + pinv rows"): one object
    broadcast of the manifest (names, dtypes, shapes, scalars), one byte broadcast of the packed arrays (a device tensor over
    RCCL; ``host_collectives``: a host tensor over gloo).  ``tables`` maps names to NumPy arrays or plain scalars and is only
    read on ``src``; every rank gets the same dict back, ``src`` included (its own arrays go through the same packing, so all
    ranks work from identical bytes).  Without a process group -- or with a single rank, unless ``force`` asks for the collectives
    all the same (tests) -- the input is returned as it is."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return dict(tables or {})
    rank = dist.get_rank()
    manifest, chunks, off = None, [], 0
    if rank == src:
        manifest = {"arrays": [], "scalars": {}}
        for name, v in (tables or {}).items():
            # arrays of plain numbers travel packed; object arrays (cftime axes) and 0-d arrays go with the pickled manifest
            if isinstance(v, np.ndarray) and v.dtype != object and v.ndim >= 1:
                a = np.ascontiguousarray(v)
                manifest["arrays"].append((name, a.dtype.str, tuple(a.shape), off, a.nbytes))
                chunks.append(a.view(np.uint8).reshape(-1) if a.dtype != np.bool_ else a.astype(np.uint8).reshape(-1))
                pad = (-a.nbytes) % 16
                if pad:
                    chunks.append(np.zeros(pad, dtype=np.uint8))
                off += a.nbytes + pad
            else:
                manifest["scalars"][name] = v
        manifest["nbytes"] = off
    box = [manifest]
    dist.broadcast_object_list(box, src=src, device=None if host_collectives else device)
    manifest = box[0]
    n = int(manifest["nbytes"])
    dev = "cpu" if host_collectives else device
    if rank == src:
        payload = torch.from_numpy(np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)).to(dev)
    else:
        payload = torch.empty((n,), dtype=torch.uint8, device=dev)
    if n:
        dist.broadcast(payload, src=src)
    raw = payload.cpu().numpy()
    out: Dict[str, object] = dict(manifest["scalars"])
    for name, dt, shape, o, nb in manifest["arrays"]:
        out[name] = raw[o:o + nb].view(np.dtype(dt)).reshape(shape).copy()
    return out


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
