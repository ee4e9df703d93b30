#!/usr/bin/env python3
"""Benchmark of the preprocess_data hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W          (plain command: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of validation + shifting-baseline anomaly + day-of-year thresholds + extreme mask
over one synthetic field that is already resident in HBM.  Metric (BASELINE.json): Mcells*timesteps/s
(input timesteps), whole job.  Default workload `cfg3` = the configuration the metric is quoted on: the
100-yr daily 1440x720 field (151 GB), resident as overlapped latitude bands (6 of 120 rows where the rank count
divides 6, else 8 of 90: whole rows of 30-row threshold tiles either way); N ranks take 1/N of the bands each
(strong scaling, N in {1,2,4,8}).  `--workload cfg2` = the 10-yr field, one 720-row band per rank (weak
scaling).  Bands are ingested with ws//2 overlap rows per interior side (marex_amd/dist.py); the collectives
are a broadcast of the host-built tables (calendar, bin edges / centres, detrend model) from rank 0 before the
timed region and an all-reduce of a few int64 scalars per step.  A rank with several bands runs them round-robin
over `--streams` engines with a HIP stream each (marex_amd.dist.EngineSet, default 3 when the workspaces fit): the product schedule.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md); 6290 GB/s measured float4 copy

WORKLOADS = {
    # BASELINE.json configs[1]: 10-yr daily x 1440x720, shifting_baseline (W=5: 10 yr of data cannot
    # hold the default W=15) + hobday_extreme p95, 5x5 pooling (gridded default)
    "cfg2": dict(start="2015-01-01", T=3652, ny=720, nx=1440, W=5, S=21, wd=11, ws=5, pct=95.0,
                 name="10yr-daily x 1440x720 (0.25deg), shifting_baseline(W=5,S=21)+hobday_extreme p95 (wd=11, ws=5)"),
    # 1/8 of BASELINE.json configs[2] (one of 8 latitude bands of the 100-yr field)
    "cfg3band": dict(start="1925-01-01", T=36500, ny=90, nx=1440, W=15, S=21, wd=11, ws=5, pct=95.0,
                     name="100yr-daily x 1440x90 band (1/8 of 0.25deg global), shifting_baseline(W=15,S=21)+hobday_extreme p95"),
    "tiny": dict(start="2015-01-01", T=3652, ny=48, nx=96, W=5, S=21, wd=11, ws=5, pct=95.0,
                 name="tiny plumbing case 10yr x 96x48"),
    # BASELINE.json configs[2], the north-star shape: the WHOLE 100-yr field (151 GB of input) stays resident as 8
    # overlapped latitude bands; every rank streams its 8/N bands through one reusable output workspace
    # (314 GB of input + output do not fit 288 GB at once).  Strong scaling over N in {1,2,4,8}.
    "cfg3": dict(start="1925-01-01", T=36500, ny=720, nx=1440, W=15, S=21, wd=11, ws=5, pct=95.0, bands=8,
                 name="100yr-daily x 1440x720 (0.25deg) in {bands} resident latitude bands, shifting_baseline(W=15,S=21)+hobday_extreme p95"),
    # BASELINE.json configs[3]: 30-yr daily x 2e6-cell unstructured mesh on 4 GPUs = 500 000 cells per GPU (weak scaling),
    # shifting_baseline + hobday_extreme p95, no spatial pooling (detect.py:1361-1385)
    "cfg4": dict(start="1995-01-01", T=10957, ny=0, nx=500_000, W=15, S=21, wd=11, ws=1, pct=95.0,
                 name="30yr-daily x 500000 cells per GPU of an unstructured mesh (2e6 cells on 4 GPUs), shifting_baseline(W=15,S=21)+hobday_extreme p95, no pooling"),
    # BASELINE.json configs[4]: the 100-yr field with detrend_fixed_baseline (orders 1, 2) + hobday_extreme p90
    "cfg5": dict(start="1925-01-01", T=36500, ny=720, nx=1440, W=None, S=21, wd=11, ws=5, pct=90.0, bands=8, detrend_orders=(1, 2),
                 name="100yr-daily x 1440x720 (0.25deg) in {bands} resident latitude bands, detrend_fixed_baseline(orders 1,2)+hobday_extreme p90"),
}


def algorithmic_bytes(T, T_out, C):
    """read 4T (x) + write 4T' (anomaly) + T' (mask) + 4*366 (thresholds) + 1 (mask) per cell."""
    return C * (4 * T + 5 * T_out + 4 * 366 + 1)


def _cpu_worker(args):
    """One host process: the NumPy oracle on its own 32x64 sub-grid (all timesteps) of the workload."""
    wl, seed, k = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from marex_amd import binning, calendar, synth
    from oracle import marex_oracle as orc

    ny, nx = 32, 64
    tm = calendar.daily_time_axis(wl["start"], wl["T"])
    tab = synth.make_tables(tm, ny, nx, seed + k)
    x = synth.synth_field(tab)
    cal = calendar.build_calendar(tm, window_year_baseline=wl["W"])
    bt = binning.hobday_bins()
    kw = {}
    if wl.get("detrend_orders"):
        model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), list(wl["detrend_orders"]), False)
        kw = dict(method_anomaly="detrend_fixed_baseline", model=model, pmodel=pmodel)
    gny, gnx, gws = (ny, nx, wl["ws"]) if wl["ny"] else (0, ny * nx, None)
    t0 = time.perf_counter()
    orc.validate_data_values(x)
    orc.preprocess_arrays(
        x, cal, ny=gny, nx=gnx, window_year_baseline=wl["W"] or 15, smooth_days_baseline=wl["S"],
        window_days_hobday=wl["wd"], window_spatial_hobday=gws, threshold_percentile=wl["pct"],
        edges=bt.edges, centres=bt.centres, **kw,
    )
    return time.perf_counter() - t0


def cpu_baseline(wl, seed):
    """Oracle (NumPy restatement of the reference path) on the host cores this job may use: one process per core, each on
    its own 32x64 sub-grid of the same workload, all timesteps; throughput = all sub-grids / wall time of the pool."""
    import multiprocessing as mp

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # CPU share of this job: a cgroup quota when there is one (v2 cpu.max, v1 cpu.cfs_quota_us), else 16 -- a one-GPU box is
    # granted a share of the host although sched_getaffinity lists all 256 cores (measured round 3: 64 workers delivered 28
    # Mcells*timesteps/s in aggregate, 16 workers 62).  MAREX_CPU_WORKERS overrides.
    quota = None
    for f, g in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if g is None:
                a, b = open(f).read().split()[:2]
                quota = None if a == "max" else int(a) / int(b)
            else:
                a, b = int(open(f).read()), int(open(g).read())
                quota = None if a <= 0 else a / b
            break
        except (OSError, ValueError):
            continue
    share = int(quota) if quota and quota >= 1 else 16
    cores = max(1, min(avail, share, 64))
    cores = int(os.environ.get("MAREX_CPU_WORKERS", cores))
    ny, nx = 32, 64
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        per = pool.map(_cpu_worker, [(wl, seed, k) for k in range(cores)])
    dt = time.perf_counter() - t0
    return {
        "value": wl["T"] * ny * nx * cores / 1e6 / dt,
        "unit": "Mcells*timesteps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"NumPy oracle, {cores} processes x one {ny}x{nx} sub-grid of the same {wl['T']}-day workload each "
                  f"({dt:.1f} s wall incl. process start; {max(per):.1f} s slowest worker; host reports {os.cpu_count()} cores, "
                  f"{avail} in the affinity mask, cgroup quota {quota if quota else 'none'}: {cores} workers; 64 workers on this "
                  f"box measured 28 Mcells*timesteps/s in aggregate, round 3)",
        # the reference's own figure for its Dask path (docs/modules/detect.rst:729-731): not measured here
        "reference_published": {"value": 7.9, "unit": "Mcells*timesteps/s", "cores": 255, "hardware": "unstated",
                                "source": "reference docs/modules/detect.rst:729-731 (BASELINE.md 1)"},
    }


def seasonal_extra(hot, shard, x, dcal, cal, step_kw, stationary_ms, amp=1.6, passes=3, wsp=None):
    """AFTER the timed region: the first band again with heteroscedastic noise added (amplitude follows the day of the year,
    so the p95 threshold of a cell swings over the year) -- the benchmark field has stationary noise, and the threshold
    kernel's band has to FOLLOW the thresholds; this line shows what a seasonal field costs.  Overwrites `x`."""
    import numpy as np
    import torch

    from marex_amd.dist import shard_step

    T = x.shape[0]
    g = torch.from_numpy((0.5 * (1 + np.sin(2 * np.pi * cal.doy / 365.25))).astype(np.float32)).to(hot.device)
    gen = torch.Generator(device=hot.device).manual_seed(1)
    for t0 in range(0, T, 2000):  # in slabs: no second field-sized temporary
        t1 = min(T, t0 + 2000)
        x[t0:t1] += amp * g[t0:t1, None] * torch.randn((t1 - t0, x.shape[1]), device=hot.device, generator=gen)
    wsp = {} if wsp is None else wsp
    hot.ctx.timing_enable(True)
    for k in range(passes + 1):
        if k == 1:
            hot.sync()
            hot.ctx.timing_reset()
        r, _, _ = shard_step(hot, [shard], [x], dcal, workspace=wsp, **step_kw)
    hot.sync()
    ms = {}
    for k in ("shifting", "tails", "thresholds", "mask"):
        tot, n = hot.ctx.timing_get(k)
        if n:
            ms[k] = tot / n
    thr = r["thr_doy_major"]  # [366, own cells]
    ocean = torch.isfinite(thr[0])
    swing = (thr[:, ocean].max(dim=0).values - thr[:, ocean].min(dim=0).values).median().item() if bool(ocean.any()) else 0.0
    return {"what": f"band 0 with N(0, ({amp} * (1 + sin(2 pi doy / 365.25)) / 2)^2) added: per-launch kernel ms, {passes} passes",
            "median_threshold_swing_K": swing, "kernel_ms": ms,
            "thresholds_ms_stationary": stationary_ms.get("thresholds"), "mask_ms_stationary": stationary_ms.get("mask"),
            "n_extreme": int(r["n_true"].item())}


def serial_extra(hot, shards, xs, dcal, step_kw, detrend, units, wsp, passes=2):
    """AFTER the timed region: the same step with every band on ONE engine and ONE stream, kernels one after the other --
    per-kernel HIP-event averages undisturbed by a neighbouring stream (the durations the rocprof summaries of a
    `--streams 1` run show), and what the multi-stream schedule of the timed region buys."""
    import time as _time

    import torch

    from marex_amd.dist import shard_step

    with torch.cuda.stream(hot.stream) if hot.stream is not None else _null():
        shard_step(hot, shards, xs, dcal, workspace=wsp, detrend=detrend, **step_kw)
        torch.cuda.synchronize()
        hot.ctx.timing_enable(True)
        hot.ctx.timing_reset()
        t0 = _time.perf_counter()
        for _ in range(passes):
            _, local, _ = shard_step(hot, shards, xs, dcal, workspace=wsp, detrend=detrend, **step_kw)
        torch.cuda.synchronize()
        dt = (_time.perf_counter() - t0) / passes
    ms = {}
    for k in ("shifting", "detrend", "fixed", "tails", "thresholds", "mask", "transpose"):
        tot, n = hot.ctx.timing_get(k)
        if n:
            ms[k] = tot / n
    return {"what": f"every band on one engine and one HIP stream, {passes} passes", "ms_per_step": dt * 1e3, "value": units / dt,
            "n_extreme": int(local[3].item()), "kernel_ms": ms}


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def band_count(world: int) -> int:
    """Latitude bands of the 100-yr field for `world` ranks: the 720 rows are 24 rows of 30-row threshold tiles, so whole tile
    rows allow 6, 8, 12 or 24 bands; the smallest count the rank count divides (6 at N = 1, 2, 3, 6; 8 at N = 4, 8), 0 if none."""
    return next((nb for nb in (6, 8, 12, 24) if nb % world == 0), 0)


def spawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` as a plain command: this process -- which has made no GPU call -- starts N ranks of itself
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them), lets rank 0 print the JSON line on
    this process's stdout, and returns the first non-zero exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others waiting in a collective: once one has failed, the rest (the processes started
    # here, by their own handles) get a few seconds and are then ended
    import time

    rc, failed_at = 0, None
    while any(pr.poll() is None for pr in procs):
        for pr in procs:
            code = pr.poll()
            if code and not rc:
                rc, failed_at = code, time.time()
        if failed_at is not None and time.time() - failed_at > 10.0:
            for pr in procs:
                if pr.poll() is None:
                    pr.terminate()
            failed_at = time.time() + 1e9  # asked once
        time.sleep(0.2)
    for pr in procs:
        rc = rc or (pr.returncode or 0)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=20240607)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("MAREX_BENCH_STREAMS", "0")),
                    help="engines (HIP streams) the bands of a rank alternate between; 0 = default (3 when their workspaces fit beside "
                         "the resident field, else 2; 1 for a single band / the one-GPU rehearsal)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed extra measurements (single-stream pass, seasonally drifting thresholds)")
    ap.add_argument("--hobday-path", default=None, choices=["tails", "bins"],
                    help="force the representation of the dayofyear histograms (default: the engine's own choice)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain command line: start the ranks from here (nothing in this process has touched the GPU)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    from marex_amd import binning, calendar, synth
    from marex_amd.dist import SUMMARY_KEYS, EngineSet, allreduce_step, broadcast_tables, plan_shards, shard_step
    from marex_amd.engine import HotPath

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # MAREX_BENCH_BACKEND=gloo + MAREX_BENCH_ONE_GPU=1: rehearse the multi-rank path on a single-GPU box
    backend = os.environ.get("MAREX_BENCH_BACKEND", "nccl")
    one_gpu = bool(os.environ.get("MAREX_BENCH_ONE_GPU"))
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    wl = WORKLOADS[args.workload]
    T, nx, W = wl["T"], wl["nx"], wl["W"]
    halo = wl["ws"] // 2
    nbands = wl.get("bands", 0)
    if nbands:
        # The band count is the rank's own tiling of ITS HBM, not part of the workload: the 720 rows are 24 rows of 30-row
        # threshold tiles, so whole tile rows allow 6, 8, 12 or 24 bands.  Fewer, taller bands carry fewer overlap rows
        # (124 / 120 instead of 94 / 90) and split evenly over two streams: 6 bands where the rank count divides 6 and two
        # band workspaces fit beside the rank's share of the field (measured at N = 1: 126.5 ms against 128.8-130.3 with 8
        # bands on three streams, 131.8 with 12), else 8.  MAREX_BENCH_BANDS overrides (experiments).
        if "MAREX_BENCH_BANDS" in os.environ:
            nbands = int(os.environ["MAREX_BENCH_BANDS"])
        else:
            nbands = band_count(world)
            if not nbands:
                raise SystemExit(f"--workload {args.workload} needs a GPU count that divides 24 (whole rows of threshold tiles per band)")
            if nbands == 6:
                cells6 = (wl["ny"] // 6 + 2 * halo) * nx
                need = (6 // world) * cells6 * 4 * T + 2 * 1.1 * (cells6 * (5 * T + 12 * 366) + cells6 * 2 * 366 * 16 * (T // 365 // 15 + 1))
                if need > torch.cuda.mem_get_info(local_rank)[0] and 8 % world == 0:
                    nbands = 8
    if nbands:  # fixed global grid cut into `nbands` bands, strong scaling: rank r takes bands r, r+world, ...
        if nbands % world:
            raise SystemExit(f"--workload {args.workload} needs a GPU count that divides {nbands}")
        all_shards = plan_shards(wl["ny"], nx, nbands, halo)
        shards = [all_shards[i] for i in range(rank, nbands, world)]
        ny_total = wl["ny"]
    elif wl["ny"] == 0:  # unstructured mesh, weak scaling: wl["nx"] cells per rank of a (world * nx)-cell mesh
        shards = [plan_shards(0, nx * world, world, 0)[rank]]
        ny_total = 0
    else:  # weak scaling: one band of wl["ny"] rows per rank
        shards = [plan_shards(wl["ny"] * world, nx, world, halo)[rank]]
        ny_total = wl["ny"] * world
    shard = shards[0]

    # default: three engines when their workspaces fit beside the resident input (measured on the 100-yr field, round 3: 141.0 ms with one
    # stream, 129.5-133.5 with two, 128.8-129.8 with three), else two; one for a single band and for the one-GPU rehearsal of several ranks
    # (an even number of bands goes over two streams: same speed as three with 8 bands, faster with 6)
    nstream = args.streams if args.streams > 0 else (1 if (len(shards) == 1 or one_gpu) else (2 if len(shards) % 2 == 0 else 3))
    nstream = max(1, min(nstream, len(shards)))
    if args.streams <= 0 and nstream == 3:
        cells = max(sh.cells_in for sh in shards)
        # an engine's workspace: anomalies (4 B) and extremes (1 B) per output step, thresholds twice, the sorted key lists
        per_engine = cells * (5 * T + 12 * 366) + cells * 2 * 366 * 16 * (T // 365 // 15 + 1)
        resident = sum(sh.cells_in for sh in shards) * 4 * T
        free = torch.cuda.mem_get_info(local_rank)[0]
        if resident + 3 * per_engine * 1.1 > free:
            nstream = 2
    eset = EngineSet(local_rank, nstream) if nstream > 1 else None
    hot = eset.engines[0] if eset else HotPath(local_rank)
    for e in (eset.engines if eset else [hot]):
        e.hobday_path = args.hobday_path
    tm = calendar.daily_time_axis(wl["start"], T)
    # host-built tables: rank 0 makes them, every rank works from the broadcast copy
    tables = None
    if rank == 0:
        cal0 = calendar.build_calendar(tm, window_year_baseline=W)
        bt0 = binning.hobday_bins()
        tables = calendar.plan_tables(cal0)
        tables.update({"bins.edges": bt0.edges, "bins.centres": bt0.centres, "bins.precision": float(bt0.precision),
                       "bins.max_anomaly": float(bt0.max_anomaly)})
        if wl.get("detrend_orders"):
            model0, pmodel0 = calendar.detrend_model(calendar.decimal_year(tm), list(wl["detrend_orders"]), False)
            tables.update({"detrend.model": model0, "detrend.pmodel": pmodel0})
    tables = broadcast_tables(tables, src=0, device=torch.device("cuda", local_rank), host_collectives=backend != "nccl")
    cal = calendar.plan_from_tables(tables)
    bt = binning.BinTable(edges=tables["bins.edges"], centres=tables["bins.centres"], precision=tables["bins.precision"],
                          max_anomaly=tables["bins.max_anomaly"])
    detrend = (tables["detrend.model"], tables["detrend.pmodel"]) if "detrend.model" in tables else None
    dcal = hot.upload_calendar(cal)
    xs = []  # resident input, one [T, cells_in] tensor per band, generated on the device before timing
    for sh in shards:
        if sh.gridded:
            tab = synth.make_tables(tm, sh.ny_in, nx, args.seed, lat_range=(sh.in0, sh.in1, sh.ny_global))
        else:
            tab = synth.make_tables(tm, 0, sh.cells_in, args.seed, unstructured=True)
        xs.append(hot.synth_field(tab, cell_base=sh.cell_base))
    torch.cuda.synchronize()

    workspace = {}  # output buffers are allocated once and reused: no allocator traffic in the timed loop

    step_kw = dict(W=W or 15, S=wl["S"], bins=bt, q=wl["pct"] / 100.0, wd=wl["wd"], ws=wl["ws"], nx=nx)

    def step():
        if eset is not None:
            r, local, mx = shard_step(eset, shards, xs, cal, detrend=detrend, **step_kw)
        else:
            r, local, mx = shard_step(hot, shards, xs, dcal, workspace=workspace, detrend=detrend, **step_kw)
        if world > 1:
            local, mx = allreduce_step(local, mx, host_collectives=backend != "nccl")
        return r, local, mx

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timers = eset if eset is not None else hot.ctx
    for _ in range(args.warmup):
        step()
    fence()
    timers.timing_enable(True)
    timers.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r, local, mx = step()
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=hot.device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    kern = {k: timers.timing_get(k) for k in ("shifting", "detrend", "fixed", "tails", "thresholds", "mask", "transpose")}
    kern = {k: v for k, v in kern.items() if v[1]}
    timers.timing_enable(False)
    path = r.get("path", "bins")
    summary = dict(zip(SUMMARY_KEYS, [int(v) for v in local.tolist()]))
    summary["max_invalid"] = int(mx.item())
    failures = []  # a line whose results are in doubt is not a benchmark line: printed with "failed", exit code 1
    if summary.get("thr_unresolved"):
        failures.append(f"threshold kernel left {summary['thr_unresolved']} outputs unresolved")

    if rank == 0:
        C_own_total = ny_total * nx if ny_total else nx * world
        units = T * C_own_total / 1e6  # Mcells*timesteps per step, whole job
        ms_step = dt / args.steps * 1e3
        value = units / (dt / args.steps)
        T_out = cal.T_out
        cells_own_rank = sum(sh.cells_own for sh in shards)
        b_alg_rank = algorithmic_bytes(T, T_out, cells_own_rank)
        # dominant kernel and its own algorithmic bytes PER LAUNCH (one launch = one band, averaged over this rank's bands;
        # DESIGN.md section 4)
        cin = sum(sh.cells_in for sh in shards) / len(shards)
        cown = cells_own_rank / len(shards)
        per_kernel_alg = {
            "detrend": cin * (4 * T + 4 * T + 1),
            "fixed": cin * (4 * T + 4 * T),
            "shifting": cin * (4 * T + 4 * T_out + 1),
            "tails": cin * 4 * T_out,            # its compulsory read (the tails it writes are internal)
            "thresholds": cown * 4 * 366,
            "mask": cown * (4 * T_out + T_out + 4 * 366),
            "transpose": cin * 8 * 366,
        }
        # With the bands on several streams a launch's HIP-event time includes what the neighbouring streams' kernels take: the
        # roofline object then speaks about a single-stream pass of the same step in this same run (every band on one engine,
        # kernels one after the other) and keeps the timed region's averages beside it
        serial = None
        if eset is not None and world == 1 and not args.no_extra:
            try:
                serial = serial_extra(hot, shards, xs, dcal, step_kw, detrend, T * (ny_total * nx if ny_total else nx * world) / 1e6,
                                      eset.workspaces[0])
            except Exception as e:  # noqa: BLE001
                serial = {"error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.empty_cache()
        if serial and "n_extreme" in serial and serial["n_extreme"] != summary["n_extreme"]:
            failures.append(f"n_extreme of the timed region ({summary['n_extreme']}) differs from the single-stream pass of the same "
                            f"step ({serial['n_extreme']})")
        per_launch = serial["kernel_ms"] if serial and serial.get("kernel_ms") else None
        dom = max(per_launch, key=per_launch.get) if per_launch else max(kern, key=lambda k: kern[k][0])
        # HBM bytes of the dominant kernel from the committed PMC passes of this same command (profiles/)
        traffic = None
        lean = (path == "tails" and detrend is None and step_kw["S"] == 21 and step_kw["W"] in (5, 15) and shard.cells_in % 4 == 0
                and os.environ.get("MAREX_SHIFT_LEAN", "1") != "0")  # engine.shifting_baseline_tails -> k_shift_lean (csrc/marex_shifting.hip)
        knames = {"shifting": "k_shift_lean" if lean else "k_shift_fast", "tails": "k_tail_extract", "transpose": "k_transpose", "detrend": "k_detrend", "fixed": "k_fixed_baseline",
                  "thresholds": "k_thr_tails" if path == "tails" else "k_thr_band",
                  "mask": "k_mask_tails" if path == "tails" else "k_mask_ge"}
        kname = knames[dom]
        for rnd in ("r04", "r03", "r02"):
            tfile = os.path.join(ROOT, "profiles", f"{rnd}_{args.workload}_traffic.json")
            if world == 1 and traffic is None and os.path.exists(tfile):
                for name, rec in json.load(open(tfile)).get("kernels", {}).items():
                    if name.startswith(kname):
                        traffic = rec["hbm_bytes"]
        avg_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in kern.items()}
        dom_ms = per_launch[dom] if per_launch else avg_ms[dom]
        achieved = per_kernel_alg[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
        out = {
            "metric": "Mcells*timesteps/s, shifting_baseline + hobday_extreme p95 (validation+anomaly+thresholds+mask)",
            "value": value,
            "unit": "Mcells*timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong" if nbands else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl["name"].replace("{bands}", str(nbands)),
                "per_gpu_grid": [sum(sh.own1 - sh.own0 for sh in shards), nx] if ny_total else [sum(sh.cells_own for sh in shards)],
                "global_grid": [ny_total, nx] if ny_total else [nx * world],
                "bands_per_gpu": len(shards),
                "bands_total": nbands or world,  # the tiling changes with the rank count (6 bands at N = 1, 2; 8 at N = 4, 8): same field, same results
                "streams_per_gpu": nstream,
                "timesteps_in": T,
                "timesteps_out": T_out,
                "parallelism": (f"lat-band x{max(world, nbands)}, {halo} overlap rows, tables broadcast from rank 0, scalar all-reduce per step" if ny_total
                                else f"cell ranges x{world}, tables broadcast from rank 0, scalar all-reduce per step"),
                "summary": summary,
                "histogram_representation": path,
            },
            # the figure north_star's ">= 40 % of HBM roofline" speaks about: SURVEY 8(d) bytes of the whole path / step time
            "pipeline_roofline": {
                "what": "whole path: algorithmic bytes of a step (SURVEY.md 8d) / ms_per_step -- the north-star fraction",
                "algorithmic_bytes_per_step_per_gpu": b_alg_rank,
                "achieved": b_alg_rank / (dt / args.steps) / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": b_alg_rank / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            },
            # the dominant KERNEL alone (its own algorithmic bytes / its own launch duration): kernel quality, not the path
            "roofline": {
                "what": "dominant kernel only: its algorithmic bytes per launch / its average launch duration",
                "bound": "hbm",
                "kernel": kname,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "avg_launch_ms": dom_ms,
                "algorithmic_bytes_per_launch": per_kernel_alg[dom],
                "timing": ("HIP events on the launch stream, single-stream pass of the same step inside this run (extra.single_stream)"
                           if per_launch else "HIP events on the launch stream over the timed region"),
            },
            "kernel_ms": avg_ms,
        }
        if world == 1 and not args.no_extra:
            # untimed additions: a failure here (say, no room left for another workspace) must not cost the line
            out["extra"] = {}
            wsp0 = eset.workspaces[0] if eset is not None else workspace
            if serial is not None:
                out["extra"]["single_stream"] = serial
                if per_launch and avg_ms.get(dom):
                    a_t = per_kernel_alg[dom] / (avg_ms[dom] * 1e-3) / 1e9
                    out["roofline"]["timed_region"] = {
                        "avg_launch_ms": avg_ms[dom], "achieved": a_t, "frac": a_t / HBM_PEAK_GBS,
                        "note": f"HIP events on each engine's stream over the timed region; {nstream} streams share the GPU, so a "
                                "launch's duration includes what its neighbours take"}
            if detrend is None:
                try:
                    base_ms = out["extra"].get("single_stream", {}).get("kernel_ms") or avg_ms
                    with torch.cuda.stream(hot.stream) if hot.stream is not None else _null():
                        out["extra"]["seasonal_field"] = seasonal_extra(hot, shards[0], xs[0], dcal, cal, step_kw, base_ms, wsp=wsp0)
                except Exception as e:  # noqa: BLE001
                    out["extra"]["seasonal_field"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(wl, args.seed)
            except Exception as e:  # noqa: BLE001  (a host without room for the worker pool still gets its GPU line)
                out["cpu_baseline"] = {"value": None, "unit": "Mcells*timesteps/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"[:300]}
        if failures:
            out["failed"] = failures
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failures:
        raise SystemExit("bench.py: " + "; ".join(failures))


if __name__ == "__main__":
    main()
