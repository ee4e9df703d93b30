"""GPU: the multi-device paths through the product.

* `preprocess_data(..., devices=[...])`: one engine + host thread per listed device, blocks dealt round-robin, the host
  stitches the Dataset.  One card is available here, so it is listed twice: two independent engines
  (contexts, streams) drive it concurrently -- the result must equal the single-device Dataset bit for bit.
* `marex_amd.dist.shard_step` + the RCCL collectives (`nccl` backend, world size 1): the code every rank of `bench.py` runs,
  HIP kernels and `torch.distributed` together under pytest.
"""
import os
import socket
import warnings

import numpy as np
import pytest
import torch

import marex_amd
from marex_amd import binning, calendar, synth
from marex_amd.dist import EngineSet, allreduce_step, broadcast_tables, gather_owned_cells, plan_shards, shard_step, stitch_cells
from marex_amd.xr_compat import DataArray
from oracle import marex_oracle as orc

pytestmark = pytest.mark.gpu


def _da(ny, nx, years=12, start="2001-01-01", unstructured=False):
    tm = calendar.daily_time_axis(start, years * 365 + 3)
    if unstructured:
        x = synth.synth_field(synth.make_tables(tm, 0, nx, unstructured=True))
        n = x.shape[1]
        return DataArray(x, dims=("time", "ncells"), coords={"time": tm, "lat": ("ncells", np.linspace(-80, 80, n)),
                                                               "lon": ("ncells", np.linspace(0, 359, n))}, name="sst"), tm
    x = synth.synth_field(synth.make_tables(tm, ny, nx)).reshape(len(tm), ny, nx)
    return DataArray(x, dims=("time", "lat", "lon"), coords={"time": tm, "lat": np.linspace(-60, 60, ny), "lon": np.linspace(0, 350, nx)},
                     name="sst"), tm


@pytest.mark.parametrize("kw", [
    dict(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", window_year_baseline=5),
    dict(method_anomaly="detrend_harmonic", method_extreme="hobday_extreme", std_normalise=True),
    dict(method_anomaly="fixed_baseline", method_extreme="global_extreme"),
])
def test_devices_list_equals_single_device(hot, kw):
    da, _ = _da(17, 24)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = marex_amd.preprocess_data(da, **kw)
        two = marex_amd.preprocess_data(da, devices=[0, 0], **kw)       # two engines on the one card, 2 latitude bands
        three = marex_amd.preprocess_data(da, devices=[0, 0, 0], **kw)  # 3 bands on 3 engines
    for ds in (two, three):
        assert set(ds.data_vars) == set(one.data_vars)
        for v in one.data_vars:
            assert np.array_equal(np.asarray(ds[v].values), np.asarray(one[v].values), equal_nan=True), v
        assert ds.attrs == one.attrs


def test_devices_list_on_a_mesh(hot):
    da, _ = _da(0, 431, unstructured=True)
    kw = dict(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", window_year_baseline=4,
              dimensions={"time": "time", "x": "ncells"}, coordinates={"time": "time", "x": "lon", "y": "lat"})
    one = marex_amd.preprocess_data(da, **kw)
    two = marex_amd.preprocess_data(da, devices=[0, 0], **kw)
    for v in one.data_vars:
        assert np.array_equal(np.asarray(two[v].values), np.asarray(one[v].values), equal_nan=True), v


def test_shard_step_with_rccl_collectives(hot):
    """world size 1 over the `nccl` backend (= RCCL): shard_step -> all_reduce -> all_gather on device tensors; two latitude
    bands on the one rank, stitched against the oracle on the whole field."""
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0,
                            device_id=torch.device("cuda", hot.device.index))
    try:
        ny, nx, W = 14, 20, 4
        tm = calendar.daily_time_axis("2003-01-01", 9 * 365 + 2)
        cal = calendar.build_calendar(tm, window_year_baseline=W)
        dcal = hot.upload_calendar(cal)
        bt = binning.hobday_bins()
        shards = plan_shards(ny, nx, 2, 2)
        xs = [hot.synth_field(synth.make_tables(tm, sh.ny_in, nx, lat_range=(sh.in0, sh.in1, ny)), cell_base=sh.cell_base) for sh in shards]
        thr_parts, ext_parts, tot = [], [], None
        for sh, x in zip(shards, xs):  # one band per step: the outputs of a band are read before the next one reuses nothing
            r, local, mx = shard_step(hot, [sh], [x], dcal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, nx=nx)
            local, mx = allreduce_step(local, mx)                      # RCCL all-reduce of the int64 scalars
            tot = local.clone() if tot is None else tot + local
            own = sh.own_cell_slice()
            thr_parts.append(r["thr_doy_major"][:, own].contiguous())
            ext_parts.append(r["extreme_events"].cpu().numpy())
        gathered = gather_owned_cells(thr_parts[0], [shards[0]], 0)    # RCCL all-gather (one rank: its own part back)
        hot.sync()
        x_full = synth.synth_field(synth.make_tables(tm, ny, nx))
        ref = orc.preprocess_arrays(x_full, cal, ny=ny, nx=nx, window_year_baseline=W, edges=bt.edges, centres=bt.centres)
        thr = torch.cat(thr_parts, dim=1).cpu().numpy()
        assert np.array_equal(thr, ref["thresholds"].T, equal_nan=True)
        assert np.array_equal(gathered.cpu().numpy(), thr_parts[0].cpu().numpy(), equal_nan=True)
        assert np.array_equal(stitch_cells(ext_parts, shards).astype(bool), ref["extreme_events"])
        v = orc.validate_data_values(x_full)
        assert [int(t) for t in tot[:4]] == [v["n_ocean"], v["total_invalid_in_ocean"], v["locations_affected"], int(ref["extreme_events"].sum())]
        # the tables of a run through the RCCL broadcast (object manifest + one device byte tensor), as bench.py's ranks get them
        tables = calendar.plan_tables(cal)
        tables.update({"bins.edges": bt.edges, "bins.centres": bt.centres, "bins.precision": 0.01})
        got = broadcast_tables(tables, src=0, device=torch.device("cuda", hot.device.index), force=True)
        cal2 = calendar.plan_from_tables(got)
        assert np.array_equal(cal2.year_plan(), cal.year_plan()) and cal2.T_out == cal.T_out and cal2.kept.dtype == np.bool_
        assert got["bins.edges"].tobytes() == bt.edges.tobytes() and got["bins.precision"] == 0.01
        # and the multi-stream schedule of a rank with several bands, reduced over RCCL
        es = EngineSet(hot.device.index, 2)
        r2, local2, mx2 = shard_step(es, shards, xs, cal2, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, nx=nx)
        local2, mx2 = allreduce_step(local2, mx2)
        torch.cuda.synchronize()
        assert [int(t) for t in local2[:4]] == [int(t) for t in tot[:4]]
    finally:
        dist.destroy_process_group()
