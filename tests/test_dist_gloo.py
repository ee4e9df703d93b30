"""Multi-GPU path on CPU: 2 ranks over gloo.

The spatial shard plan (latitude bands with overlap rows, marex_amd/dist.py) and the scalar all-reduce are
exercised with the ORACLE standing in for the device kernels (tests may use the oracle as the checker):
each rank processes its band, the stitched result must be BIT-IDENTICAL to the single-shard result.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from marex_amd import binning, calendar, synth
from marex_amd.dist import Shard, plan_shards, stitch_cells


def test_plan_shards_properties():
    sh = plan_shards(720, 1440, 8, 2)
    assert [s.own0 for s in sh] == [90 * r for r in range(8)] and sh[-1].own1 == 720
    assert sh[0].in0 == 0 and sh[0].in1 == 92 and sh[3].in0 == 268 and sh[3].in1 == 362 and sh[7].in1 == 720
    assert sum(s.cells_own for s in sh) == 720 * 1440
    assert sh[3].own_cell_slice() == slice(2 * 1440, 92 * 1440) and sh[3].cell_base == 268 * 1440
    un = plan_shards(0, 1001, 4, 2)
    assert [u.cells_own for u in un] == [251, 250, 250, 250] and all(u.in0 == u.own0 and u.in1 == u.own1 for u in un)
    odd = plan_shards(7, 5, 3, 1)
    assert [(s.own0, s.own1, s.in0, s.in1) for s in odd] == [(0, 3, 0, 4), (3, 5, 2, 6), (5, 7, 4, 7)]
    with pytest.raises(ValueError):
        plan_shards(2, 5, 3, 1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleEngine:
    """Stand-in for marex_amd.engine.HotPath on the CPU: the two methods `marex_amd.dist.shard_step` calls, computed by the
    oracle (tests may use the oracle as the checker).  Everything around it -- shard plan, owned rows / cells, the scalar
    all-reduce, the all-gather of thresholds -- is the code bench.py and the GPU ranks run."""

    device = "cpu"

    def shifting_hobday(self, x, dcal, *, W, S, bins, q, wd, ws, ny, nx, own_rows=None, workspace=None):
        import torch

        from oracle import marex_oracle as orc

        xn = x.numpy()
        r = orc.preprocess_arrays(xn, dcal, ny=ny, nx=nx, window_year_baseline=W, smooth_days_baseline=S, window_days_hobday=wd,
                                  window_spatial_hobday=ws, threshold_percentile=q * 100.0, edges=bins.edges, centres=bins.centres)
        own = slice(None) if own_rows is None else slice(own_rows[0] * nx, own_rows[1] * nx)
        low = int(r["stats"]["n_too_low"]) if own_rows is None else 0  # (per-shard warning counters are not compared here)
        return {
            "dat_anomaly": torch.from_numpy(r["dat_anomaly"]), "mask": torch.from_numpy(r["mask"].astype(np.uint8)),
            "invalid_count": torch.from_numpy((~np.isfinite(xn)).sum(axis=0).astype(np.int32)),
            "thr_doy_major": torch.from_numpy(np.ascontiguousarray(r["thresholds"].T)),
            "extreme_events": torch.from_numpy(r["extreme_events"].astype(np.uint8)),
            "n_true": torch.tensor([int(r["extreme_events"][:, own].sum())], dtype=torch.int64),
            "stats_dev": torch.tensor([0, 0, low, 0, 0, 0, 0, 0], dtype=torch.int32),
        }

    def validation_summary(self, mask, invalid, cells, wsp=None):
        import torch

        m = mask[cells[0]:cells[1]].numpy().astype(bool)
        inv = np.where(m, invalid[cells[0]:cells[1]].numpy(), 0)
        return torch.tensor([int(m.sum()), int(inv.sum()), int((inv > 0).sum()), int(inv.max()) if inv.size else 0], dtype=torch.int64)


def _worker(rank, world, port, ny, nx, W, out_dir):
    import torch
    import torch.distributed as dist

    from marex_amd.dist import allreduce_step, gather_owned_cells, shard_step

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tm = calendar.daily_time_axis("2003-01-01", 9 * 365 + 2)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    bt = binning.hobday_bins()
    all_shards = plan_shards(ny, nx, world, 2)
    shard = all_shards[rank]
    tab = synth.make_tables(tm, shard.ny_in, nx, lat_range=(shard.in0, shard.in1, ny))
    x = torch.from_numpy(synth.synth_field(tab, cell_base=shard.cell_base))
    r, local, mx = shard_step(OracleEngine(), [shard], [x], cal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, nx=nx)
    local, mx = allreduce_step(local, mx, host_collectives=True)
    own = shard.own_cell_slice()
    thr_all = gather_owned_cells(r["thr_doy_major"][:, own], all_shards, rank, host_collectives=True)   # [366, all cells]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), anom=r["dat_anomaly"].numpy(), thr=r["thr_doy_major"].numpy(),
             ext=r["extreme_events"].numpy().astype(bool), thr_all=thr_all.numpy(),
             summ=np.array([int(local[0]), int(local[1]), int(mx[0]), int(local[3])]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_band_sharding_is_bit_identical(tmp_path):
    from oracle import marex_oracle as orc

    world, ny, nx, W = 2, 10, 12, 4
    mp.spawn(_worker, args=(world, _free_port(), ny, nx, W, str(tmp_path)), nprocs=world, join=True)
    tm = calendar.daily_time_axis("2003-01-01", 9 * 365 + 2)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    bt = binning.hobday_bins()
    x = synth.synth_field(synth.make_tables(tm, ny, nx))
    ref = orc.preprocess_arrays(x, cal, ny=ny, nx=nx, window_year_baseline=W, edges=bt.edges, centres=bt.centres)
    shards = plan_shards(ny, nx, world, 2)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    # the global field restricted to a shard equals the shard's own synthetic field (global cell ids)
    assert np.array_equal(stitch_cells([p["anom"] for p in parts], shards), ref["dat_anomaly"], equal_nan=True)
    assert np.array_equal(stitch_cells([p["thr"] for p in parts], shards), ref["thresholds"].T, equal_nan=True)
    assert np.array_equal(stitch_cells([p["ext"] for p in parts], shards), ref["extreme_events"])
    v = orc.validate_data_values(x)
    for p in parts:  # every rank holds the all-reduced scalars and the all-gathered thresholds of the whole grid
        assert list(p["summ"]) == [v["n_ocean"], v["total_invalid_in_ocean"], v["max_invalid"], int(ref["extreme_events"].sum())]
        assert np.array_equal(p["thr_all"], ref["thresholds"].T, equal_nan=True)


def _bcast_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from marex_amd.dist import broadcast_tables

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tables = None
    if rank == 0:  # only the source builds anything
        tm = calendar.daily_time_axis("1999-03-01", 7 * 365 + 40)
        cal = calendar.build_calendar(tm, window_year_baseline=3)
        bt = binning.hobday_bins()
        model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), [1, 2], True)
        tables = calendar.plan_tables(cal)
        tables.update({"bins.edges": bt.edges, "bins.centres": bt.centres, "bins.precision": 0.01, "detrend.model": model,
                       "detrend.pmodel": pmodel, "empty": np.zeros((0, 3), dtype=np.float32),
                       # what the packed byte path cannot carry goes with the manifest: object arrays (cftime axes), 0-d arrays
                       "objs": np.array(["1999-03-01", None, 3], dtype=object), "zero_d": np.array(2.5, dtype=np.float64)})
    got = broadcast_tables(tables, src=0, host_collectives=True)
    cal = calendar.plan_from_tables(got)
    np.savez(os.path.join(out_dir, f"bc{rank}.npz"), year_plan=cal.year_plan(), kept=cal.kept, doy_rows=cal.doy_rows, time=cal.time.astype("int64"),
             edges=got["bins.edges"], centres=got["bins.centres"], model=got["detrend.model"], pmodel=got["detrend.pmodel"],
             scal=np.array([cal.min_year, cal.n_cal_years, cal.first_valid_year_idx, int(cal.has_duplicates), cal.T_out]),
             prec=np.array([got["bins.precision"]]), empty_shape=np.array(got["empty"].shape),
             objs_ok=np.array([got["objs"].dtype == object and list(got["objs"]) == ["1999-03-01", None, 3]
                               and got["zero_d"].shape == () and float(got["zero_d"]) == 2.5]))
    dist.barrier()
    dist.destroy_process_group()


def test_tables_broadcast_from_rank_zero(tmp_path):
    """calendar tables, bin edges / centres and the detrend model travel from rank 0 to the other ranks, which
    derive nothing themselves; the received plan is the plan."""
    world = 2
    mp.spawn(_bcast_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    tm = calendar.daily_time_axis("1999-03-01", 7 * 365 + 40)
    cal = calendar.build_calendar(tm, window_year_baseline=3)
    bt = binning.hobday_bins()
    model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), [1, 2], True)
    for r in range(world):
        p = np.load(tmp_path / f"bc{r}.npz")
        assert np.array_equal(p["year_plan"], cal.year_plan()) and np.array_equal(p["kept"], cal.kept)
        assert p["kept"].dtype == np.bool_ and np.array_equal(p["doy_rows"], cal.doy_rows)
        assert np.array_equal(p["time"], cal.time.astype("int64"))
        assert p["edges"].dtype == np.float32 and p["edges"].tobytes() == bt.edges.tobytes() and p["centres"].tobytes() == bt.centres.tobytes()
        assert p["model"].tobytes() == model.tobytes() and p["pmodel"].tobytes() == pmodel.tobytes()
        assert bool(p["objs_ok"][0])
        assert list(p["scal"]) == [cal.min_year, cal.n_cal_years, cal.first_valid_year_idx, 0, cal.T_out]
        assert float(p["prec"][0]) == 0.01 and list(p["empty_shape"]) == [0, 3]


def _bands_worker(rank, world, port, ny, nx, W, nbands, out_dir):
    """What a rank of `bench.py --gpus N` does with the 100-yr field, at toy size: bands rank, rank + N, ... of `nbands` latitude
    bands, one `shard_step` over them, the scalar all-reduce."""
    import torch
    import torch.distributed as dist

    from marex_amd.dist import SUMMARY_KEYS, allreduce_step, shard_step

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tm = calendar.daily_time_axis("2003-01-01", 9 * 365 + 2)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    bt = binning.hobday_bins()
    all_shards = plan_shards(ny, nx, nbands, 2)
    shards = [all_shards[i] for i in range(rank, nbands, world)]
    xs = [torch.from_numpy(synth.synth_field(synth.make_tables(tm, s.ny_in, nx, lat_range=(s.in0, s.in1, ny)), cell_base=s.cell_base))
          for s in shards]
    parts = []

    class Keep(OracleEngine):  # shard_step hands back the last shard's result only: keep every band's thresholds
        def shifting_hobday(self, x, dcal, **kw):
            r = super().shifting_hobday(x, dcal, **kw)
            parts.append(r["thr_doy_major"].numpy())
            return r

    _, local, mx = shard_step(Keep(), shards, xs, cal, W=W, S=21, bins=bt, q=0.95, wd=11, ws=5, nx=nx)
    local, mx = allreduce_step(local, mx, host_collectives=True)
    np.savez(os.path.join(out_dir, f"b{rank}.npz"), summ=np.array([int(v) for v in local.tolist()] + [int(mx[0])]),
             keys=np.array(SUMMARY_KEYS), **{f"thr{i}": p for i, p in zip(range(rank, nbands, world), parts)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_band_assignment_and_reduction_at_the_scale_runs_rank_counts(tmp_path, world):
    """The driver's scaling run uses N = 1, 2, 4, 8 ranks; at N = 4 and 8 bench.py cuts the field into 8 bands and rank r takes
    bands r, r + N, ...  Same assignment, same `shard_step` + all-reduce here with the oracle as the engine: every owned row is
    computed exactly once, the reduced counters are the field's, on every rank (a one-GPU box cannot hold 4 or 8 GPU ranks of
    the 100-yr field -- 8 exceed its process limit -- so this is where those rank counts are rehearsed)."""
    import importlib.util

    from oracle import marex_oracle as orc

    spec = importlib.util.spec_from_file_location("bench_for_bands", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    nbands = bench.band_count(world)
    assert nbands == 8
    ny, nx, W = 48, 8, 4   # 8 bands of 6 rows, 2 overlap rows per interior side
    mp.spawn(_bands_worker, args=(world, _free_port(), ny, nx, W, nbands, str(tmp_path)), nprocs=world, join=True)
    tm = calendar.daily_time_axis("2003-01-01", 9 * 365 + 2)
    cal = calendar.build_calendar(tm, window_year_baseline=W)
    bt = binning.hobday_bins()
    x = synth.synth_field(synth.make_tables(tm, ny, nx))
    ref = orc.preprocess_arrays(x, cal, ny=ny, nx=nx, window_year_baseline=W, edges=bt.edges, centres=bt.centres)
    v = orc.validate_data_values(x)
    shards = plan_shards(ny, nx, nbands, 2)
    thr = {}
    for r in range(world):
        p = np.load(tmp_path / f"b{r}.npz")
        summ = dict(zip([str(k) for k in p["keys"]], p["summ"][:-1]))
        assert summ["n_ocean"] == v["n_ocean"] and summ["n_extreme"] == int(ref["extreme_events"].sum()) and summ["thr_unresolved"] == 0
        assert summ["invalid_total"] == v["total_invalid_in_ocean"] and int(p["summ"][-1]) == v["max_invalid"]
        for i in range(r, nbands, world):
            assert i not in thr
            thr[i] = p[f"thr{i}"]
    assert sorted(thr) == list(range(nbands))
    assert np.array_equal(stitch_cells([thr[i] for i in range(nbands)], shards), ref["thresholds"].T, equal_nan=True)
