"""Device pipeline of the hot path: torch owns HBM buffers and streams, the HIP library does the work.

``HotPath`` is the array-level engine underneath :func:`marex_amd.detect.preprocess_data`.  It takes
``[T, C]`` float32 device tensors (C-order ``(time, cells)`` exactly like the reference's
``(time, lat, lon)`` arrays) plus the host calendar / bin tables and calls the C ABI
(``include/marex_hip.h``) stage by stage.  No stage has a CPU implementation here.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .binning import BinTable
from .calendar import N_DOY, CalendarPlan
from .exceptions import ConfigurationError, ProcessingError


import logging

_log = logging.getLogger("marex_amd")
_noted: set = set()


def _note_path(msg: str) -> None:
    """INFO, once per distinct message: a call took a slower kernel family than the tuned one (same results; README "kernel
    families") -- so that a user with, say, smooth_days_baseline=31 sees why the anomaly stage runs at a quarter of the speed."""
    if msg not in _noted:
        _noted.add(msg)
        _log.info(msg)


def shifting_kernel_family(W: int, S: int, C: int, lists: bool) -> str:
    """Which anomaly kernel ``marex_shifting_baseline[_tails]_f32`` takes for a gap-free daily calendar (csrc/marex_shifting.hip:
    ``lean_instance`` / ``fast_ws``; irregular calendars send single chunks to the general kernel on top of this)."""
    fast_w = W in (3, 4, 5, 6, 7, 10, 13, 15)
    fast = fast_w if S == 21 else (S in (11, 15) and W in (5, 10, 15))
    if S == 21 and W in (5, 15) and C >= 4 and C % 4 == 0:
        return "lean"
    return "fast" if fast else "general"


def _key_to_float(key: int) -> float:
    """Inverse of the order-preserving uint32 key used for the device-side min / max of thresholds."""
    bits = (key & 0x7FFFFFFF) if (key & 0x80000000) else (~key & 0xFFFFFFFF)
    return float(np.array([bits], dtype=np.uint32).view(np.float32)[0])


@dataclass
class DeviceCalendar:
    """Calendar tables resident on the device."""

    tindex: torch.Tensor
    year_plan: torch.Tensor
    out_index: torch.Tensor
    rowb_index: torch.Tensor
    doy_start: torch.Tensor
    doy_rows: torch.Tensor
    plan: CalendarPlan


class HotPath:
    def __init__(self, device: int | torch.device = 0, own_stream: bool = False):
        if not torch.cuda.is_available():
            raise ProcessingError(
                "marex_amd needs a HIP device (torch.cuda.is_available() is False)",
                details="the hot path has no CPU implementation",
            )
        self.device = torch.device("cuda", device if isinstance(device, int) else (device.index or 0))
        self.ctx = _lib.Context(self.device.index)
        self.lib = self.ctx.lib
        self._tables: Dict[tuple, tuple] = {}
        #: None = choose per configuration (tails_plan); "tails" / "bins" force the representation of the dayofyear
        #: histograms behind the approximate Hobday thresholds (same results either way; tests use it)
        self.hobday_path: Optional[str] = None
        #: a stream of its own (engines that share a device with another engine run under ``torch.cuda.stream(self.stream)``)
        self.stream = torch.cuda.Stream(self.device) if own_stream else None
        self._bind_stream()

    #: tests set this: fresh output buffers are filled with a byte pattern, so that an element a kernel forgets to write
    #: shows up as garbage instead of as the zeros a fresh allocation often holds
    POISON = False

    @staticmethod
    def _buf(wsp: Optional[dict], name: str, shape, dtype, device) -> torch.Tensor:
        """Output buffer: fresh when ``wsp`` is None, otherwise cached in the workspace dict and reused
        (no allocator traffic in the steady state, outputs of the previous call are overwritten)."""
        if wsp is None:
            t = torch.empty(shape, dtype=dtype, device=device)
            if HotPath.POISON and t.numel():
                t.view(torch.uint8).fill_(0xCD)
            return t
        n = 1
        for d in shape:
            n *= int(d)
        flat = wsp.get(name)
        if flat is None or flat.dtype != dtype or flat.numel() < n:
            flat = torch.empty((max(n, 1),), dtype=dtype, device=device)  # grows to the largest request, then stays
            wsp[name] = flat
        return flat[:n].view(shape)

    # ------------------------------------------------------------------ plumbing
    def _bind_stream(self) -> None:
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a: np.ndarray, dtype=None) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(a if dtype is None else a.astype(dtype)))
        return t.to(self.device, non_blocking=False)

    def upload_calendar(self, cal: CalendarPlan) -> DeviceCalendar:
        return DeviceCalendar(
            tindex=self._dev(cal.tindex, np.int32),
            year_plan=self._dev(cal.year_plan(), np.int32),
            out_index=self._dev(cal.out_index, np.int32),
            rowb_index=self._dev(cal.rowb_index, np.int32),
            doy_start=self._dev(cal.doy_start, np.int32),
            doy_rows=self._dev(cal.doy_rows, np.int32),
            plan=cal,
        )

    def sync(self) -> None:
        self.ctx.sync()

    def ctx_opt(self, name: str, default: int) -> int:
        """Python-side view of a context option set through ``ctx.options`` / ``set_option`` (host-side switches)."""
        return int(self.ctx.py_opts.get(name, default))

    @staticmethod
    def bins_shape(T_out: int, C: int):
        """Shape of the blocked bin matrix: ``[ceil(C/16), T_out, 16]`` (include/marex_hip.h)."""
        return ((C + 15) // 16, T_out, 16)

    @staticmethod
    def bins_to_rows(binsb: torch.Tensor, C: int) -> torch.Tensor:
        """Blocked bin matrix -> plain ``[T_out, C]`` (dayofyear-sorted rows); for tests / inspection."""
        nblk, T_out, _ = binsb.shape
        return binsb.permute(1, 0, 2).reshape(T_out, nblk * 16)[:, :C]

    def bin_tables(self, bins: BinTable):
        """(edges, centres) of a bin table on the device, uploaded once per table CONTENT (callers build a fresh BinTable
        per call; a handful of distinct tables at most stay resident)."""
        key = (float(bins.precision), float(bins.max_anomaly), int(bins.nb), bins.edges.tobytes())
        if key not in self._tables:
            if len(self._tables) >= 8:
                self._tables.pop(next(iter(self._tables)))
            self._tables[key] = (self._dev(bins.edges, np.float32), self._dev(bins.centres, np.float32))
        return self._tables[key]

    # ------------------------------------------------------------------ synthetic field
    def synth_field(self, tab, cell_base: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Device evaluation of :func:`marex_amd.synth.synth_field` (bit-identical)."""
        self._bind_stream()
        T, Cn = tab.T, tab.C
        if out is None:
            out = torch.empty((T, Cn), dtype=torch.float32, device=self.device)
        bufs = [
            self._dev(tab.mean, np.float32), self._dev(tab.amp, np.float32), self._dev(tab.hemi, np.uint8),
            self._dev(tab.land, np.uint8), self._dev(tab.seas, np.float32), self._dev(tab.trend, np.float32),
        ]
        rc = self.lib.marex_synth_sst_f32(
            self.ctx.handle, *[b.data_ptr() for b in bufs], tab.seed, int(cell_base), T, Cn, out.data_ptr()
        )
        self.ctx.check(rc, "marex_synth_sst_f32")
        self.sync()  # the small tables above must outlive the kernel
        return out

    # ------------------------------------------------------------------ stage a3+a5+a6+a7 (+a10 binning)
    def shifting_baseline(
        self,
        x: torch.Tensor,
        dcal: DeviceCalendar,
        W: int,
        S: int,
        bins: Optional[BinTable] = None,
        write_clim: bool = False,
        wsp: Optional[dict] = None,
    ) -> Dict[str, torch.Tensor]:
        self._bind_stream()
        assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
        T, Cn = x.shape
        cal = dcal.plan
        if cal.T != T:
            raise ProcessingError("calendar length does not match the time axis of x")
        if cal.has_duplicates:
            raise ConfigurationError(
                "shifting_baseline needs at most one timestep per (year, dayofyear)",
                details="sub-daily time axes are not supported by the device path",
            )
        T_out = cal.T_out
        out = self._buf(wsp, "anom", (T_out, Cn), torch.float32, self.device)
        if write_clim:
            out.fill_(float("nan"))
        mask = self._buf(wsp, "mask", (Cn,), torch.uint8, self.device)
        invalid = self._buf(wsp, "invalid", (Cn,), torch.int32, self.device)
        invalid.zero_()
        if bins is not None and not write_clim:
            edges = self.bin_tables(bins)[0]
            binsb = self._buf(wsp, "bins", self.bins_shape(T_out, Cn), torch.int16, self.device)
            e_ptr, b_ptr, nb = edges.data_ptr(), binsb.data_ptr(), bins.nb
        else:
            edges = binsb = None
            e_ptr, b_ptr, nb = None, None, 0
        rc = self.lib.marex_shifting_baseline_f32(
            self.ctx.handle, x.data_ptr(), T, Cn, dcal.year_plan.data_ptr(), cal.n_cal_years,
            int(W), int(S), int(write_clim),
            e_ptr, nb, T_out, out.data_ptr(), b_ptr, mask.data_ptr(), invalid.data_ptr(),
        )
        self.ctx.check(rc, "marex_shifting_baseline_f32")
        res = {"out": out, "mask": mask, "invalid_count": invalid, "_keep": edges}
        if binsb is not None:
            res["bins"] = binsb
        return res

    # ------------------------------------------------------------------ stage a10/a11
    def hobday_thresholds(
        self,
        binsb: torch.Tensor,
        first_anom: torch.Tensor,
        dcal: DeviceCalendar,
        bins: BinTable,
        q: float,
        wd: int,
        ws: int,
        ny: int,
        nx: int,
        rows: Optional[tuple] = None,
        wsp: Optional[dict] = None,
    ) -> Dict[str, object]:
        """``rows=(row0, row1)`` restricts the output to the grid rows a latitude shard owns."""
        self._bind_stream()
        T_out, Cn = binsb.shape[1], first_anom.shape[-1]
        row0, row1 = rows if rows is not None else (0, max(ny, 1))
        thr = self._buf(wsp, "thr_doy_major", (N_DOY, Cn), torch.float32, self.device)
        stats = self._buf(wsp, "thr_stats", (8,), torch.int32, self.device)  # marex_thr_stats: 8 x uint32
        stats.zero_()
        stats[0:1].fill_(-1)  # min_key = 0xFFFFFFFF (a fill kernel: `stats[0] = -1` would be a blocking host-to-device copy)
        centres = self.bin_tables(bins)[1]
        rc = self.lib.marex_hobday_thresholds_f32(
            self.ctx.handle, binsb.data_ptr(), T_out, Cn, int(ny), int(nx), dcal.doy_start.data_ptr(),
            int(np.diff(dcal.plan.doy_start).max()), first_anom.data_ptr(), centres.data_ptr(), bins.nb, float(q), int(wd), int(ws),
            float(bins.lower_bound), float(bins.upper_bound), int(row0), int(row1), thr.data_ptr(), stats.data_ptr(),
        )
        self.ctx.check(rc, "marex_hobday_thresholds_f32")
        return {"thr_doy_major": thr, "stats_dev": stats, "_keep": centres}

    @staticmethod
    def decode_thr_stats(stats_dev: torch.Tensor) -> Dict[str, float]:
        s = stats_dev.cpu().numpy().view(np.uint32)
        kmin, kmax = int(s[0]), int(s[1])
        if len(s) > 4 and int(s[4]):  # the straggler-pass limit of the list threshold kernel tripped (marex_tails.hip): never a result
            raise ProcessingError(
                f"threshold kernel left {int(s[4])} outputs unresolved (written as NaN)",
                details="internal error of the day-of-year threshold stage: the band search ran out of passes",
                suggestions=["force the bin-matrix kernels (engine.hobday_path = 'bins') and report the input"],
            )
        return {
            "min": _key_to_float(kmin) if kmin != 0xFFFFFFFF else float("nan"),
            "max": _key_to_float(kmax) if kmax != 0 else float("nan"),
            "n_too_low": int(s[2]),
            "n_too_high": int(s[3]),
        }

    # ------------------------------------------------------------------ stage a10/a11 on tails (default)
    #: rows per sorted list: what the extraction kernel writes / what the shifting-baseline kernel emits itself
    LIST_ROWS_EXTRACT = 32
    LIST_ROWS_SHIFT = 15

    def tails_plan(self, dcal: DeviceCalendar, bins: BinTable, q: float, wd: int, ws: int, C: Optional[int] = None) -> Optional[int]:
        """Rows of the largest dayofyear bucket when the tail kernels take this configuration, else None (bin-matrix
        kernels).  Results are identical on both paths (include/marex_hip.h, TAILS)."""
        if self.hobday_path == "bins":
            return None
        nd = int(np.diff(dcal.plan.doy_start).max())
        if not (bins.nb <= 511 and 1 <= nd <= 128 and ws <= 7 and nd * wd * ws * ws <= 65535 and (C is None or C <= (1 << 24))):
            return None
        if self.hobday_path != "tails" and nd < 24 and ws > 1:
            return None  # short buckets with spatial pooling (cfg2): the bin-matrix kernels are the faster ones
        if self.hobday_path != "tails" and q < 0.5:
            return None  # the lists are read from the top: a low quantile walks most of every list (the public API stops at 60 %)
        return nd  # ws == 1: the per-cell threshold kernel (no tiles), any record length

    def fused_tails_ok(self, dcal: DeviceCalendar, bins: BinTable) -> bool:
        """The fixed-baseline kernels (plain and behind the detrend fit) can emit the key lists of their own output for buckets
        of at most 128 rows and tables of at most 511 bins.  Option FIXED_TAILS: 1 (default) = for buckets of at most 48 rows
        -- the 48-row register kernel keeps four waves per SIMD with the sorting networks in it; the 128-row one drops to two and
        measured SLOWER than kernel + extraction pass on the 100-yr field (27.9 vs 21.0 ms per band, profiles/r04_experiments.md)
        --, 2 = whenever possible, 0 = never (the extraction pass makes the lists)."""
        nd = int(np.diff(dcal.plan.doy_start).max())
        mode = self.ctx_opt("FIXED_TAILS", 1)
        return bool(mode) and bool(self.ctx_opt("FIXED_REG", 1)) and 1 <= nd <= (128 if mode == 2 else 48) and bins.nb <= 511

    def _tail_buffers(self, nd: int, list_rows: int, Cn: int, wsp: Optional[dict]):
        nper = (nd + list_rows - 1) // list_rows
        nch = 2 if list_rows <= 16 else 4
        lists = self._buf(wsp, "tails", (N_DOY, nper, nch, Cn, 8), torch.int16, self.device)
        aux = self._buf(wsp, "tails_aux", (N_DOY, Cn), torch.int32, self.device)
        return lists, aux

    def tail_extract(self, anom: torch.Tensor, dcal: DeviceCalendar, bins: BinTable, wsp: Optional[dict] = None,
                     list_rows: Optional[int] = None):
        """Sorted key lists of every (dayofyear, cell) bucket of ``anom`` (include/marex_hip.h, TAILS)."""
        self._bind_stream()
        T_out, Cn = anom.shape
        edges = self.bin_tables(bins)[0]
        nd = int(np.diff(dcal.plan.doy_start).max())
        list_rows = int(list_rows or self.LIST_ROWS_EXTRACT)
        lists, aux = self._tail_buffers(nd, list_rows, Cn, wsp)
        rc = self.lib.marex_tail_extract_f32(
            self.ctx.handle, anom.data_ptr(), T_out, Cn, dcal.doy_start.data_ptr(), dcal.doy_rows.data_ptr(), nd,
            edges.data_ptr(), bins.nb, list_rows, lists.data_ptr(), aux.data_ptr(),
        )
        self.ctx.check(rc, "marex_tail_extract_f32")
        return {"tails": lists, "aux": aux, "max_bucket": nd, "list_rows": list_rows, "_keep": edges}

    def shifting_tails_ok(self, dcal: DeviceCalendar) -> bool:
        """The anomaly kernel emits its own tails for buckets of at most 6 lists of 15 rows (option SHIFT_TAILS=0: never)."""
        nd = int(np.diff(dcal.plan.doy_start).max())
        return bool(self.ctx_opt("SHIFT_TAILS", 1)) and nd <= 6 * self.LIST_ROWS_SHIFT

    def shifting_baseline_tails(self, x: torch.Tensor, dcal: DeviceCalendar, W: int, S: int, bins: BinTable,
                                wsp: Optional[dict] = None) -> Dict[str, object]:
        """Anomaly stage emitting the sorted key lists (TAILS) of its own output: no bin matrix, no extraction pass."""
        self._bind_stream()
        assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
        T, Cn = x.shape
        cal = dcal.plan
        if cal.T != T:
            raise ProcessingError("calendar length does not match the time axis of x")
        if cal.has_duplicates:
            raise ConfigurationError(
                "shifting_baseline needs at most one timestep per (year, dayofyear)",
                details="sub-daily time axes are not supported by the device path",
            )
        T_out = cal.T_out
        out = self._buf(wsp, "anom", (T_out, Cn), torch.float32, self.device)
        mask = self._buf(wsp, "mask", (Cn,), torch.uint8, self.device)
        invalid = self._buf(wsp, "invalid", (Cn,), torch.int32, self.device)
        invalid.zero_()
        edges = self.bin_tables(bins)[0]
        nd = int(np.diff(cal.doy_start).max())
        lists, aux = self._tail_buffers(nd, self.LIST_ROWS_SHIFT, Cn, wsp)
        rc = self.lib.marex_shifting_baseline_tails_f32(
            self.ctx.handle, x.data_ptr(), T, Cn, dcal.year_plan.data_ptr(), cal.n_cal_years, int(W), int(S), edges.data_ptr(),
            bins.nb, T_out, out.data_ptr(), mask.data_ptr(), invalid.data_ptr(), dcal.doy_start.data_ptr(),
            dcal.doy_rows.data_ptr(), nd, lists.data_ptr(), aux.data_ptr(),
        )
        self.ctx.check(rc, "marex_shifting_baseline_tails_f32")
        return {"out": out, "mask": mask, "invalid_count": invalid, "_keep": edges,
                "tails": {"tails": lists, "aux": aux, "max_bucket": nd, "list_rows": self.LIST_ROWS_SHIFT, "_keep": edges}}

    def hobday_thresholds_tails(self, tl: dict, anom: torch.Tensor, dcal: DeviceCalendar, bins: BinTable, q: float, wd: int,
                                ws: int, ny: int, nx: int, rows: Optional[tuple] = None, wsp: Optional[dict] = None):
        self._bind_stream()
        T_out, Cn = anom.shape
        row0, row1 = rows if rows is not None else (0, max(ny, 1))
        thr = self._buf(wsp, "thr_doy_major", (N_DOY, Cn), torch.float32, self.device)
        stats = self._buf(wsp, "thr_stats", (8,), torch.int32, self.device)  # marex_thr_stats: 8 x uint32
        stats.zero_()
        stats[0:1].fill_(-1)
        centres = self.bin_tables(bins)[1]
        rc = self.lib.marex_hobday_thresholds_tails_f32(
            self.ctx.handle, tl["tails"].data_ptr(), tl["aux"].data_ptr(), int(tl["list_rows"]), anom.data_ptr(), T_out, Cn,
            int(ny), int(nx), int(tl["max_bucket"]), centres.data_ptr(), bins.nb, float(q), int(wd), int(ws),
            float(bins.lower_bound), float(bins.upper_bound), int(row0), int(row1), thr.data_ptr(), stats.data_ptr(),
        )
        self.ctx.check(rc, "marex_hobday_thresholds_tails_f32")
        return {"thr_doy_major": thr, "stats_dev": stats, "_keep": centres}

    def mask_ge_doy_tails(self, tl: dict, anom: torch.Tensor, thr_doy_major: torch.Tensor, dcal: DeviceCalendar,
                          bins: BinTable, cells: Optional[tuple] = None, wsp: Optional[dict] = None):
        self._bind_stream()
        T_out, Cn = anom.shape
        c0, c1 = cells if cells is not None else (0, Cn)
        ext = self._buf(wsp, "extreme", (T_out, Cn), torch.uint8, self.device)
        n_true = self._buf(wsp, "n_true", (1,), torch.int64, self.device)
        n_true.zero_()
        edges = self.bin_tables(bins)[0]
        rc = self.lib.marex_mask_ge_doy_tails_f32(
            self.ctx.handle, tl["tails"].data_ptr(), tl["aux"].data_ptr(), int(tl["list_rows"]), int(tl["max_bucket"]),
            anom.data_ptr(), edges.data_ptr(), bins.nb, thr_doy_major.data_ptr(), dcal.doy_start.data_ptr(),
            dcal.doy_rows.data_ptr(), T_out, Cn, int(c0), int(c1), ext.data_ptr(), n_true.data_ptr(),
        )
        self.ctx.check(rc, "marex_mask_ge_doy_tails_f32")
        return {"extreme": ext, "n_true": n_true}

    def hobday_approx(self, anom: torch.Tensor, dcal: DeviceCalendar, bins: BinTable, q: float, wd: int, ws: int, ny: int,
                      nx: int, rows: Optional[tuple] = None, cells: Optional[tuple] = None, wsp: Optional[dict] = None,
                      binsb: Optional[torch.Tensor] = None, tails: Optional[dict] = None) -> Dict[str, object]:
        """Approximate Hobday thresholds + extreme mask of an anomaly field (detect.py:1957-2004): through tails when
        ``tails_plan`` takes the configuration, else through the bin matrix (``binsb``, made here when missing)."""
        K = self.tails_plan(dcal, bins, q, wd, ws, anom.shape[1])
        if K is not None:
            tl = tails if tails is not None else self.tail_extract(anom, dcal, bins, wsp=wsp)
            t = self.hobday_thresholds_tails(tl, anom, dcal, bins, q, wd, ws, ny, nx, rows=rows, wsp=wsp)
            m = self.mask_ge_doy_tails(tl, anom, t["thr_doy_major"], dcal, bins, cells=cells, wsp=wsp)
            keep = (tl, t["_keep"])
        else:
            if binsb is None:
                binsb = self.digitize(anom, dcal, bins, wsp=wsp)
            t = self.hobday_thresholds(binsb, anom, dcal, bins, q, wd, ws, ny, nx, rows=rows, wsp=wsp)
            m = self.mask_ge_doy(anom, t["thr_doy_major"], dcal, cells=cells, wsp=wsp, binned=(binsb, bins))
            keep = (binsb, t["_keep"])
        return {"thr_doy_major": t["thr_doy_major"], "stats_dev": t["stats_dev"], "extreme": m["extreme"], "n_true": m["n_true"],
                "path": "tails" if K is not None else "bins", "_keep": keep}

    # ------------------------------------------------------------------ stage a3 verdict
    def validation_summary(self, mask: torch.Tensor, invalid_count: torch.Tensor, cells: Optional[tuple] = None,
                           wsp: Optional[dict] = None) -> torch.Tensor:
        """Device int64 ``[n_ocean, invalid_total, invalid_cells, max_invalid]`` over the (owned) cells: the numbers of
        ``_validate_data_values`` (detect.py:205-279) from the per-cell outputs of the anomaly kernels, one launch."""
        self._bind_stream()
        Cn = mask.shape[-1]
        c0, c1 = cells if cells is not None else (0, Cn)
        out = self._buf(wsp, "validation_summary", (4,), torch.int64, self.device)
        rc = self.lib.marex_validation_summary(self.ctx.handle, mask.data_ptr(), invalid_count.data_ptr(), int(c0), int(c1),
                                               out.data_ptr())
        self.ctx.check(rc, "marex_validation_summary")
        return out

    # ------------------------------------------------------------------ stage a9 compare
    def mask_ge_doy(
        self, anom: torch.Tensor, thr_doy_major: torch.Tensor, dcal: DeviceCalendar, cells: Optional[tuple] = None,
        wsp: Optional[dict] = None, binned: Optional[tuple] = None,
    ) -> Dict[str, torch.Tensor]:
        """``cells=(c0, c1)`` restricts compare / write / count to the owned cells of a shard.  ``binned=(bin matrix,
        BinTable)`` of these anomalies lets the kernel decide most samples from their 2-byte bin (same result)."""
        self._bind_stream()
        T_out, Cn = anom.shape
        c0, c1 = cells if cells is not None else (0, Cn)
        ext = self._buf(wsp, "extreme", (T_out, Cn), torch.uint8, self.device)
        n_true = self._buf(wsp, "n_true", (1,), torch.int64, self.device)
        n_true.zero_()
        if binned is not None and binned[0] is not None:
            edges = self.bin_tables(binned[1])[0]
            rc = self.lib.marex_mask_ge_doy_bins_f32(
                self.ctx.handle, anom.data_ptr(), binned[0].data_ptr(), edges.data_ptr(), int(binned[1].nb),
                thr_doy_major.data_ptr(), dcal.doy_start.data_ptr(), dcal.doy_rows.data_ptr(), T_out, Cn, int(c0), int(c1),
                ext.data_ptr(), n_true.data_ptr(),
            )
            self.ctx.check(rc, "marex_mask_ge_doy_bins_f32")
        else:
            rc = self.lib.marex_mask_ge_doy_f32(
                self.ctx.handle, anom.data_ptr(), thr_doy_major.data_ptr(), dcal.doy_start.data_ptr(),
                dcal.doy_rows.data_ptr(), T_out, Cn, int(c0), int(c1), ext.data_ptr(), n_true.data_ptr(),
            )
            self.ctx.check(rc, "marex_mask_ge_doy_f32")
        return {"extreme": ext, "n_true": n_true}

    def transpose(self, a: torch.Tensor, wsp: Optional[dict] = None, name: str = "transposed") -> torch.Tensor:
        self._bind_stream()
        rows, cols = a.shape
        out = self._buf(wsp, name, (cols, rows), torch.float32, self.device)
        self.ctx.check(self.lib.marex_transpose_f32(self.ctx.handle, a.data_ptr(), rows, cols, out.data_ptr()), "marex_transpose_f32")
        return out

    # ------------------------------------------------------------------ whole path (shifting + hobday approx)
    def shifting_hobday(
        self,
        x: torch.Tensor,
        dcal: DeviceCalendar,
        *,
        W: int,
        S: int,
        bins: BinTable,
        q: float,
        wd: int,
        ws: int,
        ny: int,
        nx: int,
        transpose_thresholds: bool = True,
        own_rows: Optional[tuple] = None,
        workspace: Optional[dict] = None,
    ) -> Dict[str, object]:
        """validation + anomaly + thresholds + mask for ``shifting_baseline`` / ``hobday_extreme`` (approximate).

        ``own_rows=(row0, row1)``: the field is a latitude shard with overlap rows; thresholds and the
        mask are produced for the owned rows only (:mod:`marex_amd.dist`).
        """
        K = self.tails_plan(dcal, bins, q, wd, ws, x.shape[1])
        fam = shifting_kernel_family(int(W), int(S), int(x.shape[1]), K is not None)
        if fam != "lean":
            _note_path(f"shifting_baseline: window_year_baseline={W}, smooth_days_baseline={S}, {x.shape[1]} cells take the "
                       f"'{fam}' anomaly kernel (tuned path: smooth_days_baseline=21, window_year_baseline in (5, 15), cells a "
                       "multiple of 4; same results)")
        if K is None:
            nd = int(np.diff(dcal.plan.doy_start).max())
            _note_path(f"hobday_extreme: dayofyear buckets of {nd} rows with window_spatial_hobday={ws}, q={q} take the bin-matrix "
                       "threshold kernels (sorted key lists need >= 24 rows per bucket or no spatial pooling; same results)")
        if K is not None and self.shifting_tails_ok(dcal):
            a = self.shifting_baseline_tails(x, dcal, W, S, bins, wsp=workspace)
        else:
            a = self.shifting_baseline(x, dcal, W, S, bins if K is None else None, wsp=workspace)
        cells = None if own_rows is None else (own_rows[0] * nx, own_rows[1] * nx)
        h = self.hobday_approx(a["out"], dcal, bins, q, wd, ws, ny, nx, rows=own_rows, cells=cells, wsp=workspace,
                               binsb=a.get("bins"), tails=a.get("tails"))
        res = {
            "dat_anomaly": a["out"],
            "mask": a["mask"],
            "invalid_count": a["invalid_count"],
            "thr_doy_major": h["thr_doy_major"],
            "stats_dev": h["stats_dev"],
            "extreme_events": h["extreme"],
            "n_true": h["n_true"],
            "path": h["path"],
            "_keep": (a["_keep"], h["_keep"]),
        }
        if transpose_thresholds:
            res["thresholds"] = self.transpose(h["thr_doy_major"], wsp=workspace, name="thresholds")
        return res

    # ------------------------------------------------------------------ stage a13 (fixed baseline)
    def fixed_baseline(
        self,
        x: torch.Tensor,
        dcal: DeviceCalendar,
        reference_period=None,
        bins: Optional[BinTable] = None,
        count_invalid: bool = True,
        wsp: Optional[dict] = None,
        sub: Optional[torch.Tensor] = None,
        second_stage: bool = False,
        tails_bins: Optional[BinTable] = None,
    ) -> Dict[str, torch.Tensor]:
        """``x - nanmean_doy(x)`` for all timesteps (detect.py:2299-2397); ``dcal`` must be an untrimmed calendar.
        ``sub`` ``[C]``: a per-cell value taken off ``x`` on load (the deferred residual mean of :meth:`detrend`).
        ``second_stage``: called on the output of :meth:`detrend` with a shared workspace -- its mask / count buffers get
        names of their own, so that the first stage's validation outputs (the RAW field's) survive.
        ``tails_bins``: also leave the sorted key lists of the output (``"tails"``, 32 rows per list: what
        :meth:`tail_extract` would make of it) when the register kernel takes the shape -- see :meth:`fused_tails_ok`."""
        self._bind_stream()
        T, Cn = x.shape
        cal = dcal.plan
        assert cal.T == T and cal.T_out == T, "fixed_baseline needs an untrimmed calendar"
        use = None
        if reference_period is not None:
            use = self._dev(((cal.year >= reference_period[0]) & (cal.year <= reference_period[1])).astype(np.uint8))
        out = self._buf(wsp, "anom", (T, Cn), torch.float32, self.device)
        mask = self._buf(wsp, "mask2" if second_stage else "mask", (Cn,), torch.uint8, self.device)
        invalid = self._buf(wsp, "invalid2" if second_stage else "invalid", (Cn,), torch.int32, self.device)
        invalid.zero_()
        if tails_bins is not None and bins is None and self.fused_tails_ok(dcal, tails_bins):
            if sub is not None:
                assert sub.dtype == torch.float32 and sub.numel() == Cn
            edges = self.bin_tables(tails_bins)[0]
            nd = int(np.diff(cal.doy_start).max())
            lists, aux = self._tail_buffers(nd, self.LIST_ROWS_EXTRACT, Cn, wsp)
            rc = self.lib.marex_fixed_baseline_tails_f32(
                self.ctx.handle, x.data_ptr(), sub.data_ptr() if sub is not None else None, nd, T, Cn, dcal.doy_start.data_ptr(),
                dcal.doy_rows.data_ptr(), use.data_ptr() if use is not None else None, edges.data_ptr(), tails_bins.nb,
                out.data_ptr(), mask.data_ptr(), invalid.data_ptr() if count_invalid else None, lists.data_ptr(), aux.data_ptr(),
            )
            self.ctx.check(rc, "marex_fixed_baseline_tails_f32")
            if use is not None:
                self.sync()
            return {"out": out, "mask": mask, "invalid_count": invalid,
                    "tails": {"tails": lists, "aux": aux, "max_bucket": nd, "list_rows": self.LIST_ROWS_EXTRACT, "_keep": edges}}
        if bins is not None:
            edges = self.bin_tables(bins)[0]
            binsb = self._buf(wsp, "bins", self.bins_shape(T, Cn), torch.int16, self.device)
            e_ptr, b_ptr, nb = edges.data_ptr(), binsb.data_ptr(), bins.nb
        else:
            binsb, e_ptr, b_ptr, nb = None, None, None, 0
        tail = (T, Cn, dcal.doy_start.data_ptr(), dcal.doy_rows.data_ptr(), use.data_ptr() if use is not None else None, e_ptr, nb,
                out.data_ptr(), b_ptr, mask.data_ptr(), invalid.data_ptr() if count_invalid else None)
        if sub is not None:
            assert sub.dtype == torch.float32 and sub.numel() == Cn
        nd = int(np.diff(cal.doy_start).max())
        rc = self.lib.marex_fixed_baseline_sub_f32(self.ctx.handle, x.data_ptr(), sub.data_ptr() if sub is not None else None, nd, *tail)
        self.ctx.check(rc, "marex_fixed_baseline_f32")
        if use is not None:
            self.sync()  # the small table must outlive the kernel
        res = {"out": out, "mask": mask, "invalid_count": invalid}
        if binsb is not None:
            res["bins"] = binsb
        return res

    # ------------------------------------------------------------------ stage a10 binning on its own
    def digitize(self, anom: torch.Tensor, dcal: DeviceCalendar, bins: BinTable, wsp: Optional[dict] = None) -> torch.Tensor:
        """Dayofyear-sorted bin matrix of an anomaly field (rows with ``rowb_index < 0`` are skipped)."""
        self._bind_stream()
        T, Cn = anom.shape
        edges = self.bin_tables(bins)[0]
        binsb = self._buf(wsp, "bins", self.bins_shape(dcal.plan.T_out, Cn), torch.int16, self.device)
        rc = self.lib.marex_digitize_f32(
            self.ctx.handle, anom.data_ptr(), T, Cn, dcal.rowb_index.data_ptr(), edges.data_ptr(), bins.nb,
            dcal.plan.T_out, binsb.data_ptr(),
        )
        self.ctx.check(rc, "marex_digitize_f32")
        return binsb

    # ------------------------------------------------------------------ stage a12 (detrend)
    def detrend(
        self,
        x: torch.Tensor,
        model: np.ndarray,
        pmodel: np.ndarray,
        force_zero_mean: bool,
        bins_and_cal=None,
        count_invalid: bool = True,
        wsp: Optional[dict] = None,
        defer_mean: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """Residual of the least-squares fit of ``model`` (detect.py:2143-2224); optional binning of the result.
        ``defer_mean`` (with ``force_zero_mean``): ``out`` keeps its mean and ``res["mean"]`` ``[C]`` is the value still to be
        subtracted -- :meth:`fixed_baseline` takes it as ``sub`` and saves a pass over the field."""
        self._bind_stream()
        T, Cn = x.shape
        n_coef = int(model.shape[0])
        pm = self._dev(np.ascontiguousarray(pmodel, dtype=np.float64))
        mt = self._dev(np.ascontiguousarray(model.T, dtype=np.float64))
        out = self._buf(wsp, "detrended", (T, Cn), torch.float32, self.device)
        mask = self._buf(wsp, "mask", (Cn,), torch.uint8, self.device)
        invalid = self._buf(wsp, "invalid", (Cn,), torch.int32, self.device)
        mean = None
        if defer_mean and force_zero_mean:
            mean = self._buf(wsp, "detrend_mean", (Cn,), torch.float32, self.device)
            rc = self.lib.marex_detrend_deferred_mean_f32(
                self.ctx.handle, x.data_ptr(), T, Cn, pm.data_ptr(), mt.data_ptr(), n_coef, out.data_ptr(), mean.data_ptr(),
                mask.data_ptr(), invalid.data_ptr(),
            )
        else:
            rc = self.lib.marex_detrend_f32(
                self.ctx.handle, x.data_ptr(), T, Cn, pm.data_ptr(), mt.data_ptr(), n_coef, int(bool(force_zero_mean)),
                out.data_ptr(), mask.data_ptr(), invalid.data_ptr(),
            )
        self.ctx.check(rc, "marex_detrend_f32")
        self.sync()  # pm / mt must outlive the kernel
        res = {"out": out, "mask": mask, "invalid_count": invalid}
        if mean is not None:
            res["mean"] = mean
        if bins_and_cal is not None and bins_and_cal[0] is not None:
            res["bins"] = self.digitize(out, bins_and_cal[1], bins_and_cal[0], wsp=wsp)
        return res

    def detrend_fixed_baseline(self, x: torch.Tensor, model: np.ndarray, pmodel: np.ndarray, force_zero_mean: bool,
                               dcal: DeviceCalendar, reference_period=None, wsp: Optional[dict] = None,
                               tails_bins: Optional[BinTable] = None) -> Dict[str, torch.Tensor]:
        """``detrend_fixed_baseline`` (detect.py:2400-2462): residual of the fit minus its daily climatology.  One chain on the
        device when the model has at most 5 terms and dayofyear buckets at most 128 rows (the residual field is never
        materialised: 3 reads and 1 write of the field), otherwise the two stages one after the other.  Same bits either way."""
        T, Cn = x.shape
        cal = dcal.plan
        n_coef = int(model.shape[0])
        nd = int(np.diff(cal.doy_start).max())
        if n_coef > 5 or nd > 128 or not self.ctx_opt("DETREND_FUSED", 1):
            d = self.detrend(x, model, pmodel, bool(force_zero_mean), None, count_invalid=True, wsp=wsp, defer_mean=True)
            r = self.fixed_baseline(d["out"], dcal, reference_period, None, count_invalid=False, wsp=wsp, sub=d.get("mean"),
                                    second_stage=True, tails_bins=tails_bins)
            res = {"out": r["out"], "mask": d["mask"], "invalid_count": d["invalid_count"]}
            if "tails" in r:
                res["tails"] = r["tails"]
            return res
        self._bind_stream()
        assert x.dtype == torch.float32 and x.is_contiguous() and cal.T == T and cal.T_out == T
        pm = self._dev(np.ascontiguousarray(pmodel, dtype=np.float64))
        mt_host = np.ascontiguousarray(model.T, dtype=np.float64)
        mt = self._dev(mt_host)
        mts = self._dev(np.ascontiguousarray(mt_host[cal.doy_rows]))  # model rows in dayofyear-sorted row order
        use = None
        if reference_period is not None:
            use = self._dev(((cal.year >= reference_period[0]) & (cal.year <= reference_period[1])).astype(np.uint8))
        out = self._buf(wsp, "anom", (T, Cn), torch.float32, self.device)
        mask = self._buf(wsp, "mask", (Cn,), torch.uint8, self.device)
        invalid = self._buf(wsp, "invalid", (Cn,), torch.int32, self.device)
        res = {"out": out, "mask": mask, "invalid_count": invalid}
        if tails_bins is not None and self.fused_tails_ok(dcal, tails_bins):
            edges = self.bin_tables(tails_bins)[0]
            lists, aux = self._tail_buffers(nd, self.LIST_ROWS_EXTRACT, Cn, wsp)
            rc = self.lib.marex_detrend_fixed_baseline_tails_f32(
                self.ctx.handle, x.data_ptr(), T, Cn, pm.data_ptr(), mt.data_ptr(), mts.data_ptr(), n_coef, int(bool(force_zero_mean)),
                dcal.doy_start.data_ptr(), dcal.doy_rows.data_ptr(), use.data_ptr() if use is not None else None, nd,
                edges.data_ptr(), tails_bins.nb, out.data_ptr(), mask.data_ptr(), invalid.data_ptr(), lists.data_ptr(), aux.data_ptr(),
            )
            self.ctx.check(rc, "marex_detrend_fixed_baseline_tails_f32")
            res["tails"] = {"tails": lists, "aux": aux, "max_bucket": nd, "list_rows": self.LIST_ROWS_EXTRACT, "_keep": edges}
        else:
            rc = self.lib.marex_detrend_fixed_baseline_f32(
                self.ctx.handle, x.data_ptr(), T, Cn, pm.data_ptr(), mt.data_ptr(), mts.data_ptr(), n_coef, int(bool(force_zero_mean)),
                dcal.doy_start.data_ptr(), dcal.doy_rows.data_ptr(), use.data_ptr() if use is not None else None, nd,
                out.data_ptr(), mask.data_ptr(), invalid.data_ptr(),
            )
            self.ctx.check(rc, "marex_detrend_fixed_baseline_f32")
        self.sync()  # pm / mt / use must outlive the kernels
        return res

    # ------------------------------------------------------------------ stage a9 exact Hobday
    def std_normalise(self, anom: torch.Tensor, dcal: DeviceCalendar, window: int = 30,
                      wsp: Optional[dict] = None) -> Dict[str, torch.Tensor]:
        """``dat_stn`` and ``STD`` of the std_normalise branch (detect.py:2257-2278): day-of-year standard deviation,
        wrapped ``window``-day rolling RMS of it, anomaly / STD.  ``STD`` is returned dayofyear-major ``[366, C]``."""
        self._bind_stream()
        T, Cn = anom.shape
        std_day = self._buf(wsp, "std_day", (N_DOY, Cn), torch.float32, self.device)
        std_roll = self._buf(wsp, "std_roll", (N_DOY, Cn), torch.float32, self.device)
        out = self._buf(wsp, "dat_stn", (T, Cn), torch.float32, self.device)
        rc = self.lib.marex_std_rolling_doy_f32(
            self.ctx.handle, anom.data_ptr(), T, Cn, dcal.doy_start.data_ptr(), dcal.doy_rows.data_ptr(), int(window),
            std_day.data_ptr(), std_roll.data_ptr(),
        )
        self.ctx.check(rc, "marex_std_rolling_doy_f32")
        rc = self.lib.marex_div_doy_f32(
            self.ctx.handle, anom.data_ptr(), std_roll.data_ptr(), dcal.doy_start.data_ptr(), dcal.doy_rows.data_ptr(),
            T, Cn, out.data_ptr(),
        )
        self.ctx.check(rc, "marex_div_doy_f32")
        return {"dat_stn": out, "STD": std_roll}

    # ------------------------------------------------------------------ tracker pre-processing
    def fill_holes(self, data_bin: torch.Tensor, mask: torch.Tensor, ny: int, nx: int, R_fill: int,
                   regional_mode: bool = False, wsp: Optional[dict] = None, name: str = "filled") -> torch.Tensor:
        """Binary closing + opening with a disk of radius ``R_fill`` per timestep, land masked (track.py:1520-1676).
        ``data_bin``: uint8 ``[T, ny*nx]`` (0/1), ``mask``: uint8 ``[ny*nx]``."""
        self._bind_stream()
        T, Cn = data_bin.shape
        assert Cn == ny * nx
        out = self._buf(wsp, name, (T, Cn), torch.uint8, self.device)
        rc = self.lib.marex_fill_holes_u8(self.ctx.handle, data_bin.data_ptr(), mask.data_ptr(), T, int(ny), int(nx),
                                          int(R_fill), int(bool(regional_mode)), out.data_ptr())
        self.ctx.check(rc, "marex_fill_holes_u8")
        return out

    def fill_time_gaps(self, data_bin: torch.Tensor, mask: torch.Tensor, ny: int, nx: int, R_fill: int, T_fill: int,
                       regional_mode: bool = False, wsp: Optional[dict] = None) -> torch.Tensor:
        """Temporal closing over ``T_fill + 1`` steps, then ``fill_holes(R_fill // 2)`` (track.py:1678-1726)."""
        if T_fill == 0:
            return data_bin
        self._bind_stream()
        T, Cn = data_bin.shape
        tmp = self._buf(wsp, "time_closed", (T, Cn), torch.uint8, self.device)
        rc = self.lib.marex_time_closing_u8(self.ctx.handle, data_bin.data_ptr(), T, Cn, int(T_fill), tmp.data_ptr())
        self.ctx.check(rc, "marex_time_closing_u8")
        return self.fill_holes(tmp, mask, ny, nx, int(R_fill) // 2, regional_mode, wsp=wsp, name="gap_filled")

    def label_objects_2d(self, data_bin: torch.Tensor, ny: int, nx: int, wrap_x: bool = True,
                         wsp: Optional[dict] = None) -> Dict[str, torch.Tensor]:
        """Per-timestep 8-connected components (track.py:2013-2031): ``labels`` int32 ``[T, C]`` (1 + smallest linear
        index of the component, 0 = background) and ``areas`` int32 ``[T, C]`` (cell count, stored at the root cell)."""
        self._bind_stream()
        T, Cn = data_bin.shape
        labels = self._buf(wsp, "labels", (T, Cn), torch.int32, self.device)
        areas = self._buf(wsp, "areas", (T, Cn), torch.int32, self.device)
        rc = self.lib.marex_label2d_i32(self.ctx.handle, data_bin.data_ptr(), T, int(ny), int(nx), int(bool(wrap_x)),
                                        labels.data_ptr(), areas.data_ptr())
        self.ctx.check(rc, "marex_label2d_i32")
        return {"labels": labels, "areas": areas}

    def filter_small_objects(self, data_bin: torch.Tensor, ny: int, nx: int, area_filter_quartile: float = 0.5,
                             area_filter_absolute: Optional[float] = None, regional_mode: bool = False,
                             wsp: Optional[dict] = None) -> Dict[str, object]:
        """Remove the objects smaller than a percentile (or an absolute number) of cells (track.py:1755-1911, gridded).
        Series with more than 2^31 - 2 cells are labelled in time blocks (objects never span timesteps here)."""
        T, Cn = data_bin.shape
        tb = max(1, min(T, (2**31 - 2) // Cn))
        blocks = []
        for i, t0 in enumerate(range(0, T, tb)):
            sub = None if wsp is None else wsp.setdefault(f"ccl{i}", {})
            lab = self.label_objects_2d(data_bin[t0:t0 + tb], ny, nx, wrap_x=not regional_mode, wsp=sub)
            blocks.append((t0, lab["labels"], lab["areas"]))
        per_block = [a.reshape(-1)[a.reshape(-1) > 0] for _, _, a in blocks]  # ordered by (time, first cell)
        obj_areas = torch.cat(per_block)
        n_before = int(obj_areas.numel())
        if n_before == 0:
            raise ProcessingError("No objects found for area-based filtering")
        if area_filter_absolute is not None:
            thr = float(area_filter_absolute)
        else:  # np.percentile(areas, 100 q), "linear": two order statistics from a device sort, NumPy's lerp on the host
            srt = torch.sort(obj_areas).values
            virt = (n_before - 1) * float(np.float64(area_filter_quartile * 100.0) / 100.0)
            lo = int(np.floor(virt))
            g = virt - lo
            hi = min(lo + 1, n_before - 1)
            a, b = float(srt[lo].item()), float(srt[hi].item())
            thr = a + (b - a) * g if g < 0.5 else b - (b - a) * (1.0 - g)
        out = self._buf(wsp, "filtered", tuple(data_bin.shape), torch.uint8, self.device)
        dropped = False  # the reference's `object_ids_keep[0] = -1`: the first object of the whole list is never kept
        n_after = int((obj_areas.to(torch.float64) >= thr).sum().item())
        for (t0, labels, areas), pa in zip(blocks, per_block):
            first = 0
            if not dropped and pa.numel() > 0:
                flat = areas.reshape(-1)
                root = int(torch.nonzero(flat > 0)[0].item())
                first = root + 1
                if float(flat[root].item()) >= thr:
                    n_after -= 1
                dropped = True
            rc = self.lib.marex_filter_by_area_u8(self.ctx.handle, labels.data_ptr(), areas.data_ptr(), labels.numel(), thr,
                                                  first, out[t0:t0 + labels.shape[0]].data_ptr())
            self.ctx.check(rc, "marex_filter_by_area_u8")
        return {"filtered": out, "area_threshold": thr, "object_areas": obj_areas, "n_before": n_before, "n_after": n_after,
                "labels": blocks[0][1] if len(blocks) == 1 else [b[1] for b in blocks]}

    def fill_holes_mesh(self, data_bin: torch.Tensor, mask: torch.Tensor, nbr: torch.Tensor, R_fill: int,
                        wsp: Optional[dict] = None) -> torch.Tensor:
        """``fill_holes`` on an unstructured mesh (track.py:1543-1606): ``nbr`` int32 ``[3, C]``, 0-based, -1 = none."""
        self._bind_stream()
        T, Cn = data_bin.shape
        out = self._buf(wsp, "filled_mesh", (T, Cn), torch.uint8, self.device)
        rc = self.lib.marex_fill_holes_mesh_u8(self.ctx.handle, data_bin.data_ptr(), mask.data_ptr(), nbr.data_ptr(), T, Cn,
                                               int(R_fill), out.data_ptr())
        self.ctx.check(rc, "marex_fill_holes_mesh_u8")
        return out

    def filter_small_objects_mesh(self, data_bin: torch.Tensor, mask: torch.Tensor, nbr: torch.Tensor,
                                  area_filter_quartile: float = 0.5, area_filter_absolute: Optional[float] = None,
                                  wsp: Optional[dict] = None) -> Dict[str, object]:
        """``filter_small_objects`` on an unstructured mesh (track.py:1776-1857): sizes in cells, percentile over the
        clusters larger than 50 (5) cells, keep STRICTLY larger than the threshold."""
        self._bind_stream()
        T, Cn = data_bin.shape
        labels = self._buf(wsp, "labels_mesh", (T, Cn), torch.int32, self.device)
        areas = self._buf(wsp, "areas_mesh", (T, Cn), torch.int32, self.device)
        rc = self.lib.marex_label_mesh_i32(self.ctx.handle, data_bin.data_ptr(), mask.data_ptr(), nbr.data_ptr(), T, Cn,
                                           labels.data_ptr(), areas.data_ptr())
        self.ctx.check(rc, "marex_label_mesh_i32")
        flat = areas.reshape(-1)
        big = flat[flat > (5 if area_filter_absolute is not None else 50)]
        n_before = int(big.numel())
        if n_before == 0:
            raise ProcessingError("No objects found for area-based filtering")
        if area_filter_absolute is not None:
            thr = float(area_filter_absolute)
        else:
            srt = torch.sort(big).values
            virt = (n_before - 1) * float(np.float64(area_filter_quartile * 100) / 100.0)
            lo = int(np.floor(virt))
            g = virt - lo
            hi = min(lo + 1, n_before - 1)
            a, b = float(srt[lo].item()), float(srt[hi].item())
            thr = a + (b - a) * g if g < 0.5 else b - (b - a) * (1.0 - g)
        out = self._buf(wsp, "filtered_mesh", (T, Cn), torch.uint8, self.device)
        # strict ">" : areas are integers, so "> thr" == ">= floor(thr) + 1"
        rc = self.lib.marex_filter_by_area_u8(self.ctx.handle, labels.data_ptr(), areas.data_ptr(), labels.numel(),
                                              float(np.floor(thr) + 1.0), 0, out.data_ptr())
        self.ctx.check(rc, "marex_filter_by_area_u8")
        return {"filtered": out, "area_threshold": thr, "object_areas": big, "n_before": n_before,
                "n_after": int((big.to(torch.float64) > thr).sum().item()), "labels": labels}

    def hobday_thresholds_exact(self, anom: torch.Tensor, dcal: DeviceCalendar, percentile: float, wd: int,
                                wsp: Optional[dict] = None) -> torch.Tensor:
        """``np.nanpercentile`` per (dayofyear window, cell), float32, layout ``[366, C]`` (detect.py:1921-1956)."""
        self._bind_stream()
        T_out, Cn = anom.shape
        nd = np.diff(dcal.plan.doy_start).astype(np.int64)
        half = int(wd) // 2
        ext = np.concatenate([nd[-half:], nd, nd[:half]]) if half else nd
        max_rows = int(np.convolve(ext, np.ones(wd, dtype=np.int64), mode="valid").max())
        q32 = np.float32(percentile) / np.float32(100)  # NumPy's own float32 quantile
        thr = self._buf(wsp, "thr_doy_major", (N_DOY, Cn), torch.float32, self.device)
        overflow = torch.zeros((1,), dtype=torch.int32, device=self.device)
        rc = self.lib.marex_hobday_exact_f32(
            self.ctx.handle, anom.data_ptr(), T_out, Cn, dcal.doy_start.data_ptr(), dcal.doy_rows.data_ptr(),
            max(max_rows, 1), float(q32), float(q32), int(wd), thr.data_ptr(), overflow.data_ptr(),
        )
        self.ctx.check(rc, "marex_hobday_exact_f32")
        if int(overflow.item()) != 0:
            raise ProcessingError("exact Hobday percentile: selection buffer overflow (internal sizing error)")
        return thr

    # ------------------------------------------------------------------ stage a14 global thresholds
    def global_threshold(self, anom: torch.Tensor, percentile: float, method_percentile: str, bins: Optional[BinTable]):
        """Per-cell constant threshold, float64 ``[C]`` (detect.py:2873-2912) + warning statistics."""
        from .binning import global_bins

        self._bind_stream()
        T_out, Cn = anom.shape
        thr = torch.empty((Cn,), dtype=torch.float64, device=self.device)
        q = float(percentile) / 100.0
        if method_percentile == "exact":
            rc = self.lib.marex_global_threshold_f32(
                self.ctx.handle, anom.data_ptr(), T_out, Cn, q, 1, None, None, 0, 0.0, 0.0, thr.data_ptr(), None, None
            )
            self.ctx.check(rc, "marex_global_threshold_f32")
            return {"thr_f64": thr, "stats": {"n_too_low": 0, "n_too_high": 0, "min": float("nan"), "max": float("nan")}}
        gb = global_bins(bins.precision, bins.max_anomaly)
        edges = self._dev(gb.edges.astype(np.float64))
        centres = self._dev(gb.centres.astype(np.float64))
        stats = torch.zeros((8,), dtype=torch.int32, device=self.device)
        minmax = torch.tensor([float("inf"), float("-inf")], dtype=torch.float64, device=self.device)
        rc = self.lib.marex_global_threshold_f32(
            self.ctx.handle, anom.data_ptr(), T_out, Cn, q, 0, edges.data_ptr(), centres.data_ptr(), gb.nb,
            float(gb.lower_bound), float(gb.upper_bound), thr.data_ptr(), stats.data_ptr(), minmax.data_ptr(),
        )
        self.ctx.check(rc, "marex_global_threshold_f32")
        self.sync()
        s = stats.cpu().numpy().view(np.uint32)
        mm = minmax.cpu().numpy()
        return {
            "thr_f64": thr,
            "stats": {
                "n_too_low": int(s[2]), "n_too_high": int(s[3]),
                "min": float(mm[0]) if np.isfinite(mm[0]) else float("nan"),
                "max": float(mm[1]) if np.isfinite(mm[1]) else float("nan"),
            },
        }

    def mask_ge_const(self, anom: torch.Tensor, thr_f64: torch.Tensor, wsp: Optional[dict] = None) -> Dict[str, torch.Tensor]:
        self._bind_stream()
        T_out, Cn = anom.shape
        ext = self._buf(wsp, "extreme", (T_out, Cn), torch.uint8, self.device)
        n_true = self._buf(wsp, "n_true", (1,), torch.int64, self.device)
        n_true.zero_()
        rc = self.lib.marex_mask_ge_const_f32(
            self.ctx.handle, anom.data_ptr(), thr_f64.data_ptr(), T_out, Cn, ext.data_ptr(), n_true.data_ptr()
        )
        self.ctx.check(rc, "marex_mask_ge_const_f32")
        return {"extreme": ext, "n_true": n_true}
