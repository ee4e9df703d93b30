"""Host-side calendar tables for the ``preprocess_data`` hot path.

The device kernels never see dates.  Everything that depends on the calendar
(`dt.year`, `dt.dayofyear`, which timesteps feed which climatology window,
which rows survive the shifting-baseline trim, decimal years for the detrend
model) is computed here once with NumPy/pandas and handed to the kernels as
small integer tables.

Reference semantics restated here
---------------------------------
* year / dayofyear labels:          marEx/detect.py:1605-1614
* shifting-baseline window          marEx/detect.py:1622-1640  (years ``Y-W .. Y-1``,
  only target years ``>= min_year + W``)
* trim of the first ``W`` years     marEx/detect.py:615-641
* day-of-year window wrap (366)     marEx/detect.py:1929-1934, 2494-2496
* decimal year                      marEx/detect.py:2031-2058
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

N_DOY = 366  # day-of-year axis is always 1..366 (detect.py:1665, 2644)


def _to_year_doy(time) -> Tuple[np.ndarray, np.ndarray]:
    """``dt.year`` / ``dt.dayofyear`` of a time coordinate (datetime64 / DatetimeIndex / list of dates)."""
    import pandas as pd

    idx = pd.DatetimeIndex(np.asarray(time))
    return np.asarray(idx.year, dtype=np.int32), np.asarray(idx.dayofyear, dtype=np.int16)


@dataclass
class CalendarPlan:
    """Integer tables describing one time axis.

    Attributes
    ----------
    year, doy : [T] calendar labels of every input timestep.
    min_year, n_cal_years : first calendar year and span ``max_year-min_year+1``.
    tindex : [n_cal_years, 366] int32 -- timestep holding (year, doy) or -1.
    kept : [T] bool -- rows that are part of the output.
    out_index : [T] int32 -- output row of timestep t (time order) or -1.
    rowb_index : [T] int32 -- position of timestep t in the doy-sorted order of the kept rows, or -1.
    doy_start : [367] int32 -- ``doy_rows[doy_start[d-1]:doy_start[d]]`` are the kept rows with dayofyear d.
    doy_rows : [T_out] int32 -- output rows sorted by (dayofyear, time).
    doy_out : [T_out] int16 -- dayofyear of every output row.
    first_valid_year_idx : first calendar-year index with a climatology (``W`` for shifting baseline, 0 otherwise).
    """

    year: np.ndarray
    doy: np.ndarray
    min_year: int
    n_cal_years: int
    tindex: np.ndarray
    kept: np.ndarray
    out_index: np.ndarray
    rowb_index: np.ndarray
    doy_start: np.ndarray
    doy_rows: np.ndarray
    doy_out: np.ndarray
    first_valid_year_idx: int
    has_duplicates: bool = False
    time: Optional[np.ndarray] = field(default=None, repr=False)

    @property
    def T(self) -> int:
        return int(self.year.shape[0])

    @property
    def T_out(self) -> int:
        return int(self.doy_rows.shape[0])

    @property
    def year_idx(self) -> np.ndarray:
        return (self.year - self.min_year).astype(np.int32)

    @property
    def n_years_present(self) -> int:
        return int(np.unique(self.year).size)

    def year_plan(self) -> np.ndarray:
        """``[n_cal_years, 366, 4]`` int32 ``{timestep, output row, bin-matrix row, 0}`` (-1 = none): everything the
        shifting-baseline kernel needs to know about one (calendar year, dayofyear) in a single 16-byte load."""
        plan = np.full((self.n_cal_years, N_DOY, 4), -1, dtype=np.int32)
        plan[..., 3] = 0
        t = self.tindex
        have = t >= 0
        plan[..., 0] = t
        plan[..., 1][have] = self.out_index[t[have]]
        plan[..., 2][have] = self.rowb_index[t[have]]
        return plan


def build_calendar(time=None, *, year=None, doy=None, window_year_baseline: Optional[int] = None) -> CalendarPlan:
    """Build the tables for a time axis.

    ``window_year_baseline`` (``W``) switches on the shifting-baseline trim:
    output rows are those with ``year >= min_year + W`` (detect.py:638-641).
    With ``W=None`` every row is kept (fixed / detrended baselines).
    """
    if time is not None:
        year, doy = _to_year_doy(time)
        time = np.asarray(time)
    else:
        year = np.asarray(year, dtype=np.int32)
        doy = np.asarray(doy, dtype=np.int16)
    if year.ndim != 1 or year.shape != doy.shape or year.size == 0:
        raise ValueError("year/doy must be non-empty 1-D arrays of equal length")
    if doy.min() < 1 or doy.max() > N_DOY:
        raise ValueError("dayofyear labels must lie in 1..366")

    T = year.size
    min_year = int(year.min())
    max_year = int(year.max())
    n_cal = max_year - min_year + 1
    yi = (year - min_year).astype(np.int64)

    tindex = np.full((n_cal, N_DOY), -1, dtype=np.int32)
    flat = yi * N_DOY + (doy.astype(np.int64) - 1)
    # Later duplicates of one (year, doy) label would overwrite earlier ones: detect and flag.
    uniq, counts = np.unique(flat, return_counts=True)
    has_dup = bool((counts > 1).any())
    tindex.reshape(-1)[flat] = np.arange(T, dtype=np.int32)

    if window_year_baseline is not None:
        W = int(window_year_baseline)
        kept = year >= (min_year + W)
        first_valid = W
    else:
        kept = np.ones(T, dtype=bool)
        first_valid = 0

    out_index = np.full(T, -1, dtype=np.int32)
    n_out = int(kept.sum())
    out_index[kept] = np.arange(n_out, dtype=np.int32)

    doy_out = doy[kept].astype(np.int16)
    # stable sort keeps time order inside every dayofyear bucket
    doy_rows = np.argsort(doy_out, kind="stable").astype(np.int32)
    doy_start = np.zeros(N_DOY + 1, dtype=np.int32)
    np.cumsum(np.bincount(doy_out.astype(np.int64) - 1, minlength=N_DOY), out=doy_start[1:])
    rowb_of_out = np.empty(n_out, dtype=np.int32)
    rowb_of_out[doy_rows] = np.arange(n_out, dtype=np.int32)
    rowb_index = np.full(T, -1, dtype=np.int32)
    rowb_index[kept] = rowb_of_out

    return CalendarPlan(
        year=year,
        doy=doy,
        min_year=min_year,
        n_cal_years=n_cal,
        tindex=tindex,
        kept=kept,
        out_index=out_index,
        rowb_index=rowb_index,
        doy_start=doy_start,
        doy_rows=doy_rows,
        doy_out=doy_out,
        first_valid_year_idx=first_valid,
        has_duplicates=has_dup,
        time=time,
    )


_PLAN_ARRAYS = ("year", "doy", "tindex", "kept", "out_index", "rowb_index", "doy_start", "doy_rows", "doy_out")
_PLAN_SCALARS = ("min_year", "n_cal_years", "first_valid_year_idx", "has_duplicates")


def plan_tables(cal: CalendarPlan, prefix: str = "cal.") -> dict:
    """A plan as a flat ``{name: array or scalar}`` dict -- what rank 0 broadcasts (marex_amd.dist.broadcast_tables)."""
    out = {prefix + k: getattr(cal, k) for k in _PLAN_ARRAYS}
    out.update({prefix + k: getattr(cal, k) for k in _PLAN_SCALARS})
    if cal.time is not None:
        out[prefix + "time"] = np.asarray(cal.time)
    return out


def plan_from_tables(tables: dict, prefix: str = "cal.") -> CalendarPlan:
    """Inverse of :func:`plan_tables`: nothing is derived again, the received arrays ARE the plan."""
    kw = {k: tables[prefix + k] for k in _PLAN_ARRAYS}
    kw.update({k: tables[prefix + k] for k in _PLAN_SCALARS})
    kw["min_year"], kw["n_cal_years"] = int(kw["min_year"]), int(kw["n_cal_years"])
    kw["first_valid_year_idx"], kw["has_duplicates"] = int(kw["first_valid_year_idx"]), bool(kw["has_duplicates"])
    return CalendarPlan(time=tables.get(prefix + "time"), **kw)


def decimal_year(time) -> np.ndarray:
    """``year + days_elapsed / days_in_year`` as float64 (detect.py:2051-2057)."""
    import pandas as pd

    t = pd.DatetimeIndex(np.asarray(time))
    start = pd.to_datetime(t.year.astype(str) + "-01-01")
    nxt = pd.to_datetime((t.year + 1).astype(str) + "-01-01")
    elapsed = (t - start).days
    duration = (nxt - start).days
    return np.asarray(t.year + elapsed / duration, dtype=np.float64)


def detrend_model(dy: np.ndarray, detrend_orders, remove_harmonics: bool) -> Tuple[np.ndarray, np.ndarray]:
    """Model rows and pseudo-inverse of the detrend fit (detect.py:2143-2169).

    Returns ``(model [n_coef, T], pmodel [T, n_coef])`` in float64.
    """
    dy = np.asarray(dy, dtype=np.float64)
    rows = [np.ones(dy.size)]
    centred = dy - np.mean(dy)
    for order in detrend_orders:
        rows.append(centred**order)
    if remove_harmonics:
        rows.extend([np.sin(2 * np.pi * dy), np.cos(2 * np.pi * dy), np.sin(4 * np.pi * dy), np.cos(4 * np.pi * dy)])
    model = np.array(rows)
    for i in range(1, model.shape[0]):
        model[i] = model[i] - np.mean(model[i]) * model[0]
    pmodel = np.linalg.pinv(model)
    return model, pmodel


def daily_time_axis(start: str, periods: int) -> np.ndarray:
    """Proleptic-Gregorian daily axis used by the synthetic workloads."""
    t0 = np.datetime64(start, "D")
    return t0 + np.arange(periods).astype("timedelta64[D]")
