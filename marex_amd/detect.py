"""Drop-in host API of the hot path: ``preprocess_data`` and its public sub-functions.

Same call signatures, option names, Dataset schema, attrs and error behaviour as the reference
(marEx/detect.py:287-313, 891-907, 1119-1133, 1511-1517, 1691-1698); the numerics run on the MI355X
through :class:`marex_amd.engine.HotPath` (HIP kernels behind the C ABI of include/marex_hip.h).
There is no CPU fallback: without a HIP device the compute entry points raise.

Deliberate differences:
* in-memory (NumPy-backed) inputs are accepted -- the reference insists on Dask-backed arrays because it
  builds a Dask graph (detect.py:558-568); Dask-backed xarray inputs are materialised with ``.values``;
* ``dask_chunks`` / ``use_temp_checkpoints`` are accepted and ignored (identity on values);
* ``window_days_hobday=1`` raises a ConfigurationError (it crashes the reference).
"""

from __future__ import annotations

import logging
import warnings
from typing import Dict, List, Literal, Optional, Tuple

import numpy as np

from . import binning, calendar
from .exceptions import ConfigurationError, DataValidationError, ProcessingError, create_data_validation_error
from .xr_compat import DataArray, Dataset, coord_values, to_numpy

logger = logging.getLogger("marex_amd")

_ANOMALY_METHODS = ["detrend_harmonic", "shifting_baseline", "fixed_baseline", "detrend_fixed_baseline"]
_EXTREME_METHODS = ["global_extreme", "hobday_extreme"]

_engine_cache: Dict[object, object] = {}


def get_engine(device: int = 0, replica: int = 0):
    """The per-device :class:`~marex_amd.engine.HotPath` (created on first use; raises without a GPU).  ``replica`` > 0:
    further independent engines (own context and stream) on the same device -- ``devices=[0, 0]`` drives one card from two
    host threads."""
    key = device if replica == 0 else (device, replica)
    if key not in _engine_cache:
        from .engine import HotPath

        _engine_cache[key] = HotPath(device, own_stream=replica > 0)
    return _engine_cache[key]


# ======================================================================================
# validation helpers (detect.py:53-202)
# ======================================================================================
def _validate_dimensions_exist(da, dimensions: Dict[str, str]) -> None:
    missing = [f"'{actual}' (for {concept})" for concept, actual in dimensions.items() if actual not in da.dims]
    if missing:
        available = list(da.dims)
        raise create_data_validation_error(
            f"Missing required dimensions: {', '.join(missing)}",
            details=f"Dataset has dimensions: {available}",
            suggestions=["Check dimension names in your data", "Update the 'dimensions' parameter to match your data structure"],
            data_info={"missing_dimensions": missing, "available_dimensions": available, "provided_dimensions": dimensions},
        )


def _validate_coordinates_exist(da, coordinates: Dict[str, str]) -> None:
    missing = [f"'{actual}' (for {concept})" for concept, actual in coordinates.items() if actual not in da.coords]
    if missing:
        available = list(da.coords.keys())
        raise create_data_validation_error(
            f"Missing required coordinates: {', '.join(missing)}",
            details=f"Dataset has coordinates: {available}",
            suggestions=["Check coordinate names in your data", "Update the 'coordinates' parameter to match your data structure"],
            data_info={"missing_coordinates": missing, "available_coordinates": available, "provided_coordinates": coordinates},
        )


def _infer_dims_coords(da, dimensions: Optional[Dict[str, str]], coordinates: Optional[Dict[str, str]]):
    """Defaults and existence checks of the dimension / coordinate name maps (detect.py:131-202)."""
    if dimensions is None:
        dimensions = {"time": "time", "x": "lon", "y": "lat"}
    if "time" not in dimensions:
        dimensions = {"time": "time", **dimensions}
    if coordinates is None:
        if "y" not in dimensions:
            raise create_data_validation_error(
                "Coordinates parameter must be explicitly specified for unstructured data",
                details="Unstructured data requires coordinate names for x and y spatial coordinates",
                suggestions=["Example: coordinates={'time': 'time', 'x': 'lon', 'y': 'lat'}"],
                data_info={"data_structure": "unstructured (2D)", "dimensions": dimensions},
            )
        coordinates = dimensions.copy()
    elif "time" not in coordinates:
        coordinates = {"time": dimensions.get("time", "time"), **coordinates}
    _validate_dimensions_exist(da, dimensions)
    _validate_coordinates_exist(da, coordinates)
    return dimensions, coordinates


def _get_preprocessing_steps(
    method_anomaly: str,
    method_extreme: str,
    std_normalise: bool,
    detrend_orders: List[int],
    window_year_baseline: int,
    smooth_days_baseline: int,
    window_days_hobday: int,
    window_spatial_hobday: Optional[int],
    reference_period: Optional[Tuple[int, int]] = None,
) -> List[str]:
    """Provenance strings stored in ``attrs['preprocessing_steps']`` (detect.py:844-888; golden-pinned)."""
    steps: List[str] = []
    ref = f"{reference_period[0]}-{reference_period[1]}" if reference_period is not None else None
    if method_anomaly == "detrend_harmonic":
        steps.append(f"Removed polynomial trend orders={detrend_orders} & seasonal cycle")
        if std_normalise:
            steps.append("Normalised by 30-day rolling STD")
    elif method_anomaly == "shifting_baseline":
        steps += [f"Rolling climatology using {window_year_baseline} years", f"Smoothed with {smooth_days_baseline}-day window"]
    elif method_anomaly == "fixed_baseline":
        steps.append(f"Daily climatology computed from {ref}" if ref else "Daily climatology computed from full time series")
    elif method_anomaly == "detrend_fixed_baseline":
        steps.append(f"Removed polynomial trend orders={detrend_orders}")
        steps.append(
            f"Daily climatology computed from detrended data ({ref})" if ref else "Daily climatology computed from detrended data"
        )
    if method_extreme == "global_extreme":
        steps.append("Global percentile threshold applied to all days")
    elif method_extreme == "hobday_extreme":
        txt = f"Day-of-year thresholds with {window_days_hobday} day window"
        if window_spatial_hobday is not None:
            txt += f" & {window_spatial_hobday} spatial neighbours"
        steps.append(txt)
    return steps


# ======================================================================================
# array <-> labelled-array plumbing
# ======================================================================================
class _Field:
    """A DataArray flattened to the device layout ``[T, C]`` plus what is needed to rebuild labelled outputs."""

    def __init__(self, da, dimensions: Dict[str, str], coordinates: Dict[str, str]):
        self.dimensions, self.coordinates = dimensions, coordinates
        self.tdim = dimensions["time"]
        self.gridded = "y" in dimensions
        self.sdims = [dimensions["y"], dimensions["x"]] if self.gridded else [dimensions["x"]]
        extra = [d for d in da.dims if d != self.tdim and d not in self.sdims]
        if extra:
            raise create_data_validation_error(
                f"Unsupported extra dimensions {extra}", details="expected (time, y, x) or (time, cells) data"
            )
        self.da = da
        dev = getattr(da, "device_tensor", None)
        if dev is not None:  # field already in HBM (marex_amd.zarr_io.open_dataarray_device): no host copy, no upload
            if tuple(da.dims) != (self.tdim, *self.sdims):
                raise create_data_validation_error("device-resident input must be laid out (time, y, x) / (time, cells)",
                                                   details=f"dims {tuple(da.dims)}")
            self.sshape = tuple(dev.shape[1:])
            self.x = None
            self._x_dev = dev.reshape(dev.shape[0], -1)
            self.shape = tuple(self._x_dev.shape)
        else:
            arr = to_numpy(da.transpose(self.tdim, *self.sdims))
            self.sshape = tuple(arr.shape[1:])
            self.x = np.ascontiguousarray(arr.reshape(arr.shape[0], -1), dtype=np.float32)  # cast as detect.py:600
            self._x_dev = None
            self.shape = tuple(self.x.shape)
        self.ny, self.nx = (self.sshape if self.gridded else (0, self.sshape[0]))
        self.time = coord_values(da, coordinates["time"])
        self.scoords = {}
        for d in self.sdims:
            if d in da.coords:
                self.scoords[d] = da.coords[d]
        for key in ("x", "y"):
            name = coordinates.get(key)
            if name is not None and name in da.coords and name not in self.scoords:
                self.scoords[name] = da.coords[name]

    def device_x(self, eng):
        """The field as a float32 ``[T, C]`` tensor on the engine's device (cast as detect.py:600)."""
        import torch

        if self._x_dev is not None:
            return self._x_dev.to(device=eng.device, dtype=torch.float32).contiguous()
        return _pipe(eng).upload(self.x, np.float32)

    def block(self, sh) -> "_FieldBlock":
        """The cells a spatial block ingests (``marex_amd.dist.Shard``): a latitude band with its overlap rows, or a range of
        cells -- contiguous on the cell axis either way."""
        return _FieldBlock(self, sh)

    def labelled(self, data: np.ndarray, lead: Optional[Tuple[str, np.ndarray]], trail: Optional[Tuple[str, np.ndarray]] = None):
        """Wrap ``data`` with dims ``(lead?, *spatial, trail?)``; ``data``'s cell axis is still flat."""
        dims, coords, shape = [], {}, []
        if lead is not None:
            dims.append(lead[0]); coords[lead[0]] = lead[1]; shape.append(len(lead[1]))
        dims += self.sdims
        shape += list(self.sshape)
        if trail is not None:
            dims.append(trail[0]); coords[trail[0]] = trail[1]; shape.append(len(trail[1]))
        coords.update(self.scoords)
        return DataArray(np.asarray(data).reshape(shape), dims=dims, coords=coords)


def _pipe(eng):
    """The engine's pinned-buffer transfer pipeline (marex_amd.transfer), created on first use."""
    if getattr(eng, "_pipe", None) is None:
        from .transfer import PinnedPipe

        import os

        # 16 host threads / 128 MiB chunks measured best on the MI355X box (161 GB of results in 4.7 s; first-touch page
        # faults of the destination arrays are part of what the threads share)
        threads = int(os.environ.get("MAREX_PIPE_THREADS", str(min(16, os.cpu_count() or 4))))
        eng._pipe = PinnedPipe(eng.device, chunk_bytes=int(os.environ.get("MAREX_PIPE_CHUNK_MB", "128")) << 20,
                               nbuf=int(os.environ.get("MAREX_PIPE_NBUF", "4")), threads=threads)
    return eng._pipe


class _FieldBlock:
    """What the device stages read of a ``_Field``, restricted to one spatial block."""

    def __init__(self, field: _Field, sh):
        self.parent, self.shard = field, sh
        self.c0, self.c1 = sh.cell_base, sh.cell_base + sh.cells_in
        self.time, self.gridded = field.time, field.gridded
        self.ny, self.nx = (sh.ny_in, field.nx) if field.gridded else (0, sh.cells_in)
        self.shape = (field.shape[0], sh.cells_in)

    def device_x(self, eng):
        import torch

        f = self.parent
        if f._x_dev is not None:
            return f._x_dev[:, self.c0:self.c1].to(device=eng.device, dtype=torch.float32).contiguous()
        return _pipe(eng).upload(f.x[:, self.c0:self.c1], np.float32)


def plan_blocks(field: _Field, eng, halo: int, per_cell_bytes: int, min_blocks: int = 1, engines=None):
    """Spatial blocks that fit the free HBM -- the device-side counterpart of the reference's Dask layout for this path
    (space chunked, ``time: -1``; detect.py:2617-2620, 785-792): latitude bands with ``halo`` overlap rows per interior
    side on grids, cell ranges on meshes.  ``MAREX_BLOCKS=n`` forces the number of blocks (tests; tuning).
    ``engines``: every engine that will hold a block at the same time; the budget of a block is then the smallest share any
    of them gets -- 80 % of the free memory of its card divided by the number of engines listed on that card."""
    import os

    import torch

    from .dist import plan_shards

    ny, nx = (field.ny, field.nx) if field.gridded else (0, field.nx)
    forced = int(os.environ.get("MAREX_BLOCKS", "0"))
    if forced > 0:
        n = min(max(forced, int(min_blocks)), ny if field.gridded else max(nx, 1))
        return plan_shards(ny, nx, max(n, 1), halo)
    if eng.device.type != "cuda":
        return plan_shards(ny, nx, 1, halo)
    torch.cuda.empty_cache()
    per_card: Dict[int, int] = {}
    for e in (engines or [eng]):
        per_card[e.device.index or 0] = per_card.get(e.device.index or 0, 0) + 1
    budget = min(int(torch.cuda.mem_get_info(torch.device("cuda", card))[0] * 0.8) // k for card, k in per_card.items())
    n = max(1, min(int(min_blocks), ny if field.gridded else nx))
    while True:
        shards = plan_shards(ny, nx, n, halo)
        if max(sh.cells_in for sh in shards) * per_cell_bytes <= budget:
            return shards
        if n >= (ny if field.gridded else nx):
            raise create_data_validation_error(
                "Field does not fit the device even one latitude row / cell at a time",
                details=f"{per_cell_bytes} bytes per cell, {budget} bytes of HBM free")
        n = min(max(n + 1, int(n * 1.25)), ny if field.gridded else nx)


def _raise_if_invalid(field: _Field, summary: Dict[str, int]) -> None:
    """Error texts of ``_validate_data_values`` (detect.py:224-279) from the device-side counts."""
    T, C = field.shape
    if summary["n_ocean"] == 0:
        raise create_data_validation_error(
            "Dataset contains no valid (finite) data",
            details="All values in the first time step are NaN or infinite",
            suggestions=["Check your input data for data quality issues", "Verify the data was loaded correctly"],
            data_info={"total_values": int(T * C), "total_spatial_locations": int(C)},
        )
    if summary["max_invalid"] > 0:
        raise create_data_validation_error(
            f"Dataset contains {summary['invalid_total']} invalid values in {summary['invalid_cells']} ocean locations",
            details=(
                f"Found invalid data across time series. Worst location has {summary['max_invalid']} "
                f"invalid time steps out of {T}."
            ),
            suggestions=[
                "Remove or interpolate NaN/infinite values before preprocessing",
                "For ocean data, ensure land mask is properly applied before preprocessing",
            ],
            data_info={
                "total_invalid_values_in_ocean": summary["invalid_total"],
                "locations_affected": summary["invalid_cells"],
                "total_ocean_locations": summary["n_ocean"],
                "max_invalid_at_one_location": summary["max_invalid"],
                "total_time_steps": int(T),
            },
        )


def _check_reference_period_allowed(reference_period, method_anomaly: str) -> None:
    if reference_period is not None and method_anomaly not in ("fixed_baseline", "detrend_fixed_baseline"):
        raise ConfigurationError(
            f"reference_period is not supported for method_anomaly='{method_anomaly}'",
            details="reference_period is only applicable to 'fixed_baseline' and 'detrend_fixed_baseline' methods",
            suggestions=["Remove the reference_period parameter, or", "Use method_anomaly='fixed_baseline' or 'detrend_fixed_baseline'"],
        )


def _check_detrend_orders(detrend_orders) -> None:
    """detect.py:2104-2126."""
    if not detrend_orders:
        raise ConfigurationError(
            "detrend_orders cannot be empty",
            details="At least one polynomial order must be specified for detrending",
            suggestions=["Use detrend_orders=[1] for linear detrending"],
        )
    bad = [o for o in detrend_orders if o < 1]
    if bad:
        raise ConfigurationError(
            f"Invalid polynomial orders: {bad}",
            details="Polynomial orders must be positive integers (≥ 1)",
            suggestions=["Use only positive integers for polynomial orders"],
        )


def _check_reference_period_values(reference_period, years: np.ndarray) -> None:
    """detect.py:2334-2355."""
    if reference_period is None:
        return
    a, b = reference_period
    if a > b:
        raise ConfigurationError(
            f"Invalid reference_period: start year ({a}) must be <= end year ({b})",
            details="The reference_period tuple must be (start_year, end_year) with start_year <= end_year",
            suggestions=[f"Swap the order: use reference_period=({b}, {a})"],
        )
    if not ((years >= a) & (years <= b)).any():
        lo, hi = int(years.min()), int(years.max())
        raise ConfigurationError(
            f"No data found in reference_period ({a}, {b})",
            details=f"Dataset spans {lo}-{hi} but no timesteps fall within the specified period",
            suggestions=[f"Adjust reference_period to overlap with data range ({lo}-{hi})", "Set reference_period=None to use the full time series"],
        )


def _validate_extreme_options(
    gridded: bool, method_extreme, threshold_percentile, window_days_hobday, window_spatial_hobday,
    method_percentile, precision, max_anomaly,
) -> Optional[int]:
    """Option checks of ``identify_extremes`` in the reference's order (detect.py:1277-1470).

    Returns the effective spatial window (5 on gridded data when ``None`` and hobday, detect.py:1451-1452).
    """
    if method_percentile not in ("exact", "approximate"):
        raise ConfigurationError(
            f"Unknown method_percentile '{method_percentile}'",
            details="Invalid method_percentile parameter",
            suggestions=["Use 'exact' for precise percentile computation (memory intensive)",
                         "Use 'approximate' for efficient histogram-based computation (default)"],
            context={"provided_method": method_percentile, "valid_methods": ["exact", "approximate"]},
        )
    if method_percentile == "exact":
        for pname, val, dflt in (("precision", precision, 0.01), ("max_anomaly", max_anomaly, 5.0)):
            if val != dflt:
                raise ConfigurationError(
                    f"Parameter '{pname}' cannot be used with method_percentile='exact'",
                    details=f"The {pname} parameter ({pname}={val}) is only used by the approximate histogram method",
                    suggestions=[f"Remove the '{pname}' parameter when using method_percentile='exact'"],
                    context={"method_percentile": method_percentile, f"provided_{pname}": val, f"default_{pname}": dflt},
                )
    if threshold_percentile < 60 and method_percentile == "approximate":
        raise ConfigurationError(
            f"Percentile threshold {threshold_percentile}% is not supported with method_percentile='approximate'",
            details="Low percentile thresholds (<60%) produce undefined and unsupported behaviour when using approximate histogram methods",
            suggestions=["Use method_percentile='exact' for percentiles below 60%"],
            context={"threshold_percentile": threshold_percentile, "method_percentile": method_percentile, "min_supported_percentile": 60},
        )
    if window_spatial_hobday is not None:
        if not gridded:
            raise ConfigurationError(
                "window_spatial_hobday is not supported for unstructured grids",
                details="Spatial smoothing with window_spatial_hobday requires structured grids with both x and y dimensions.",
                suggestions=["Remove the window_spatial_hobday parameter for unstructured grids"],
                context={"grid_type": "unstructured", "window_spatial_hobday": window_spatial_hobday},
            )
        if method_extreme != "hobday_extreme":
            raise ConfigurationError(
                "window_spatial_hobday can only be used with method_extreme='hobday_extreme'",
                details="The window_spatial_hobday parameter is only implemented for the Hobday extreme method.",
                suggestions=["Remove the window_spatial_hobday parameter when using method_extreme='global_extreme'"],
                context={"method_extreme": method_extreme, "window_spatial_hobday": window_spatial_hobday},
            )
        if method_percentile == "exact":
            raise ConfigurationError(
                "window_spatial_hobday is not supported with method_percentile='exact'",
                details="The window_spatial_hobday parameter is only implemented for the approximate percentile method.",
                suggestions=["Remove the window_spatial_hobday parameter when using method_percentile='exact'"],
                context={"method_percentile": method_percentile, "window_spatial_hobday": window_spatial_hobday},
            )
    if method_extreme == "hobday_extreme" and window_days_hobday is not None and window_days_hobday % 2 == 0:
        raise ConfigurationError(
            "window_days_hobday must be an odd number",
            details=f"window_days_hobday={window_days_hobday} is even, which would create asymmetric temporal windows.",
            suggestions=[f"Use window_days_hobday={window_days_hobday + 1} or {window_days_hobday - 1}"],
            context={"window_days_hobday": window_days_hobday, "is_odd": False},
        )
    ws_eff = window_spatial_hobday
    if method_extreme == "hobday_extreme" and ws_eff is None and gridded and method_percentile == "approximate":
        ws_eff = 5
    if method_extreme == "hobday_extreme" and window_spatial_hobday is None and gridded and method_percentile == "exact":
        ws_eff = None  # the default 5 is set (detect.py:1452) but the exact branch never pools
    if method_extreme == "hobday_extreme" and ws_eff is not None and ws_eff % 2 == 0:
        raise ConfigurationError(
            "window_spatial_hobday must be an odd number",
            details=f"window_spatial_hobday={ws_eff} is even, which would create asymmetric spatial windows.",
            suggestions=["Choose an odd number."],
            context={"window_spatial_hobday": ws_eff, "is_odd": False},
        )
    if method_extreme not in _EXTREME_METHODS:
        raise ConfigurationError(
            f"Unknown extreme method '{method_extreme}'",
            details="Invalid method_extreme parameter",
            suggestions=["Use 'global_extreme' for efficient constant percentile threshold",
                         "Use 'hobday_extreme' for day-of-year specific thresholds"],
            context={"provided_method": method_extreme, "valid_methods": _EXTREME_METHODS},
        )
    if method_extreme == "hobday_extreme" and method_percentile == "approximate" and window_days_hobday < 3:
        raise ConfigurationError(
            "window_days_hobday must be at least 3 with method_percentile='approximate'",
            details="a 1-day window makes the reference's wrap padding degenerate (it raises a broadcast ValueError there)",
            suggestions=["Use window_days_hobday >= 3"],
        )
    return ws_eff


# ======================================================================================
# core on flat arrays (device)
# ======================================================================================
def _anomaly_core(eng, field: _Field, method_anomaly, window_year_baseline, smooth_days_baseline, detrend_orders,
                  force_zero_mean, reference_period, want_bins: Optional[binning.BinTable]):
    """Runs the anomaly stage on the device.  Returns dict with device tensors + the calendar plan."""
    import torch

    x = field.device_x(eng)
    if method_anomaly == "shifting_baseline":
        cal = calendar.build_calendar(field.time, window_year_baseline=int(window_year_baseline))
        total_years = cal.n_cal_years
        if total_years < window_year_baseline:  # detect.py:622-636
            raise create_data_validation_error(
                "Insufficient data for shifting_baseline method",
                details=f"Dataset spans {total_years} years but requires at least {window_year_baseline} years",
                suggestions=["Use more years of data to meet minimum requirement",
                             f"Reduce window_year_baseline parameter (currently {window_year_baseline})"],
                data_info={"available_years": int(total_years), "required_years": int(window_year_baseline)},
            )
        if cal.T_out == 0:
            raise create_data_validation_error(
                "Insufficient data for shifting_baseline method",
                details=f"no timestep is left after removing the first {window_year_baseline} years",
            )
        dcal = eng.upload_calendar(cal)
        tails_bins = None
        if callable(want_bins):
            tails_bins = want_bins(dcal, "tails")  # bin table when the anomaly kernel should emit the sorted key lists itself
            want_bins = want_bins(dcal)
        from .engine import _note_path, shifting_kernel_family

        fam = shifting_kernel_family(int(window_year_baseline), int(smooth_days_baseline), int(x.shape[1]), tails_bins is not None)
        if fam != "lean":
            _note_path(f"shifting_baseline: window_year_baseline={window_year_baseline}, smooth_days_baseline={smooth_days_baseline}, "
                       f"{x.shape[1]} cells take the '{fam}' anomaly kernel (tuned path: smooth_days_baseline=21, "
                       "window_year_baseline in (5, 15), cells a multiple of 4; same results)")
        if tails_bins is not None:
            r = eng.shifting_baseline_tails(x, dcal, int(window_year_baseline), int(smooth_days_baseline), tails_bins)
        else:
            r = eng.shifting_baseline(x, dcal, int(window_year_baseline), int(smooth_days_baseline), want_bins)
        return {"anom": r["out"], "mask": r["mask"], "invalid": r["invalid_count"], "bins": r.get("bins"), "tails": r.get("tails"),
                "cal": cal, "dcal": dcal}
    cal = calendar.build_calendar(field.time)
    dcal = eng.upload_calendar(cal)
    # bin table when the threshold stage will work from sorted key lists: the fixed-baseline kernels then emit them themselves
    tails_bins = want_bins(dcal, "tails_fixed") if callable(want_bins) else None
    want_bins = want_bins(dcal) if callable(want_bins) else want_bins
    if method_anomaly == "fixed_baseline":
        _check_reference_period_values(reference_period, cal.year)
        r = eng.fixed_baseline(x, dcal, reference_period, want_bins, count_invalid=True, tails_bins=tails_bins)
    elif method_anomaly in ("detrend_harmonic", "detrend_fixed_baseline"):
        _check_detrend_orders(detrend_orders)
        if 1 not in detrend_orders and len(detrend_orders) > 1:
            print("Warning: Higher-order detrending without linear term may be unstable")  # detect.py:2135-2136
        harm = method_anomaly == "detrend_harmonic"
        model, pmodel = calendar.detrend_model(calendar.decimal_year(field.time), detrend_orders, harm)
        if method_anomaly == "detrend_fixed_baseline":
            _check_reference_period_values(reference_period, cal.year)
            if want_bins is None:  # one chain on the device; the residual field is never materialised
                r = eng.detrend_fixed_baseline(x, model, pmodel, bool(force_zero_mean), dcal, reference_period, tails_bins=tails_bins)
            else:  # a bin matrix is wanted: the two stages, the residual mean subtracted by the climatology kernel while it reads
                d = eng.detrend(x, model, pmodel, bool(force_zero_mean), None, count_invalid=True, defer_mean=True)
                r = eng.fixed_baseline(d["out"], dcal, reference_period, want_bins, count_invalid=False, sub=d.get("mean"))
                r["mask"], r["invalid_count"] = d["mask"], d["invalid_count"]
        else:
            r = eng.detrend(x, model, pmodel, bool(force_zero_mean), (want_bins, dcal), count_invalid=True)
    else:
        raise ConfigurationError(
            f"Unknown anomaly method '{method_anomaly}'",
            details="Invalid method_anomaly parameter",
            suggestions=["Use 'detrend_harmonic' for efficient processing with trend and harmonic removal",
                         "Use 'shifting_baseline' for accurate climatology (requires more data)",
                         "Use 'fixed_baseline' to remove a single daily climatology across all years",
                         "Use 'detrend_fixed_baseline' for trend removal followed by fixed climatology"],
            context={"provided_method": method_anomaly, "valid_methods": _ANOMALY_METHODS},
        )
    return {"anom": r["out"], "mask": r["mask"], "invalid": r["invalid_count"], "bins": r.get("bins"), "tails": r.get("tails"),
            "cal": cal, "dcal": dcal}


def _validation_summary(eng, a, own: Optional[slice] = None) -> Dict[str, int]:
    n = a["mask"].shape[-1]
    cells = (0, n) if own is None else own.indices(n)[:2]
    v = eng.validation_summary(a["mask"], a["invalid"], cells).cpu().tolist()
    return {"n_ocean": int(v[0]), "invalid_total": int(v[1]), "invalid_cells": int(v[2]), "max_invalid": int(v[3])}


class _Bounds:
    """What ``_warn_threshold_range`` reads of a bin table."""

    def __init__(self, lower_bound: float, upper_bound: float):
        self.lower_bound, self.upper_bound = lower_bound, upper_bound


def warn_threshold_stats(thr_stats, n_arrays: int, max_anomaly: float) -> None:
    """The reference's two threshold-range warnings (detect.py:2711-2730), once per threshold ARRAY of the field: ``thr_stats`` =
    ``(lower_bound, upper_bound, statistics)`` per block and array, in block order with the ``n_arrays`` arrays of a block
    (``thresholds``, ``thresholds_stn``) interleaved."""
    per_table: Dict[tuple, list] = {}
    for lo, hi, st in thr_stats:
        per_table.setdefault((lo, hi), []).append(st)
    for (lo, hi), sts in per_table.items():
        per_array = [sts[i::n_arrays] for i in range(n_arrays)] if len(sts) >= n_arrays else [sts]
        for group in per_array:
            mx = [g["max"] for g in group if g["max"] == g["max"]]
            mn = [g["min"] for g in group if g["min"] == g["min"]]
            _warn_threshold_range({"n_too_high": sum(g["n_too_high"] for g in group), "n_too_low": sum(g["n_too_low"] for g in group),
                                   "max": max(mx) if mx else float("nan"), "min": min(mn) if mn else float("nan")}, _Bounds(lo, hi), max_anomaly)


def _warn_threshold_range(stats: Dict[str, float], bt, max_anomaly: float) -> None:
    """The two UserWarnings of detect.py:2711-2730."""
    if stats["n_too_high"] > 0:
        warnings.warn(
            f"Quantile values exceed expected range: max={stats['max']:.4f} > {bt.upper_bound:.4f}. "
            f"Consider increasing max_anomaly parameter (currently {max_anomaly:.2f}) or using a lower percentile threshold.",
            UserWarning, stacklevel=3,
        )
    if stats["n_too_low"] > 0:
        warnings.warn(
            f"Quantile values below expected range in some locations: min={stats['min']:.4f} < {bt.lower_bound:.4f}. "
            "This is likely due to a constant anomaly in certain (e.g. due to sea ice). "
            "Double check the computed threshold values are correct.",
            UserWarning, stacklevel=3,
        )


def _extremes_core(eng, a, field, method_extreme, threshold_percentile, window_days_hobday, ws_eff,
                   method_percentile, bt: Optional[binning.BinTable], max_anomaly, wsp=None, rows=None, defer=None):
    """Threshold + mask stage on the device.  Returns (extreme uint8 [T', C], thresholds (host layout), dims tag).
    ``rows``: grid rows of a latitude band whose thresholds are wanted (the others are overlap rows); ``defer``: a dict
    that collects the threshold-range statistics instead of warning at once (block-wise runs warn once, at the end)."""
    cal, dcal = a["cal"], a["dcal"]

    def range_stats(stats, table):
        if defer is None:
            _warn_threshold_range(stats, table, max_anomaly)
        else:
            defer["stats"].append((stats, table))

    if method_extreme == "hobday_extreme":
        n_years = cal.n_years_present if cal.T_out == cal.T else int(np.unique(cal.year[cal.kept]).size)
        n_above = n_years * window_days_hobday * (ws_eff if ws_eff is not None else 1) ** 2 * (1.0 - threshold_percentile / 100.0)
        if n_above < 50 and not (defer or {}).get("logged"):  # detect.py:1905-1915
            if defer is not None:
                defer["logged"] = True
            logger.warning(
                f"Not enough samples for accurate extreme detection: {n_above} < 50. "
                "Consider using a lower threshold_percentile, increasing your time-series size, "
                "increasing the window_days_hobday, or using a larger window_spatial_hobday."
            )
        if method_percentile == "exact":
            thr_doy = eng.hobday_thresholds_exact(a["anom"], dcal, float(threshold_percentile), int(window_days_hobday))
            m = eng.mask_ge_doy(a["anom"], thr_doy, dcal)
            return m["extreme"], thr_doy, "doy_first", m["n_true"]
        t = eng.hobday_approx(
            a["anom"], dcal, bt, threshold_percentile / 100.0, int(window_days_hobday),
            int(ws_eff) if ws_eff else 1, field.ny, field.nx, rows=rows, binsb=a.get("bins"), tails=a.get("tails"),
        )
        thr = eng.transpose(t["thr_doy_major"])
        range_stats(eng.decode_thr_stats(t["stats_dev"]), bt)
        return t["extreme"], thr, "doy_last", t["n_true"]
    # global_extreme
    g = eng.global_threshold(a["anom"], float(threshold_percentile), method_percentile, bt)
    if method_percentile == "approximate":
        range_stats(g["stats"], binning.global_bins(bt.precision, bt.max_anomaly))
    m = eng.mask_ge_const(a["anom"], g["thr_f64"])
    return m["extreme"], g["thr_f64"], "none", m["n_true"]


# ======================================================================================
# public API
# ======================================================================================
def dataset_attrs(method_anomaly="shifting_baseline", method_extreme="hobday_extreme", threshold_percentile=95,
                  std_normalise=False, detrend_orders=None, window_year_baseline=15, smooth_days_baseline=21,
                  window_days_hobday=11, window_spatial_hobday=None, reference_period=None, force_zero_mean=True,
                  method_percentile="approximate", precision=0.01, max_anomaly=5.0, **_other) -> Dict[str, object]:
    """The Dataset attrs of ``preprocess_data`` exactly as detect.py:731-783 builds them -- a function of the options alone
    (defaults = those of ``preprocess_data``; options that leave no attr are ignored)."""
    if detrend_orders is None:
        detrend_orders = [1]
    attrs: Dict[str, object] = {
        "method_anomaly": method_anomaly,
        "method_extreme": method_extreme,
        "threshold_percentile": threshold_percentile,
        "preprocessing_steps": _get_preprocessing_steps(
            method_anomaly, method_extreme, std_normalise, detrend_orders, window_year_baseline,
            smooth_days_baseline, window_days_hobday, window_spatial_hobday, reference_period,
        ),
    }
    if method_anomaly == "detrend_harmonic":
        attrs.update({"detrend_orders": detrend_orders, "force_zero_mean": force_zero_mean, "std_normalise": std_normalise})
    elif method_anomaly == "shifting_baseline":
        attrs.update({"window_year_baseline": window_year_baseline, "smooth_days_baseline": smooth_days_baseline})
    elif method_anomaly == "fixed_baseline":
        if reference_period is not None:
            attrs["reference_period"] = list(reference_period)
    elif method_anomaly == "detrend_fixed_baseline":
        attrs.update({"detrend_orders": detrend_orders, "force_zero_mean": force_zero_mean})
        if reference_period is not None:
            attrs["reference_period"] = list(reference_period)
    if method_extreme == "hobday_extreme":
        attrs["window_days_hobday"] = window_days_hobday
    attrs.update({"method_percentile": method_percentile, "precision": precision, "max_anomaly": max_anomaly})
    return attrs


def preprocess_data(
    da,
    method_anomaly: Literal["detrend_harmonic", "shifting_baseline", "fixed_baseline", "detrend_fixed_baseline"] = "shifting_baseline",
    method_extreme: Literal["global_extreme", "hobday_extreme"] = "hobday_extreme",
    threshold_percentile: float = 95,
    window_year_baseline: int = 15,
    smooth_days_baseline: int = 21,
    window_days_hobday: int = 11,
    window_spatial_hobday: Optional[int] = None,
    std_normalise: bool = False,
    detrend_orders: Optional[List[int]] = None,
    force_zero_mean: bool = True,
    reference_period: Optional[Tuple[int, int]] = None,
    method_percentile: Literal["exact", "approximate"] = "approximate",
    precision: float = 0.01,
    max_anomaly: float = 5.0,
    dask_chunks: Optional[Dict[str, int]] = None,
    dimensions: Optional[Dict[str, str]] = None,
    coordinates: Optional[Dict[str, str]] = None,
    neighbours=None,
    cell_areas=None,
    use_temp_checkpoints: bool = False,
    verbose: Optional[bool] = None,
    quiet: Optional[bool] = None,
    device: int = 0,
    devices: Optional[List[int]] = None,
    require_dask: bool = False,
    _validation: str = "raise",
    _own_rows: Optional[Tuple[int, int]] = None,
    _defer_warnings: bool = False,
):
    """Anomalies, thresholds and the boolean extreme mask of a (time, [lat,] lon / cells) field.

    ``require_dask=True`` (extension; default False): refuse an input that is not Dask-backed with the reference's own
    ``DataValidationError`` ("Input DataArray must be Dask-backed", detect.py:558-568).  The eager call takes in-memory and
    device-resident arrays on purpose (documented deviation); a caller that wants the reference's contract to the letter --
    or that must not materialise a larger-than-memory field by accident -- switches it on, or uses
    :func:`marex_amd.dask_adapter.preprocess_data_lazy`, which always insists.

    ``_validation="return"`` (internal: :mod:`marex_amd.dask_adapter` runs one spatial block of a larger field per call): the
    verdict of ``_validate_data_values`` is not raised but returned as ``ds.attrs["_validation"]`` -- a block may be all land,
    and the error is about the whole field; ``_own_rows=(r0, r1)``: the grid rows (cells on a mesh) of the block its counts
    are about -- the others are overlap rows (the call then runs as ONE device block: a caller that cuts the field itself sizes its
    blocks to the HBM); ``_defer_warnings``: the threshold-range statistics (detect.py:2711-2730) are returned as
    ``ds.attrs["_thr_stats"]`` instead of being warned about, so that a caller with several blocks warns once for the field.

    ``devices=[0, 1, ...]`` (extension): the field is cut into at least that many spatial blocks (latitude bands
    with ``ws//2`` overlap rows / cell ranges -- cells are independent along time) and every listed device works through its
    share from its own host thread; nothing is exchanged between devices, the host stitches the Dataset.  The reference fans
    the same call out over a Dask cluster (helper.py:232-411).

    Which kernel family a configuration takes (same results on every path; ``ctx.set_option`` / ``MAREX_<NAME>`` force the
    other one, the bench line names the choice in ``config.histogram_representation``):

    * anomaly, ``shifting_baseline``: ``k_shift_fast`` for a gap-free daily calendar with ``smooth_days_baseline`` 21 and
      ``window_year_baseline`` in {3, 4, 5, 6, 7, 10, 13, 15}, or 11 / 15 days with 5 / 10 / 15 years, and the reference's
      ``arange`` bin table; every other width, window, table, or calendar with gaps: the general ``k_shifting`` (about 4x slower);
    * approximate Hobday thresholds: sorted key lists ("tails") when a dayofyear bucket holds at least 24 samples (24 output
      years) or there is no spatial pooling -- ``k_thr_tails`` (tiles, pooling) or ``k_thr_cells`` (no pooling and a window of
      at most 16 lists: 11 days x up to 15 years) -- otherwise (short buckets WITH pooling, e.g. 10 years of data) the bin matrix
      and ``k_thr_band``; more than 128 samples per bucket, more than 511 bins or ``window_spatial_hobday`` > 7: the general
      sliding-histogram kernel;
    * mask: from the lists (``k_mask_tails``) on the tails path, from the bin matrix for buckets of at least 24 rows, else the
      plain compare.

    Mirror of ``marEx.preprocess_data`` (detect.py:287-841).  Returns a Dataset with ``dat_anomaly``
    (float32), ``mask`` (bool), ``extreme_events`` (bool), ``thresholds`` (float32, dims
    ``(*space, dayofyear)`` for the approximate Hobday method, ``(dayofyear, *space)`` for the exact one,
    ``(*space)`` float64 for ``global_extreme``) and the reference's attrs.
    """
    if detrend_orders is None:
        detrend_orders = [1]
    if dask_chunks is None:
        dask_chunks = {"time": 25}
    if verbose:
        logger.setLevel(logging.DEBUG)
    elif quiet:
        logger.setLevel(logging.WARNING)
    logger.info(f"Starting data preprocessing - Method: {method_anomaly} -> {method_extreme}")

    dimensions, coordinates = _infer_dims_coords(da, dimensions, coordinates)
    if require_dask:  # detect.py:558-568, same place in the sequence of checks (after the dims / coords inference)
        data = getattr(da, "data", None)
        if not (hasattr(data, "__dask_graph__") or hasattr(data, "dask")):
            raise create_data_validation_error(
                "Input DataArray must be Dask-backed",
                details="Preprocessing requires chunked data for efficient computation",
                suggestions=["Convert to Dask array: da = da.chunk({'time': 30})",
                             "Load with chunking: xr.open_dataset('file.nc', chunks={'time': 30})"],
                data_info={"data_type": type(data).__name__, "shape": tuple(da.shape)},
            )
    _check_reference_period_allowed(reference_period, method_anomaly)
    if method_anomaly not in _ANOMALY_METHODS:
        raise ConfigurationError(
            f"Unknown anomaly method '{method_anomaly}'",
            details="Invalid method_anomaly parameter",
            suggestions=["Use 'detrend_harmonic', 'shifting_baseline', 'fixed_baseline' or 'detrend_fixed_baseline'"],
            context={"provided_method": method_anomaly, "valid_methods": _ANOMALY_METHODS},
        )
    gridded = "y" in dimensions
    ws_eff = _validate_extreme_options(
        gridded, method_extreme, threshold_percentile, window_days_hobday, window_spatial_hobday,
        method_percentile, precision, max_anomaly,
    )

    field = _Field(da, dimensions, coordinates)
    bt = binning.hobday_bins(precision, max_anomaly) if method_percentile == "approximate" else None
    need_bins = bt if (method_extreme == "hobday_extreme" and method_percentile == "approximate") else None
    want_stn = bool(std_normalise) and method_anomaly == "detrend_harmonic"

    # Spatial blocks sized to the free HBM (one block when everything fits): every stage is per cell along time except
    # the ws x ws pooling, so latitude bands carry ws//2 overlap rows and nothing is exchanged between blocks.
    T = field.shape[0]
    halo = (int(ws_eff) // 2) if (need_bins is not None and ws_eff) else 0
    per_cell = (4 * T + 7 * T) + (8 * T if method_anomaly.startswith("detrend") else 0) + (11 * T if want_stn else 0) + 366 * 24
    dev_list = [int(d) for d in devices] if devices else [int(device)]
    seen: Dict[int, int] = {}
    engines = []
    for dv in dev_list:  # the same card listed twice gets a second, independent engine (own context and stream)
        engines.append(get_engine(dv, seen.get(dv, 0)))
        seen[dv] = seen.get(dv, 0) + 1
    eng = engines[0]
    if _own_rows is not None:
        # the caller's block IS the device block: its overlap rows must not be cut again (they would be counted twice)
        from .dist import plan_shards as _plan_shards

        blocks = _plan_shards(field.ny if field.gridded else 0, field.nx, 1, halo)
        if eng.device.type == "cuda":
            import torch as _torch

            free = int(_torch.cuda.mem_get_info(eng.device)[0])
            if blocks[0].cells_in * int(per_cell * 1.25) > free:
                raise ProcessingError(
                    "Spatial block does not fit the device",
                    details=f"{blocks[0].cells_in} cells x {int(per_cell * 1.25)} bytes per cell, {free} bytes of HBM free",
                    suggestions=["use smaller blocks (block_rows / block_cells of preprocess_data_lazy)"],
                )
    else:
        blocks = plan_blocks(field, eng, halo, int(per_cell * 1.25), min_blocks=len(engines), engines=engines)
    single = len(blocks) == 1
    if not single:
        logger.info(f"Field processed in {len(blocks)} spatial blocks of <= {max(b.cells_in for b in blocks)} cells"
                    + (f" on {len(engines)} devices" if len(engines) > 1 else ""))

    import threading

    out: Dict[str, np.ndarray] = {}
    out_lock = threading.Lock()
    kinds: Dict[str, str] = {}
    logged = {"logged": False}
    total = {"n_ocean": 0, "invalid_total": 0, "invalid_cells": 0, "max_invalid": 0}
    n_true_total, cal = 0, None
    doy_axis = ("dayofyear", np.arange(1, calendar.N_DOY + 1))
    C_all = field.shape[1]

    def put(e, name, t, kind, sh):
        """Owned cells of a block result -> the host array of the whole field (``kind``: where the cell axis is)."""
        own = sh.own_cell_slice()
        g0 = sh.own0 * field.nx if field.gridded else sh.own0
        axis = {"cells_last": t.ndim - 1, "doy_last": 0, "doy_first": 1, "none": 0}[kind]
        view = t[(slice(None),) * axis + (own,)]
        with out_lock:
            if name not in out:
                shape = list(view.shape)
                if not single:
                    shape[axis] = C_all
                out[name] = np.empty(shape, dtype=np.dtype(str(t.dtype).replace("torch.", "")))
        dst = out[name] if single else out[name][(slice(None),) * axis + (slice(g0, g0 + view.shape[axis]),)]
        if view.dim() == 2 and axis == 1 and view.numel() * view.element_size() >= (32 << 20):
            e.sync()
            _pipe(e).download(view, dst)  # the big [T', cells] arrays: chunked through pinned buffers
        else:
            dst[...] = view.cpu().numpy()

    def run_block(e, sh, stop):
        """Anomaly + threshold + mask stages of one spatial block on engine ``e``; its owned cells go to the host arrays.
        Returns the block's validation counts, deferred warning statistics, calendar and extreme count."""
        fb = field if single else field.block(sh)
        rows = None if single or not field.gridded else (sh.own0 - sh.in0, sh.own1 - sh.in0)
        defer = {"stats": [], "logged": logged["logged"]}
        bins_for = None
        if need_bins is not None:
            # the anomaly kernels emit the bin matrix only where the threshold stage will use it (engine.tails_plan: the
            # tail kernels take long series and read the anomalies -- or the lists the anomaly kernel emits -- themselves)
            def bins_for(dcal, what="bins"):
                k = e.tails_plan(dcal, need_bins, threshold_percentile / 100.0, int(window_days_hobday),
                                 int(ws_eff) if ws_eff else 1, int(fb.shape[1]))
                if what == "tails":
                    return need_bins if (k is not None and e.shifting_tails_ok(dcal)) else None
                if what == "tails_fixed":
                    return need_bins if k is not None else None
                return need_bins if k is None else None

        a = _anomaly_core(e, fb, method_anomaly, window_year_baseline, smooth_days_baseline, detrend_orders,
                          force_zero_mean, reference_period, bins_for)
        own_cells = sh.own_cell_slice()
        if _own_rows is not None and single:
            own_cells = slice(_own_rows[0] * (field.nx if field.gridded else 1), _own_rows[1] * (field.nx if field.gridded else 1))
        part = _validation_summary(e, a, own_cells)
        res = {"part": part, "cal": a["cal"], "defer": defer, "n_true": 0, "kinds": {}}
        if _validation == "raise" and (part["max_invalid"] > 0 or stop["bad"] or (single and part["n_ocean"] == 0)):
            stop["bad"] = stop["bad"] or part["max_invalid"] > 0
            return res  # the run ends in the reference's validation error: only the counts of the other blocks matter
        ext, thr, thr_kind, n_true = _extremes_core(
            e, a, fb, method_extreme, threshold_percentile, window_days_hobday, ws_eff, method_percentile, bt, max_anomaly,
            rows=rows, defer=defer,
        )
        put(e, "dat_anomaly", a["anom"], "cells_last", sh)
        put(e, "mask", a["mask"], "cells_last", sh)
        put(e, "extreme_events", ext, "cells_last", sh)
        put(e, "thresholds", thr, thr_kind, sh)
        res["kinds"]["thresholds"] = thr_kind
        # standardised anomalies and their own extremes (detect.py:2257-2293, 686-715): detrend_harmonic only
        if want_stn:
            if sh is blocks[0]:
                logger.info("Processing standardised anomalies for extreme identification")
            sn = e.std_normalise(a["anom"], a["dcal"])
            a_stn = {"anom": sn["dat_stn"], "cal": a["cal"], "dcal": a["dcal"], "bins": None}
            ext_s, thr_s, kind_s, _ = _extremes_core(
                e, a_stn, fb, method_extreme, threshold_percentile, window_days_hobday, ws_eff, method_percentile, bt,
                max_anomaly, wsp={}, rows=rows, defer=defer,
            )
            put(e, "dat_stn", sn["dat_stn"], "cells_last", sh)
            put(e, "STD", e.transpose(sn["STD"], name="std_cells_major"), "doy_last", sh)  # flox appends the group dim
            put(e, "extreme_events_stn", ext_s, "cells_last", sh)
            put(e, "thresholds_stn", thr_s, kind_s, sh)
            res["kinds"]["thresholds_stn"] = kind_s
        res["n_true"] = int(n_true.item()) if single else 0
        logged["logged"] = logged["logged"] or defer["logged"]
        e.sync()
        return res

    stop = {"bad": False}
    results = [None] * len(blocks)
    if len(engines) == 1:
        for i, sh in enumerate(blocks):
            results[i] = run_block(eng, sh, stop)
    else:
        import torch

        def worker(k):
            e = engines[k]
            with torch.cuda.device(e.device), torch.cuda.stream(e.stream if e.stream is not None else torch.cuda.current_stream(e.device)):
                for i in range(k, len(blocks), len(engines)):  # block i on device i mod N, in order
                    results[i] = run_block(e, blocks[i], stop)

        errors = []

        def guarded(k):
            try:
                worker(k)
            except BaseException as exc:  # re-raised in the calling thread
                errors.append(exc)

        threads = [threading.Thread(target=guarded, args=(k,)) for k in range(len(engines))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if errors:
            raise errors[0]
    defer = {"stats": []}
    for res in results:  # block order, as a sequential run would have collected them
        for k in ("n_ocean", "invalid_total", "invalid_cells"):
            total[k] += res["part"][k]
        total["max_invalid"] = max(total["max_invalid"], res["part"]["max_invalid"])
        defer["stats"].extend(res["defer"]["stats"])
        kinds.update(res["kinds"])
        n_true_total += res["n_true"]
        cal = res["cal"]
    if _validation == "raise":
        _raise_if_invalid(field, total)
    if method_anomaly == "shifting_baseline":
        logger.info(f"Trimming data to start from {cal.min_year + window_year_baseline} (removing first {window_year_baseline} years)")
    # the reference's two threshold-range warnings, once per threshold array (detect.py:2711-2730)
    thr_stats = [(float(table.lower_bound), float(table.upper_bound), dict(st)) for st, table in defer["stats"]]
    if not _defer_warnings:
        warn_threshold_stats(thr_stats, 2 if want_stn else 1, max_anomaly)
    if not single:
        n_true_total = int(np.count_nonzero(out["extreme_events"]))

    def with_doy(name, kind):
        if kind == "doy_last":
            return field.labelled(out[name], None, doy_axis)
        if kind == "doy_first":
            return field.labelled(out[name], doy_axis)
        return field.labelled(out[name], None)

    time_out = field.time[cal.kept]
    tlead = (field.tdim, time_out)
    ds = Dataset()
    ds["dat_anomaly"] = field.labelled(out["dat_anomaly"], tlead)
    as_bool = lambda a: a.view(np.bool_) if a.dtype == np.uint8 else a.astype(bool)  # noqa: E731  (kernels write 0 / 1)
    ds["mask"] = field.labelled(as_bool(out["mask"]), None)
    ds["extreme_events"] = field.labelled(as_bool(out["extreme_events"]), tlead)
    ds["thresholds"] = with_doy("thresholds", kinds["thresholds"])
    if want_stn:
        ds["dat_stn"] = field.labelled(out["dat_stn"], tlead)
        ds["STD"] = with_doy("STD", "doy_last")
        ds["extreme_events_stn"] = field.labelled(as_bool(out["extreme_events_stn"]), tlead)
        ds["thresholds_stn"] = with_doy("thresholds_stn", kinds["thresholds_stn"])
    if neighbours is not None:
        ds["neighbours"] = neighbours.astype(np.int32)
    if cell_areas is not None:
        ds["cell_areas"] = cell_areas.astype(np.float32)

    ds.attrs.update(dataset_attrs(
        method_anomaly=method_anomaly, method_extreme=method_extreme, threshold_percentile=threshold_percentile,
        std_normalise=std_normalise, detrend_orders=detrend_orders, window_year_baseline=window_year_baseline,
        smooth_days_baseline=smooth_days_baseline, window_days_hobday=window_days_hobday,
        window_spatial_hobday=window_spatial_hobday, reference_period=reference_period, force_zero_mean=force_zero_mean,
        method_percentile=method_percentile, precision=precision, max_anomaly=max_anomaly))
    if _validation != "raise":
        ds.attrs["_validation"] = dict(total)
    if _defer_warnings:
        ds.attrs["_thr_stats"] = thr_stats
    logger.info(f"Preprocessing completed successfully - {n_true_total} extreme events identified")
    return ds


def compute_normalised_anomaly(
    da,
    method_anomaly: str = "shifting_baseline",
    dimensions: Optional[Dict[str, str]] = None,
    coordinates: Optional[Dict[str, str]] = None,
    window_year_baseline: int = 15,
    smooth_days_baseline: int = 21,
    std_normalise: bool = False,
    detrend_orders: Optional[List[int]] = None,
    force_zero_mean: bool = True,
    reference_period: Optional[Tuple[int, int]] = None,
    use_temp_checkpoints: bool = False,
    verbose: Optional[bool] = None,
    quiet: Optional[bool] = None,
    device: int = 0,
):
    """Anomaly stage only (detect.py:891-1116).  Returns ``Dataset{dat_anomaly, mask}`` over ALL timesteps.

    For ``shifting_baseline`` the first ``window_year_baseline`` years are NaN (no climatology), as in the
    reference where the trim happens later in ``preprocess_data``.
    """
    if detrend_orders is None:
        detrend_orders = [1]
    dimensions, coordinates = _infer_dims_coords(da, dimensions, coordinates)
    _check_reference_period_allowed(reference_period, method_anomaly)
    if method_anomaly not in _ANOMALY_METHODS:
        raise ConfigurationError(
            f"Unknown anomaly method '{method_anomaly}'",
            details="Invalid method_anomaly parameter",
            context={"provided_method": method_anomaly, "valid_methods": _ANOMALY_METHODS},
        )
    field = _Field(da, dimensions, coordinates)
    eng = get_engine(device)
    a = _anomaly_core(eng, field, method_anomaly, window_year_baseline, smooth_days_baseline, detrend_orders,
                      force_zero_mean, reference_period, None)
    eng.sync()
    cal = a["cal"]
    anom = a["anom"].cpu().numpy()
    if method_anomaly == "shifting_baseline":
        full = np.full(field.shape, np.nan, dtype=np.float32)
        full[cal.kept] = anom
        anom = full
    ds = Dataset()
    ds["dat_anomaly"] = field.labelled(anom, (field.tdim, field.time))
    ds["mask"] = field.labelled(a["mask"].cpu().numpy().astype(bool), None)
    if std_normalise and method_anomaly == "detrend_harmonic":  # detect.py:2257-2293
        sn = eng.std_normalise(a["anom"], a["dcal"])
        std_t = eng.transpose(sn["STD"], name="std_cells_major")
        eng.sync()
        ds["dat_stn"] = field.labelled(sn["dat_stn"].cpu().numpy(), (field.tdim, field.time))
        ds["STD"] = field.labelled(std_t.cpu().numpy(), None, ("dayofyear", np.arange(1, calendar.N_DOY + 1)))
    return ds


def identify_extremes(
    da,
    method_extreme: str = "hobday_extreme",
    threshold_percentile: float = 95,
    dimensions: Optional[Dict[str, str]] = None,
    coordinates: Optional[Dict[str, str]] = None,
    window_days_hobday: int = 11,
    window_spatial_hobday: Optional[int] = None,
    method_percentile: str = "approximate",
    precision: float = 0.01,
    max_anomaly: float = 5.0,
    use_temp_checkpoints: bool = False,
    verbose: Optional[bool] = None,
    quiet: Optional[bool] = None,
    device: int = 0,
):
    """Threshold + mask stage on given anomalies (detect.py:1119-1503).  Returns ``(extremes, thresholds)``."""
    import torch

    dimensions, coordinates = _infer_dims_coords(da, dimensions, coordinates)
    gridded = "y" in dimensions and dimensions["y"] in da.dims
    ws_eff = _validate_extreme_options(
        gridded, method_extreme, threshold_percentile, window_days_hobday, window_spatial_hobday,
        method_percentile, precision, max_anomaly,
    )
    field = _Field(da, dimensions, coordinates)
    eng = get_engine(device)
    bt = binning.hobday_bins(precision, max_anomaly) if method_percentile == "approximate" else None
    cal = calendar.build_calendar(field.time)
    dcal = eng.upload_calendar(cal)
    anom = field.device_x(eng)
    a = {"anom": anom, "cal": cal, "dcal": dcal, "bins": None}
    ext, thr, kind, _ = _extremes_core(
        eng, a, field, method_extreme, threshold_percentile, window_days_hobday, ws_eff, method_percentile, bt, max_anomaly
    )
    eng.sync()
    extremes = field.labelled(ext.cpu().numpy().astype(bool), (field.tdim, field.time))
    doy_axis = ("dayofyear", np.arange(1, calendar.N_DOY + 1))
    if kind == "doy_last":
        thresholds = field.labelled(thr.cpu().numpy(), None, doy_axis)
    elif kind == "doy_first":
        thresholds = field.labelled(thr.cpu().numpy(), doy_axis)
    else:
        thresholds = field.labelled(thr.cpu().numpy(), None)
    return extremes, thresholds


def _climatology(da, window_year_baseline, smooth_days, dimensions, coordinates, device):
    dimensions, coordinates = _infer_dims_coords(da, dimensions, coordinates)
    field = _Field(da, dimensions, coordinates)
    eng = get_engine(device)
    import torch

    cal_trim = calendar.build_calendar(field.time, window_year_baseline=int(window_year_baseline))
    dcal = eng.upload_calendar(cal_trim)
    x = field.device_x(eng)
    r = eng.shifting_baseline(x, dcal, int(window_year_baseline), int(smooth_days), None, write_clim=True)
    eng.sync()
    full = np.full(field.shape, np.nan, dtype=np.float32)
    full[cal_trim.kept] = r["out"].cpu().numpy()
    return field.labelled(full, (field.tdim, field.time))


def rolling_climatology(da, window_year_baseline: int = 15, dimensions=None, coordinates=None,
                        use_temp_checkpoints: bool = False, device: int = 0):
    """Day-of-year climatology of the previous ``window_year_baseline`` years for every timestep (detect.py:1511-1688)."""
    return _climatology(da, window_year_baseline, 1, dimensions, coordinates, device)


def smoothed_rolling_climatology(da, window_year_baseline: int = 15, smooth_days_baseline: int = 21, dimensions=None,
                                 coordinates=None, use_temp_checkpoints: bool = False, device: int = 0):
    """``rolling_climatology`` of the centred ``smooth_days_baseline`` rolling mean (detect.py:1691-1816)."""
    return _climatology(da, window_year_baseline, smooth_days_baseline, dimensions, coordinates, device)
