"""Error convention of the drop-in API (no GPU needed: every check fires before the device is touched).

Same exception classes and message patterns as the reference's tests/test_error_handling.py.
"""
import numpy as np
import pytest

import marex_amd
from marex_amd import ConfigurationError, DataValidationError, calendar
from marex_amd.detect import _infer_dims_coords
from marex_amd.xr_compat import DataArray


@pytest.fixture(scope="module")
def da():
    tm = calendar.daily_time_axis("2000-01-01", 3 * 365)
    x = np.random.default_rng(0).normal(15, 1, (tm.size, 4, 5)).astype(np.float32)
    return DataArray(x, dims=("time", "lat", "lon"), coords={"time": tm, "lat": np.arange(4.0), "lon": np.arange(5.0)})


@pytest.fixture(scope="module")
def da_unstructured():
    tm = calendar.daily_time_axis("2000-01-01", 3 * 365)
    x = np.random.default_rng(0).normal(15, 1, (tm.size, 12)).astype(np.float32)
    return DataArray(x, dims=("time", "ncells"), coords={"time": tm, "lon": ("ncells", np.arange(12.0)), "lat": ("ncells", np.arange(12.0))})


def test_missing_dimensions_and_coordinates(da):
    with pytest.raises(DataValidationError, match=r"Missing required dimensions"):
        marex_amd.preprocess_data(da, dimensions={"time": "time", "x": "longitude", "y": "lat"})
    with pytest.raises(DataValidationError, match=r"Missing required coordinates"):
        marex_amd.preprocess_data(da, coordinates={"time": "time", "x": "nope", "y": "lat"})
    with pytest.raises(DataValidationError, match=r"Missing required dimensions"):
        marex_amd.compute_normalised_anomaly(da, dimensions={"time": "t", "x": "lon", "y": "lat"})


def test_unstructured_needs_coordinates(da_unstructured):
    with pytest.raises(DataValidationError, match=r"Coordinates parameter must be explicitly specified for unstructured data"):
        marex_amd.preprocess_data(da_unstructured, dimensions={"time": "time", "x": "ncells"})
    dims, coords = _infer_dims_coords(da_unstructured, {"x": "ncells"}, {"x": "lon", "y": "lat"})
    assert dims["time"] == "time" and coords["time"] == "time"  # partial defaults (detect.py:164-165, 193-194)


def test_unknown_methods(da):
    with pytest.raises(ConfigurationError, match=r"Unknown anomaly method"):
        marex_amd.preprocess_data(da, method_anomaly="invalid_method")
    with pytest.raises(ConfigurationError, match=r"Unknown extreme method 'invalid_extreme'"):
        marex_amd.preprocess_data(da, method_extreme="invalid_extreme")
    with pytest.raises(ConfigurationError, match="Unknown method_percentile 'invalid_method'"):
        marex_amd.preprocess_data(da, method_percentile="invalid_method")


def test_option_compatibility(da, da_unstructured):
    with pytest.raises(ConfigurationError, match="Percentile threshold 50% is not supported with method_percentile='approximate'"):
        marex_amd.preprocess_data(da, threshold_percentile=50)
    with pytest.raises(ConfigurationError, match="Parameter 'precision' cannot be used with method_percentile='exact'"):
        marex_amd.preprocess_data(da, method_percentile="exact", precision=0.05)
    with pytest.raises(ConfigurationError, match="Parameter 'max_anomaly' cannot be used with method_percentile='exact'"):
        marex_amd.preprocess_data(da, method_percentile="exact", max_anomaly=10.0)
    with pytest.raises(ConfigurationError, match="window_days_hobday must be an odd number"):
        marex_amd.preprocess_data(da, window_days_hobday=10)
    with pytest.raises(ConfigurationError, match="window_spatial_hobday must be an odd number"):
        marex_amd.preprocess_data(da, window_spatial_hobday=4)
    with pytest.raises(ConfigurationError, match="window_spatial_hobday can only be used with method_extreme='hobday_extreme'"):
        marex_amd.preprocess_data(da, method_extreme="global_extreme", window_spatial_hobday=5)
    with pytest.raises(ConfigurationError, match="window_spatial_hobday is not supported with method_percentile='exact'"):
        marex_amd.preprocess_data(da, method_percentile="exact", window_spatial_hobday=5)
    with pytest.raises(ConfigurationError, match="window_spatial_hobday is not supported for unstructured grids"):
        marex_amd.preprocess_data(
            da_unstructured, dimensions={"time": "time", "x": "ncells"},
            coordinates={"time": "time", "x": "lon", "y": "lat"}, window_spatial_hobday=5,
        )
    with pytest.raises(ConfigurationError, match="reference_period is not supported"):
        marex_amd.preprocess_data(da, method_anomaly="shifting_baseline", reference_period=(2000, 2001))
    with pytest.raises(ConfigurationError, match="reference_period is not supported"):
        marex_amd.compute_normalised_anomaly(da, method_anomaly="detrend_harmonic", reference_period=(2000, 2001))


def test_error_objects_carry_structure(da):
    with pytest.raises(ConfigurationError) as ei:
        marex_amd.preprocess_data(da, window_days_hobday=4)
    err = ei.value
    assert isinstance(err, marex_amd.MarExError) and err.error_code == "CONFIGURATION"
    assert err.context["window_days_hobday"] == 4 and err.suggestions
    assert "Suggestions:" in str(err) and "Error Code: CONFIGURATION" in str(err)
    e2 = marex_amd.create_data_validation_error("msg", data_info={"a": 1}, details="d")
    assert e2.error_code == "DATA_VALIDATION" and e2.context == {"a": 1} and "Details: d" in str(e2)


def test_compute_without_gpu_fails_loudly(da):
    """No CPU fallback: on a box without a HIP device the compute entry raises instead of computing."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(marex_amd.ProcessingError, match="needs a HIP device"):
        marex_amd.preprocess_data(da, window_year_baseline=2)


def test_kernel_family_note_names_the_slower_paths_once(caplog):
    """Options whose defaults encode performance cliffs: a call that leaves the tuned anomaly kernel says so at
    INFO, once per distinct configuration."""
    import logging

    from marex_amd import engine

    assert engine.shifting_kernel_family(15, 21, 1440, True) == "lean" and engine.shifting_kernel_family(5, 21, 8, False) == "lean"
    assert engine.shifting_kernel_family(10, 21, 1440, True) == "fast" and engine.shifting_kernel_family(15, 21, 1441, True) == "fast"
    assert engine.shifting_kernel_family(15, 15, 1440, True) == "fast" and engine.shifting_kernel_family(15, 31, 1440, True) == "general"
    assert engine.shifting_kernel_family(9, 21, 1440, True) == "general"
    engine._noted.discard("probe message")
    with caplog.at_level(logging.INFO, logger="marex_amd"):
        engine._note_path("probe message")
        engine._note_path("probe message")
    assert [r.getMessage() for r in caplog.records].count("probe message") == 1


def test_require_dask_is_the_references_error(da):
    """detect.py:558-568 on request: the eager call accepts in-memory arrays (documented deviation), `require_dask=True` refuses
    them with the reference's message -- after the dims / coords inference, before anything touches the device."""
    with pytest.raises(DataValidationError, match=r"Input DataArray must be Dask-backed") as ei:
        marex_amd.preprocess_data(da, require_dask=True)
    assert "chunk({'time': 30})" in str(ei.value) or any("chunk" in s for s in getattr(ei.value, "suggestions", []))
    with pytest.raises(DataValidationError, match=r"Missing required dimensions"):  # the order of the reference's checks
        marex_amd.preprocess_data(da, dimensions={"time": "time", "x": "longitude", "y": "lat"}, require_dask=True)
