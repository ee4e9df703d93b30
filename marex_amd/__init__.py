"""marex_amd -- MI355X-native (gfx950) hot path behind marEx-style ``preprocess_data``.

Public names mirror the reference's re-exports for this path (marEx/__init__.py:36-42, 45-58).
Importing the package never touches the GPU; the HIP extension is loaded on the first compute call and
its absence is an error (no CPU fallback).
"""
from .detect import (
    compute_normalised_anomaly,
    identify_extremes,
    preprocess_data,
    rolling_climatology,
    smoothed_rolling_climatology,
)
from .exceptions import (
    ConfigurationError,
    DataValidationError,
    DependencyError,
    MarExError,
    ProcessingError,
    create_data_validation_error,
)
from .dask_adapter import preprocess_data_lazy
from .xr_compat import DataArray, Dataset

__all__ = [
    "preprocess_data", "preprocess_data_lazy", "compute_normalised_anomaly", "identify_extremes", "rolling_climatology",
    "smoothed_rolling_climatology", "MarExError", "DataValidationError", "ConfigurationError",
    "ProcessingError", "DependencyError", "create_data_validation_error", "DataArray", "Dataset",
]
__version__ = "0.1.0"
