"""Histogram bin tables of the approximate percentile method (host side).

The bit pattern of the float32 edge table decides which bin an anomaly falls
into, so it is produced with exactly the NumPy expression the reference uses
and passed to the device as a table -- the kernels never re-derive edges.

Reference: marEx/detect.py:2601-2608 (2-D / Hobday path, float32 edges) and
2770-2784 (1-D / global path, float64 edges).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class BinTable:
    edges: np.ndarray  # [nb+1] first entry -inf
    centres: np.ndarray  # [nb] centres[0] == 0
    precision: float
    max_anomaly: float

    @property
    def nb(self) -> int:
        return int(self.centres.shape[0])

    @property
    def lower_bound(self):
        """Clamp value: thresholds below ``edges[3]`` are raised to it (detect.py:2709, 2732)."""
        return self.edges[3]

    @property
    def upper_bound(self):
        """Warning bound ``edges[-2]`` (detect.py:2708)."""
        return self.edges[-2]


def hobday_bins(precision: float = 0.01, max_anomaly: float = 5.0) -> BinTable:
    """float32 edges / centres of ``_compute_histogram_quantile_2d`` (detect.py:2603-2608)."""
    edges = np.concatenate(
        [[-np.inf], np.arange(-precision, max_anomaly + precision, precision, dtype=np.float32)], dtype=np.float32
    )
    centres = (edges[1:] + edges[:-1]) / 2
    centres[0] = 0.0
    return BinTable(edges=edges, centres=centres.astype(np.float32), precision=precision, max_anomaly=max_anomaly)


def global_bins(precision: float = 0.01, max_anomaly: float = 5.0) -> BinTable:
    """float64 edges / centres of ``_compute_histogram_quantile_1d`` (detect.py:2772, 2783-2784)."""
    edges = np.concatenate([[-np.inf], np.arange(-precision, max_anomaly + precision, precision)])
    centres = (edges[1:] + edges[:-1]) / 2
    centres[0] = 0.0
    return BinTable(edges=edges, centres=centres, precision=precision, max_anomaly=max_anomaly)
