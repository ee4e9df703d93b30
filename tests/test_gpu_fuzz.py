"""GPU: seeded random configurations of shifting_baseline + hobday_extreme against the oracle, bit for bit.

Each case draws a start date, a series length, W, S, wd, ws, the percentile, a grid shape (gridded or unstructured) and
optional damage to the field (NaN gaps in ocean cells, cells that start as NaN, constant cells, spikes past max_anomaly),
so that the kernel selection logic (fast / general anomaly kernel per chunk, tile shapes, speculative / exact threshold
paths, short / long buckets) is exercised in combinations no hand-written case lists.
"""
import os

import numpy as np
import pytest

from tests.test_gpu_shifting_hobday import check_all, run_case

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    year0 = int(rng.integers(1950, 2015))
    start = f"{year0}-{int(rng.integers(1, 13)):02d}-{int(rng.integers(1, 29)):02d}" if rng.random() < 0.5 else f"{year0}-01-01"
    W = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 16, 18]))
    n_years = W + int(rng.integers(3, 12))
    periods = n_years * 365 + int(rng.integers(0, 300))
    S = int(rng.choice([21, 21, 21, 1, 7, 30, 31]))
    wd = int(rng.choice([3, 5, 11, 11, 21, 31]))
    unstructured = rng.random() < 0.25
    if unstructured:
        ny, nx, ws = 1, int(rng.integers(3, 700)), None
    else:
        ny, nx = int(rng.integers(3, 40)), int(rng.integers(4, 50))
        ws = int(rng.choice([1, 3, 5, 5, 7]))
    pct = float(rng.choice([60.0, 90.0, 95.0, 95.0, 99.0, 100.0]))
    damage = rng.random() < 0.5

    def mutate(x):
        ocean = np.flatnonzero(np.isfinite(x[0]))
        if ocean.size < 6:
            return
        pick = rng.choice(ocean, size=5, replace=False)
        T = x.shape[0]
        a = int(rng.integers(0, T - 30))
        x[a:a + int(rng.integers(1, 30)), pick[0]] = np.nan
        x[: int(rng.integers(1, 60)), pick[1]] = np.nan
        x[:, pick[2]] = np.float32(rng.normal(10, 3))
        x[:, pick[3]] += np.float32(30.0) * (np.arange(T) % int(rng.integers(5, 40)) == 0)
        x[int(rng.integers(0, T)), pick[4]] = np.inf if rng.random() < 0.5 else -np.inf

    return dict(start=start, periods=periods, ny=ny, nx=nx, W=W, S=S, wd=wd, ws=ws, pct=pct, unstructured=unstructured,
                seed=20240607 + seed, mutate=mutate if damage else None)


# MAREX_FUZZ_SEEDS="a:b" widens the sweep (one-off hunts; the default 36 cases keep the suite at a few minutes)
_lo, _hi = (int(v) for v in os.environ.get("MAREX_FUZZ_SEEDS", "0:36").split(":"))


@pytest.mark.parametrize("seed", list(range(_lo, _hi)))
def test_random_configuration_matches_the_oracle(hot, seed):
    c = _case(seed)
    # seed % 3: 0 = the representation the engine picks, 1 = tails forced (short tails re-read buckets), 2 = bin matrix
    # (odd seeds there: mask from the bin matrix whatever the bucket length)
    path = (None, "tails", "bins")[seed % 3]
    r = run_case(hot, c["start"], c["periods"], c["ny"], c["nx"], c["W"], c["S"], c["wd"], c["ws"], pct=c["pct"],
                 unstructured=c["unstructured"], seed=c["seed"], mutate=c["mutate"], path=path,
                 opts={"MASK_BINS": 1} if seed % 2 else {})
    check_all(*r)
