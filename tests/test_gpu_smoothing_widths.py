"""GPU: the fast anomaly kernel instantiated for other smoothing widths than the default 21 days
(``smooth_days_baseline`` = 11, 15 with ``window_year_baseline`` = 5, 10, 15): same bits as the oracle -- and as the general
kernel, which keeps every other (W, S) -- on both histogram representations, with leap days, a mid-year start, the NaN rim of
the smoothing at both ends of the series, gaps and late-starting cells."""
import numpy as np
import pytest

from tests.test_gpu_shifting_hobday import PATHS, check_all, run_case

pytestmark = pytest.mark.gpu


def _mutate(x):
    x[100:140, 7] = np.nan           # a gap inside a window
    x[:900, 9] = np.nan              # a cell that starts late
    x[1234, 11] = np.inf
    x[:, 12] = np.nan                # land


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("W,S,years,start", [(5, 11, 12, "2003-01-01"), (15, 15, 22, "1990-01-01"), (10, 11, 17, "1999-07-19"),
                                             (15, 11, 19, "2001-03-02"), (5, 15, 9, "2004-02-29"), (10, 15, 30, "1981-01-01")])
def test_fast_kernel_for_other_smoothing_widths(hot, W, S, years, start, path):
    case = run_case(hot, start, years * 365 + years // 4, 7, 23, W, S, 11, 5, mutate=_mutate, path=path)
    check_all(*case)
    # the general kernel on the same input: identical anomalies (it is what ran these widths before)
    ref = run_case(hot, start, years * 365 + years // 4, 7, 23, W, S, 11, 5, mutate=_mutate, path=path, opts={"SHIFT_FAST": 0})
    assert np.array_equal(case[4]["dat_anomaly"].cpu().numpy(), ref[4]["dat_anomaly"].cpu().numpy(), equal_nan=True)
    assert np.array_equal(case[4]["extreme_events"].cpu().numpy(), ref[4]["extreme_events"].cpu().numpy())


def test_unstructured_short_series_other_width(hot):
    check_all(*run_case(hot, "2010-01-01", 8 * 365 + 2, 1, 300, 5, 11, 5, None, unstructured=True))
