"""Host calendar tables (marex_amd/calendar.py) against pandas and the reference's known values."""
import numpy as np
import pandas as pd

from marex_amd import calendar


def test_year_doy_tables_and_trim():
    tm = calendar.daily_time_axis("1999-03-15", 9 * 365 + 100)
    cal = calendar.build_calendar(tm, window_year_baseline=3)
    idx = pd.DatetimeIndex(tm)
    assert np.array_equal(cal.year, idx.year) and np.array_equal(cal.doy, idx.dayofyear)
    assert cal.min_year == 1999 and cal.first_valid_year_idx == 3
    assert np.array_equal(cal.kept, idx.year >= 2002)  # detect.py:638-641
    # tindex is the inverse map of (year, doy)
    t = np.arange(cal.T)
    assert np.array_equal(cal.tindex[cal.year - cal.min_year, cal.doy - 1], t)
    assert (cal.tindex >= 0).sum() == cal.T
    # dayofyear-sorted rows: stable in time, bucket boundaries consistent
    assert cal.doy_start[0] == 0 and cal.doy_start[-1] == cal.T_out
    d_sorted = cal.doy_out[cal.doy_rows]
    assert np.all(np.diff(d_sorted) >= 0)
    for d in (1, 59, 60, 366):
        rows = cal.doy_rows[cal.doy_start[d - 1] : cal.doy_start[d]]
        assert np.all(cal.doy_out[rows] == d) and np.all(np.diff(rows) > 0)
    # rowb_index inverts doy_rows on the kept rows
    assert np.array_equal(cal.doy_rows[cal.rowb_index[cal.kept]], cal.out_index[cal.kept])
    assert not cal.has_duplicates


def test_leap_day_labels():
    tm = calendar.daily_time_axis("2000-01-01", 366 + 365)
    cal = calendar.build_calendar(tm)
    assert cal.doy[59] == 60 and cal.doy[365] == 366  # 29 Feb 2000 and 31 Dec 2000
    assert cal.tindex[1, 365] == -1  # 2001 has no dayofyear 366
    assert cal.T_out == cal.T


def test_duplicates_are_flagged():
    tm = np.array(["2001-01-01T00", "2001-01-01T12", "2001-01-02T00"], dtype="datetime64[h]")
    assert calendar.build_calendar(tm).has_duplicates


def test_decimal_year_known_values():
    # tests/test_detect_helpers.py:20-152 of the reference pins these properties
    dy = calendar.decimal_year(np.array(["2000-01-01", "2000-07-02", "2001-01-01", "2001-07-02"], dtype="datetime64[D]"))
    assert dy[0] == 2000.0 and dy[2] == 2001.0
    assert abs(dy[1] - (2000 + 183 / 366)) < 1e-12 and abs(dy[3] - (2001 + 182 / 365)) < 1e-12


def test_detrend_model_shapes_and_orthogonality():
    tm = calendar.daily_time_axis("1990-01-01", 4000)
    model, pmodel = calendar.detrend_model(calendar.decimal_year(tm), [1, 2], True)
    assert model.shape == (7, 4000) and pmodel.shape == (4000, 7)
    assert np.allclose(model[1:].mean(axis=1), 0, atol=1e-9)
    assert np.allclose(model @ pmodel, np.eye(7), atol=1e-8)
