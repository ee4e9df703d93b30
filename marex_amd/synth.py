"""Synthetic gridded SST fields for tests and ``bench.py``.

The field is defined so that host (NumPy) and device (HIP kernel ``marex_synth_sst_f32``)
produce IDENTICAL bits: all transcendental terms are evaluated once on the host into small
per-cell / per-timestep float32 tables, the noise comes from a counter-based integer hash of
``(seed, t, c)`` (splitmix64 finaliser) turned into an Irwin-Hall(8) variate with exact integer
arithmetic, and the remaining float32 operations are performed in one fixed order.

    x[t,c] = ((mean[c] + amp[c]*seas[t,hemi[c]]) + trend[t]) + noise_amp*z[t,c]     (NaN on land)

Units are deg C so that 1e-5-relative tolerances are meaningful.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .calendar import _to_year_doy

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_K_SEED = np.uint64(0x9E3779B97F4A7C15)
_K_T = np.uint64(0xD1B54A32D192ED03)
_K_C = np.uint64(0x8CB92BA72F3D8DD7)
_K_2 = np.uint64(0xA5A5A5A5A5A5A5A5)

#: 1 / (2 * std of a sum of eight uniform 16-bit integers); makes z ~ unit variance
Z_SCALE = np.float32(1.0 / (2.0 * np.sqrt(8.0 * (65536.0**2 - 1.0) / 12.0)))
NOISE_AMP = np.float32(0.8)


def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _sum16(h: np.ndarray) -> np.ndarray:
    m = np.uint64(0xFFFF)
    return (h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + (h >> np.uint64(48))


def noise_z(seed: int, t: np.ndarray, c: np.ndarray) -> np.ndarray:
    """Irwin-Hall(8) noise, float32, broadcast over ``t`` and ``c`` (uint64 counters)."""
    with np.errstate(over="ignore"):
        key = np.uint64(seed) * _K_SEED + t.astype(np.uint64) * _K_T + c.astype(np.uint64) * _K_C
        h1 = _mix64(key)
        h2 = _mix64(key ^ _K_2)
        s = _sum16(h1) + _sum16(h2)
    centred = (2 * s.astype(np.int64) - 8 * 65535).astype(np.float32)  # |.| < 2**20: exact
    return centred * Z_SCALE


@dataclass
class SynthTables:
    mean: np.ndarray  # [C] f32
    amp: np.ndarray  # [C] f32
    hemi: np.ndarray  # [C] u8 (0 = north/phase 30, 1 = south/phase 212)
    land: np.ndarray  # [C] u8
    seas: np.ndarray  # [T, 2] f32
    trend: np.ndarray  # [T] f32
    seed: int
    ny: int
    nx: int

    @property
    def C(self) -> int:
        return int(self.mean.shape[0])

    @property
    def T(self) -> int:
        return int(self.trend.shape[0])


def make_tables(time, ny: int, nx: int, seed: int = 20240607, *, unstructured: bool = False, lat_range=None) -> SynthTables:
    """Tables for a ``ny x nx`` lat-lon grid (or ``C = nx`` unstructured cells when ``ny == 0`` / ``unstructured``).

    ``lat_range=(j0, j1, ny_global)`` builds the tables of latitude rows ``j0..j1-1`` of a larger global grid
    so that spatial shards see exactly the cells of the global field (cell ids stay global).
    """
    _, doy = _to_year_doy(time)
    T = doy.size
    tt = np.arange(T, dtype=np.float64)
    ph = np.stack(
        [np.sin(2 * np.pi * (doy - 30.0) / 365.25), np.sin(2 * np.pi * (doy - 212.0) / 365.25)], axis=1
    ).astype(np.float32)
    trend = (0.02 * tt / 365.25).astype(np.float32)

    if unstructured or ny == 0:
        C = nx
        cc = np.arange(C, dtype=np.float64)
        lat = np.arcsin(2 * (cc + 0.5) / C - 1.0)
        lon = (cc * 2.399963229728653) % (2 * np.pi)  # golden-angle spiral
        ny_eff = 0
    else:
        if lat_range is None:
            j0, j1, nyg = 0, ny, ny
        else:
            j0, j1, nyg = lat_range
            assert j1 - j0 == ny
        jj = np.arange(j0, j1, dtype=np.float64)
        latv = (-90.0 + (jj + 0.5) * 180.0 / nyg) * np.pi / 180.0
        lonv = (np.arange(nx, dtype=np.float64) + 0.5) * 2 * np.pi / nx
        lat = np.repeat(latv, nx)
        lon = np.tile(lonv, ny)
        ny_eff = ny
    mean = (15.0 + 12.0 * np.cos(lat)).astype(np.float32)
    amp = (3.0 + 5.0 * np.abs(np.sin(lat))).astype(np.float32)
    hemi = (lat < 0).astype(np.uint8)
    # smooth "continents" (~30 % of the cells) plus an all-land first global row
    land = (np.sin(3 * lon) * np.cos(2 * lat) + 0.3 * np.sin(7 * lon + 1.0) + 0.2 * np.cos(5 * lat) > 0.38).astype(np.uint8)
    if ny_eff and (lat_range is None or lat_range[0] == 0):
        land[:nx] = 1
    return SynthTables(mean, amp, hemi, land, ph, trend, int(seed), ny_eff, nx)


def cell_id_base(tab: SynthTables, lat_range=None) -> int:
    """Global id of the first cell of a latitude shard (noise counters use global ids)."""
    return 0 if lat_range is None else int(lat_range[0]) * tab.nx


def synth_field(tab: SynthTables, t0: int = 0, t1: int | None = None, cell_base: int = 0) -> np.ndarray:
    """Host evaluation of the field, ``[t1-t0, C]`` float32 (fixed float32 operation order)."""
    t1 = tab.T if t1 is None else t1
    t = np.arange(t0, t1, dtype=np.uint64)[:, None]
    c = (np.arange(tab.C, dtype=np.uint64) + np.uint64(cell_base))[None, :]
    z = noise_z(tab.seed, t, c)
    seas = tab.seas[t0:t1][:, tab.hemi.astype(np.int64)]  # [t, C]
    a = tab.amp[None, :] * seas
    b = tab.mean[None, :] + a
    cc = b + tab.trend[t0:t1, None]
    x = cc + NOISE_AMP * z
    x = x.astype(np.float32)
    x[:, tab.land.astype(bool)] = np.nan
    return x
