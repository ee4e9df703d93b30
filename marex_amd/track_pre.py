"""First half of the tracker's pre-processing stage on the device: the step that follows
``preprocess_data`` in the reference pipeline and consumes ``extreme_events`` / ``mask`` while they are in HBM.

Mirrors ``marEx.tracker.fill_holes`` (track.py:1520-1676, gridded branch) and ``marEx.tracker.fill_time_gaps``
(track.py:1678-1726), ``marEx.tracker.identify_objects(time_connectivity=False)`` (track.py:1912-2049) and
``marEx.tracker.filter_small_objects`` (track.py:1755-1911), for gridded data and -- given ``neighbours`` -- for
unstructured meshes.
"""
from __future__ import annotations

import numpy as np

from .exceptions import ConfigurationError


def _as_u8(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a.values if hasattr(a, "values") else a)).astype(np.uint8)


def _check(data_bin, mask, R_fill, T_fill, neighbours=None):
    d = np.asarray(data_bin.values if hasattr(data_bin, "values") else data_bin)
    m = np.asarray(mask.values if hasattr(mask, "values") else mask)
    if neighbours is not None:
        nb = np.asarray(neighbours.values if hasattr(neighbours, "values") else neighbours)
        if d.ndim != 2 or m.shape != d.shape[1:] or nb.shape != (3, d.shape[1]):  # track.py:1063-1090
            raise ConfigurationError("Invalid neighbour array shape for unstructured grid",
                                     details=f"data {d.shape}, mask {m.shape}, neighbours {nb.shape}; expected (time, ncells), (ncells), (3, ncells)")
    elif d.ndim != 3 or m.shape != d.shape[1:]:
        raise ConfigurationError("fill_holes / fill_time_gaps on the device need gridded data (time, y, x) and a (y, x) mask",
                                 details=f"data {d.shape}, mask {m.shape}")
    if T_fill % 2 != 0:  # track.py:704-709
        raise ConfigurationError("T_fill must be even for temporal symmetry", details=f"Provided T_fill={T_fill} is odd")
    if int(R_fill) < 0 or int(R_fill) > 63:
        raise ConfigurationError("R_fill must be between 0 and 63 on the device path", details=f"R_fill={R_fill}")
    return d, m


def _nbr0(neighbours) -> np.ndarray:
    """The reference's ``neighbours.astype(np.int32) - 1`` (track.py:1060): 1-based input, 0 = no neighbour."""
    nb = np.asarray(neighbours.values if hasattr(neighbours, "values") else neighbours)
    return np.ascontiguousarray(nb.astype(np.int32) - 1)


def _wrap(data_bin, res):
    return data_bin.copy(data=res) if hasattr(data_bin, "copy") and hasattr(data_bin, "dims") else res


def fill_holes(data_bin, mask, R_fill: int, regional_mode: bool = False, neighbours=None, device: int = 0):
    """Fill holes and remove specks: binary closing then opening with a disk of radius ``R_fill`` (track.py:1520-1676);
    with ``neighbours`` (1-based ``[3, ncells]``) the unstructured-mesh form on ``(time, ncells)`` data.
    Returns a bool array (or DataArray with the input's labels) of the input's shape."""
    import torch

    from .detect import get_engine

    d, m = _check(data_bin, mask, R_fill, 0, neighbours)
    eng = get_engine(device)
    if neighbours is not None:
        x = torch.from_numpy(_as_u8(d)).to(eng.device)
        out = eng.fill_holes_mesh(x, torch.from_numpy(_as_u8(m)).to(eng.device), torch.from_numpy(_nbr0(neighbours)).to(eng.device),
                                  int(R_fill))
        eng.sync()
        return _wrap(data_bin, out.cpu().numpy().astype(bool))
    T, ny, nx = d.shape
    x = torch.from_numpy(_as_u8(d).reshape(T, ny * nx)).to(eng.device)
    mk = torch.from_numpy(_as_u8(m).reshape(-1)).to(eng.device)
    out = eng.fill_holes(x, mk, ny, nx, int(R_fill), regional_mode)
    eng.sync()
    res = out.cpu().numpy().astype(bool).reshape(T, ny, nx)
    return data_bin.copy(data=res) if hasattr(data_bin, "copy") and hasattr(data_bin, "dims") else res


def fill_time_gaps(data_bin, mask, R_fill: int, T_fill: int = 2, regional_mode: bool = False, neighbours=None, device: int = 0):
    """Close gaps of up to ``T_fill`` steps in time, then ``fill_holes(R_fill // 2)`` (track.py:1678-1726)."""
    import torch

    from .detect import get_engine

    d, m = _check(data_bin, mask, R_fill, T_fill, neighbours)
    eng = get_engine(device)
    if neighbours is not None:
        if T_fill == 0:
            return data_bin
        x = torch.from_numpy(_as_u8(d)).to(eng.device)
        tmp = torch.empty_like(x)
        eng._bind_stream()
        eng.ctx.check(eng.lib.marex_time_closing_u8(eng.ctx.handle, x.data_ptr(), x.shape[0], x.shape[1], int(T_fill), tmp.data_ptr()),
                      "marex_time_closing_u8")
        out = eng.fill_holes_mesh(tmp, torch.from_numpy(_as_u8(m)).to(eng.device), torch.from_numpy(_nbr0(neighbours)).to(eng.device),
                                  int(R_fill) // 2)
        eng.sync()
        return _wrap(data_bin, out.cpu().numpy().astype(bool))
    T, ny, nx = d.shape
    x = torch.from_numpy(_as_u8(d).reshape(T, ny * nx)).to(eng.device)
    mk = torch.from_numpy(_as_u8(m).reshape(-1)).to(eng.device)
    out = eng.fill_time_gaps(x, mk, ny, nx, int(R_fill), int(T_fill), regional_mode)
    eng.sync()
    res = out.cpu().numpy().astype(bool).reshape(T, ny, nx)
    return data_bin.copy(data=res) if hasattr(data_bin, "copy") and hasattr(data_bin, "dims") else res


def identify_objects_2d(data_bin, regional_mode: bool = False, device: int = 0):
    """Per-timestep 8-connected components, periodic in x unless ``regional_mode`` (track.py:2013-2031 with
    ``time_connectivity=False``).  Returns ``(ID field int32 [T, ny, nx], number of objects)``; IDs are unique across
    time, 0 = background; their numbering (1 + smallest linear index of the object) differs from the reference's."""
    import torch

    from .detect import get_engine

    d = np.asarray(data_bin.values if hasattr(data_bin, "values") else data_bin)
    if d.ndim != 3:
        raise ConfigurationError("identify_objects_2d on the device needs gridded data (time, y, x)", details=f"data {d.shape}")
    eng = get_engine(device)
    T, ny, nx = d.shape
    x = torch.from_numpy(_as_u8(d).reshape(T, ny * nx)).to(eng.device)
    r = eng.label_objects_2d(x, ny, nx, wrap_x=not regional_mode)
    eng.sync()
    n = int((r["areas"] > 0).sum().item())
    return r["labels"].cpu().numpy().reshape(T, ny, nx), n


def filter_small_objects(data_bin, area_filter_quartile: float = 0.5, area_filter_absolute=None, regional_mode: bool = False,
                         mask=None, neighbours=None, device: int = 0):
    """Remove objects smaller than the ``area_filter_quartile`` percentile of all object areas (or an absolute number of
    cells).  Returns ``(filtered, area_threshold, object_areas, N_objects_prefiltered, N_objects_filtered)`` like
    track.py:1755-1911 (gridded branch: areas in cells)."""
    import torch

    from .detect import get_engine

    d = np.asarray(data_bin.values if hasattr(data_bin, "values") else data_bin)
    if neighbours is not None:  # unstructured mesh: sizes in cells, clusters > 50 (5) cells enter the percentile, keep "> threshold"
        d, m = _check(data_bin, mask, 0, 0, neighbours)
        eng = get_engine(device)
        r = eng.filter_small_objects_mesh(torch.from_numpy(_as_u8(d)).to(eng.device), torch.from_numpy(_as_u8(m)).to(eng.device),
                                          torch.from_numpy(_nbr0(neighbours)).to(eng.device), area_filter_quartile, area_filter_absolute)
        eng.sync()
        return (_wrap(data_bin, r["filtered"].cpu().numpy().astype(bool)), r["area_threshold"], r["object_areas"].cpu().numpy(),
                r["n_before"], r["n_after"])
    if d.ndim != 3:
        raise ConfigurationError("filter_small_objects on the device needs gridded data (time, y, x)", details=f"data {d.shape}")
    eng = get_engine(device)
    T, ny, nx = d.shape
    x = torch.from_numpy(_as_u8(d).reshape(T, ny * nx)).to(eng.device)
    r = eng.filter_small_objects(x, ny, nx, area_filter_quartile, area_filter_absolute, regional_mode)
    eng.sync()
    res = r["filtered"].cpu().numpy().astype(bool).reshape(T, ny, nx)
    out = data_bin.copy(data=res) if hasattr(data_bin, "copy") and hasattr(data_bin, "dims") else res
    return out, r["area_threshold"], r["object_areas"].cpu().numpy(), r["n_before"], r["n_after"]


def run_preprocess(extreme_events, mask, R_fill: int, T_fill: int = 2, area_filter_quartile: float = 0.5,
                   area_filter_absolute=None, regional_mode: bool = False, device: int = 0):
    """The whole pre-processing stage of ``marEx.tracker.run_preprocess`` (track.py:1283-1360) for gridded data, on the
    device without intermediate host copies: ``fill_holes`` -> ``fill_time_gaps`` -> ``filter_small_objects``.

    Returns ``(data_bin_filtered, object_stats)`` with ``object_stats = (total_area_IDed, N_objects_prefiltered,
    N_objects_filtered, area_threshold, accepted_area_fraction, preprocessed_area_fraction)`` as in the reference
    (areas in cells; ``accepted_area`` sums the objects STRICTLY above the threshold, as track.py:1337 does)."""
    import torch

    from .detect import get_engine

    d, m = _check(extreme_events, mask, R_fill, T_fill)
    eng = get_engine(device)
    T, ny, nx = d.shape
    x = torch.from_numpy(_as_u8(d).reshape(T, ny * nx)).to(eng.device)
    mk = torch.from_numpy(_as_u8(m).reshape(-1)).to(eng.device)
    raw_area = float(x.sum().item())
    a = eng.fill_holes(x, mk, ny, nx, int(R_fill), regional_mode)
    g = eng.fill_time_gaps(a, mk, ny, nx, int(R_fill), int(T_fill), regional_mode)
    r = eng.filter_small_objects(g, ny, nx, area_filter_quartile, area_filter_absolute, regional_mode)
    eng.sync()
    areas = r["object_areas"].to(torch.float64)
    total = float(areas.sum().item())
    accepted = float(areas[areas > r["area_threshold"]].sum().item())
    processed = float(r["filtered"].sum().item())
    res = r["filtered"].cpu().numpy().astype(bool).reshape(T, ny, nx)
    stats = (total, r["n_before"], r["n_after"], r["area_threshold"], accepted / total, raw_area / processed if processed else float("nan"))
    return _wrap(extreme_events, res), stats
