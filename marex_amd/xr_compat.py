"""xarray adapter layer.

The reference boundary is ``xr.DataArray`` in / ``xr.Dataset`` out (marEx/detect.py:287-313).  xarray is
not installed in the build image nor on the GPU box, so this module provides

* ``DataArray`` / ``Dataset``: the real xarray classes when xarray is importable, otherwise small
  duck-typed stand-ins that carry ``dims / coords / attrs / values`` -- exactly the part of the
  xarray interface the hot-path adapter reads and writes;
* helper accessors used by :mod:`marex_amd.detect` that work on both.

The stand-ins are plain containers: they perform no computation of the hot path.
"""

from __future__ import annotations

from typing import Any, Dict, Iterable, Mapping, Optional, Sequence, Tuple

import numpy as np

try:  # pragma: no cover - xarray is absent in this image
    import xarray as _xr

    HAVE_XARRAY = True
except Exception:  # pragma: no cover
    _xr = None
    HAVE_XARRAY = False


class _MiniDataArray:
    """Minimal labelled array (used only when xarray is missing)."""

    def __init__(self, data, dims: Optional[Sequence[str]] = None, coords: Optional[Mapping[str, Any]] = None,
                 name: Optional[str] = None, attrs: Optional[Mapping[str, Any]] = None):
        self.data = data if hasattr(data, "shape") and hasattr(data, "dtype") else np.asarray(data)
        if dims is None:
            dims = tuple(f"dim_{i}" for i in range(self.data.ndim))
        self.dims: Tuple[str, ...] = tuple(dims)
        if len(self.dims) != self.data.ndim:
            raise ValueError(f"dims {self.dims} do not match data of rank {self.data.ndim}")
        self.name = name
        self.attrs: Dict[str, Any] = dict(attrs or {})
        self.coords: Dict[str, "_MiniDataArray"] = {}
        for k, v in (coords or {}).items():
            self.coords[k] = _as_coord(k, v, self.dims)

    # ---- array protocol
    @property
    def values(self) -> np.ndarray:
        return np.asarray(self.data)

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def ndim(self):
        return self.data.ndim

    @property
    def size(self):
        return int(np.prod(self.shape))

    @property
    def sizes(self) -> Dict[str, int]:
        return dict(zip(self.dims, self.shape))

    def __array__(self, dtype=None):
        a = self.values
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        raise TypeError("only coordinate lookup by name is supported by the stand-in DataArray")

    def __repr__(self):  # pragma: no cover
        return f"<marex_amd.DataArray {self.name!r} dims={self.sizes} dtype={self.dtype}>"

    # ---- the few transformations the adapter / tests use
    def transpose(self, *dims):
        if Ellipsis in dims:
            i = dims.index(Ellipsis)
            rest = [d for d in self.dims if d not in dims]
            dims = tuple(dims[:i]) + tuple(rest) + tuple(dims[i + 1:])
        order = [self.dims.index(d) for d in dims]
        return _MiniDataArray(np.transpose(self.values, order), dims, self.coords, self.name, self.attrs)

    def astype(self, dtype):
        return _MiniDataArray(self.values.astype(dtype), self.dims, self.coords, self.name, self.attrs)

    def isel(self, indexers: Optional[Mapping[str, Any]] = None, **kw):
        indexers = dict(indexers or {}, **kw)
        idx = tuple(indexers.get(d, slice(None)) for d in self.dims)
        data = self.values[idx]
        dims = tuple(d for d in self.dims if not np.isscalar(indexers.get(d, slice(None))))
        coords = {}
        for k, c in self.coords.items():
            if all(d in dims or d in indexers for d in c.dims):
                cidx = tuple(indexers.get(d, slice(None)) for d in c.dims)
                cd = tuple(d for d in c.dims if not np.isscalar(indexers.get(d, slice(None))))
                if all(d in dims for d in cd):
                    coords[k] = _MiniDataArray(c.values[cidx], cd, None, k)
        return _MiniDataArray(data, dims, coords, self.name, self.attrs)

    def chunk(self, *a, **k):
        return self

    def compute(self):
        return self

    def persist(self):
        return self

    def load(self):
        return self

    def sum(self, *a, **k):
        return self.values.sum(*a, **k)

    def mean(self, *a, **k):
        return self.values.mean(*a, **k)

    def any(self):
        return bool(self.values.any())

    def copy(self):
        return _MiniDataArray(self.values.copy(), self.dims, self.coords, self.name, dict(self.attrs))


def _as_coord(name, v, parent_dims) -> _MiniDataArray:
    if isinstance(v, _MiniDataArray):
        return v
    if isinstance(v, tuple) and len(v) == 2:  # (dims, data)
        d, data = v
        d = (d,) if isinstance(d, str) else tuple(d)
        return _MiniDataArray(np.asarray(data), d, None, name)
    a = np.asarray(v)
    if a.ndim == 0:
        return _MiniDataArray(a, (), None, name)
    return _MiniDataArray(a, (name,), None, name)


class _MiniDataset:
    """Minimal Dataset stand-in (dict of DataArrays + coords + attrs)."""

    def __init__(self, data_vars: Optional[Mapping[str, Any]] = None, coords: Optional[Mapping[str, Any]] = None,
                 attrs: Optional[Mapping[str, Any]] = None):
        self.data_vars: Dict[str, _MiniDataArray] = {}
        self.coords: Dict[str, _MiniDataArray] = {}
        self.attrs: Dict[str, Any] = dict(attrs or {})
        self.encoding: Dict[str, Any] = {}  # like xarray's: never written by a store
        for k, v in (coords or {}).items():
            self.coords[k] = _as_coord(k, v, ())
        for k, v in (data_vars or {}).items():
            self[k] = v

    def __setitem__(self, key, value):
        if not isinstance(value, _MiniDataArray):
            raise TypeError("Dataset values must be DataArray")
        value.name = key
        self.data_vars[key] = value
        for ck, cv in value.coords.items():
            self.coords.setdefault(ck, cv)

    def __getitem__(self, key):
        if key in self.data_vars:
            return self.data_vars[key]
        return self.coords[key]

    def __getattr__(self, key):
        dv = self.__dict__.get("data_vars", {})
        if key in dv:
            return dv[key]
        co = self.__dict__.get("coords", {})
        if key in co:
            return co[key]
        raise AttributeError(key)

    def __contains__(self, key):
        return key in self.data_vars or key in self.coords

    @property
    def dims(self) -> Dict[str, int]:
        out: Dict[str, int] = {}
        for v in self.data_vars.values():
            out.update(v.sizes)
        return out

    sizes = dims

    def to_zarr(self, store, mode: str = "w", chunks=None, **_ignored):
        """``Dataset.to_zarr`` as the reference's batch script calls it (examples/batch jobs/run_detect.py:83)."""
        from .zarr_io import write_dataset

        import os

        if mode not in ("w", "w-"):
            raise ValueError("the stand-in Dataset writes whole stores only (mode='w' or 'w-')")
        if mode == "w-" and os.path.exists(str(store)):  # xarray: "w-" means create, fail if the store exists
            raise FileExistsError(f"path {str(store)!r} contains a store (mode='w-' never overwrites)")
        write_dataset(str(store), self, chunks)

    def __repr__(self):  # pragma: no cover
        return f"<marex_amd.Dataset vars={list(self.data_vars)} attrs={list(self.attrs)}>"


if HAVE_XARRAY:  # pragma: no cover
    DataArray = _xr.DataArray
    Dataset = _xr.Dataset
else:
    DataArray = _MiniDataArray
    Dataset = _MiniDataset


def coord_values(da, name: str) -> np.ndarray:
    """Values of coordinate ``name`` of a DataArray (xarray or stand-in)."""
    return np.asarray(da.coords[name].values)


def to_numpy(da) -> np.ndarray:
    """Materialise the data of a DataArray on the host (computes Dask-backed xarray arrays)."""
    v = da.values
    return np.asarray(v)
