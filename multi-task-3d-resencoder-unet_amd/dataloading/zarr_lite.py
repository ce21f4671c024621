"""Minimal zarr **v2** array reader / writer (the `zarr` package is not in this image).

Covers what the reference's patch feeder needs from `zarr.open(path, mode='r')[z0:z1, y0:y1, x0:x1]`
(dataloading/dataset.py:103-137, helpers.py:71-128): a directory store with a `.zarray` JSON (shape, chunks, dtype,
order, fill_value, compressor, dimension_separator), chunk files `i.j.k` (or `i/j/k`), partial edge chunks stored at
full chunk size, missing chunks = fill value.  Compressors: none, `zlib`, `gzip` (Python's zlib); `blosc` and any filter
raise -- the codecs live in `numcodecs`, which is absent too.  Basic indexing only (ints and unit-step slices).
Format knowledge comes from the public zarr v2 spec; nothing here could be cross-checked against the real package."""
import builtins
import gzip
import json
import os
import zlib

import numpy as np


class ZarrLiteError(RuntimeError):
    pass


class Array:
    def __init__(self, path):
        self.path = str(path)
        meta_file = os.path.join(self.path, ".zarray")
        if not os.path.exists(meta_file):
            if os.path.exists(os.path.join(self.path, ".zgroup")):
                raise ZarrLiteError(f"{path} is a zarr GROUP; open one of its arrays (e.g. {path}/0)")
            raise ZarrLiteError(f"{path}: no .zarray (zarr v2 directory stores only)")
        with builtins.open(meta_file) as f:
            m = json.load(f)
        if m.get("zarr_format") != 2:
            raise ZarrLiteError(f"{path}: zarr_format {m.get('zarr_format')} (only v2)")
        if m.get("filters"):
            raise ZarrLiteError(f"{path}: filters {m['filters']} need numcodecs")
        comp = m.get("compressor")
        self._codec = None if comp is None else comp.get("id")
        if self._codec not in (None, "zlib", "gzip"):
            raise ZarrLiteError(f"{path}: compressor '{self._codec}' needs numcodecs (supported: none, zlib, gzip)")
        self.shape = tuple(m["shape"])
        self.chunks = tuple(m["chunks"])
        self.dtype = np.dtype(m["dtype"])
        self.order = m.get("order", "C")
        fv = m.get("fill_value")
        self.fill_value = 0 if fv is None else fv
        self._sep = m.get("dimension_separator", ".")
        self.ndim = len(self.shape)
        self.store = self           # `arr.store.close()` of the reference's loop is a no-op here

    def close(self):
        pass

    @property
    def size(self):
        return int(np.prod(self.shape))

    def _chunk(self, idx):
        name = self._sep.join(str(i) for i in idx)
        fn = os.path.join(self.path, *name.split("/")) if self._sep == "/" else os.path.join(self.path, name)
        if not os.path.exists(fn):
            return None
        with builtins.open(fn, "rb") as f:
            raw = f.read()
        if self._codec == "zlib":
            raw = zlib.decompress(raw)
        elif self._codec == "gzip":
            raw = gzip.decompress(raw)
        a = np.frombuffer(raw, dtype=self.dtype)
        if a.size != int(np.prod(self.chunks)):
            raise ZarrLiteError(f"{fn}: {a.size} elements, expected a full chunk of {self.chunks}")
        return a.reshape(self.chunks, order=self.order)

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = key.index(Ellipsis)
            key = key[:i] + (slice(None),) * (self.ndim - len(key) + 1) + key[i + 1:]
        key = key + (slice(None),) * (self.ndim - len(key))
        lo, hi, squeeze = [], [], []
        for d, k in enumerate(key):
            if isinstance(k, (int, np.integer)):
                k = int(k) + (self.shape[d] if k < 0 else 0)
                if not 0 <= k < self.shape[d]:
                    raise IndexError(f"index {k} out of range for axis {d}")
                lo.append(k), hi.append(k + 1), squeeze.append(d)
            elif isinstance(k, slice):
                a, b, st = k.indices(self.shape[d])
                if st != 1:
                    raise ZarrLiteError("only unit-step slices")
                lo.append(a), hi.append(max(a, b))
            else:
                raise ZarrLiteError(f"unsupported index {k!r}")
        out = np.empty([h - l for l, h in zip(lo, hi)], dtype=self.dtype)
        if out.size:
            first = [l // c for l, c in zip(lo, self.chunks)]
            last = [(h - 1) // c for h, c in zip(hi, self.chunks)]
            for idx in np.ndindex(*[b - a + 1 for a, b in zip(first, last)]):
                cidx = tuple(a + i for a, i in zip(first, idx))
                src, dst = [], []
                for d, ci in enumerate(cidx):
                    c0 = ci * self.chunks[d]
                    a, b = max(lo[d], c0), min(hi[d], c0 + self.chunks[d])
                    src.append(slice(a - c0, b - c0)), dst.append(slice(a - lo[d], b - lo[d]))
                ch = self._chunk(cidx)
                out[tuple(dst)] = self.fill_value if ch is None else ch[tuple(src)]
        return out.reshape([s for d, s in enumerate(out.shape) if d not in squeeze]) if squeeze else out


def open(path, mode="r"):   # noqa: A001  (mirrors zarr.open)
    if mode != "r":
        raise ZarrLiteError("read-only (use write_array to create a store)")
    return Array(path)


def write_array(path, data, chunks, compressor=None, fill_value=0, dimension_separator="."):
    """write `data` as a zarr v2 directory store (compressor None or 'zlib'); all-fill chunks are skipped like zarr does"""
    data = np.asarray(data)
    chunks = tuple(int(c) for c in chunks)
    if compressor not in (None, "zlib"):
        raise ZarrLiteError("write_array: compressor None or 'zlib'")
    os.makedirs(path, exist_ok=True)
    meta = dict(zarr_format=2, shape=list(data.shape), chunks=list(chunks), dtype=data.dtype.str, order="C",
                fill_value=fill_value, filters=None, dimension_separator=dimension_separator,
                compressor=None if compressor is None else {"id": "zlib", "level": 1})
    with builtins.open(os.path.join(path, ".zarray"), "w") as f:
        json.dump(meta, f)
    grid = [(s + c - 1) // c for s, c in zip(data.shape, chunks)]
    for idx in np.ndindex(*grid):
        block = np.full(chunks, fill_value, dtype=data.dtype)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, data.shape))
        sub = data[sl]
        block[tuple(slice(0, n) for n in sub.shape)] = sub
        if not (block != fill_value).any():
            continue
        raw = block.tobytes(order="C")
        if compressor == "zlib":
            raw = zlib.compress(raw, 1)
        name = dimension_separator.join(str(i) for i in idx)
        fn = os.path.join(path, *name.split("/")) if dimension_separator == "/" else os.path.join(path, name)
        os.makedirs(os.path.dirname(fn), exist_ok=True)
        with builtins.open(fn, "wb") as f:
            f.write(raw)
    return Array(path)
