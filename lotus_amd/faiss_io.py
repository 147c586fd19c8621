"""Reader/writer for the faiss ``IndexFlat`` file the reference persists next to ``vecs``
(``faiss.write_index`` / ``read_index``, ``lotus/vector_store/faiss_vs.py:30,34``), so index directories are
interchangeable with stock LOTUS.

Layout (little-endian; faiss ``impl/index_write.cpp``: ``write_index`` -> fourcc, ``write_index_header``,
``WRITEXBVECTOR(codes)`` - recalled; faiss is not installable in this image, so the layout is
pinned by a hand-packed byte fixture, ``tests/golden/faiss_flat_*.index``, not by a faiss-written file):

    offset  0  char[4]  fourcc  "IxFI" (inner product) / "IxF2" (L2) / "IxFl" (other metrics)
            4  int32    d
            8  int64    ntotal
           16  int64    dummy (1 << 20)
           24  int64    dummy (1 << 20)
           32  uint8    is_trained
           33  int32    metric_type          (+ float32 metric_arg when metric_type > 1)
           37  uint64   size of the code vector in 4-byte units (= ntotal * d)
           45  float32  codes[ntotal * d]    <- the flat row-major matrix: mappable in place
"""
from __future__ import annotations

import struct

import numpy as np

_DUMMY = 1 << 20
_HEADER = "<iqqqBi"


def write_index_flat(path: str, x, metric: int) -> None:
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    fourcc = b"IxFI" if metric == 0 else b"IxF2"
    with open(path, "wb") as f:
        f.write(fourcc)
        f.write(struct.pack(_HEADER, d, n, _DUMMY, _DUMMY, 1, int(metric)))
        f.write(struct.pack("<Q", n * d))
        x.tofile(f)


def _read_header(f, path: str):
    fourcc = f.read(4)
    if fourcc not in (b"IxFI", b"IxF2", b"IxFl"):
        raise ValueError(f"{path}: not a faiss IndexFlat file (fourcc {fourcc!r})")
    d, n, _, _, _, metric = struct.unpack(_HEADER, f.read(struct.calcsize(_HEADER)))
    if metric > 1:
        f.read(4)  # metric_arg
    (count,) = struct.unpack("<Q", f.read(8))
    if count != n * d:
        raise ValueError(f"{path}: code size {count} does not match ntotal*d = {n * d}")
    return n, d, metric, f.tell()


def read_index_flat(path: str):
    """-> (float32 [ntotal, d], metric)."""
    with open(path, "rb") as f:
        n, d, metric, _ = _read_header(f, path)
        x = np.fromfile(f, dtype=np.float32, count=n * d).reshape(n, d)
    return x, metric


def mmap_index_flat(path: str):
    """-> (read-only float32 memmap [ntotal, d] over the file's code section, metric): no copy, pages are read on
    first touch - a rank that only needs rows [lo, hi) reads only those."""
    with open(path, "rb") as f:
        n, d, metric, off = _read_header(f, path)
    if n == 0:
        return np.zeros((0, d), np.float32), metric
    return np.memmap(path, dtype=np.float32, mode="r", offset=off, shape=(n, d)), metric
