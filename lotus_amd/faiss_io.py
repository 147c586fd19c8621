"""Reader/writer for the faiss ``IndexFlat`` file the reference persists next to ``vecs``
(``faiss.write_index`` / ``read_index``, ``lotus/vector_store/faiss_vs.py:30,34``), so index directories are
interchangeable with stock LOTUS.

Layout (little-endian; recalled from faiss ``impl/index_write.cpp`` - SURVEY.md Appendix A.5 - and NOT verifiable
here because faiss is not installable in this image): fourcc ``IxFI`` (inner product) / ``IxF2`` (L2);
``int32 d, int64 ntotal, int64 dummy, int64 dummy, uint8 is_trained, int32 metric_type``; then the flat codes as
``uint64 count`` (in 4-byte units) followed by ``count`` float32 values.
"""
from __future__ import annotations

import struct

import numpy as np

_DUMMY = 1 << 20


def write_index_flat(path: str, x, metric: int) -> None:
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    fourcc = b"IxFI" if metric == 0 else b"IxF2"
    with open(path, "wb") as f:
        f.write(fourcc)
        f.write(struct.pack("<iqqqBi", d, n, _DUMMY, _DUMMY, 1, int(metric)))
        f.write(struct.pack("<Q", n * d))
        x.tofile(f)


def read_index_flat(path: str):
    """-> (float32 [ntotal, d], metric)."""
    with open(path, "rb") as f:
        fourcc = f.read(4)
        if fourcc not in (b"IxFI", b"IxF2", b"IxFl"):
            raise ValueError(f"{path}: not a faiss IndexFlat file (fourcc {fourcc!r})")
        d, n, _, _, _, metric = struct.unpack("<iqqqBi", f.read(4 + 8 + 8 + 8 + 1 + 4))
        if metric > 1:
            f.read(4)  # metric_arg
        (count,) = struct.unpack("<Q", f.read(8))
        if count != n * d:
            raise ValueError(f"{path}: code size {count} does not match ntotal*d = {n * d}")
        x = np.fromfile(f, dtype=np.float32, count=count).reshape(n, d)
    return x, metric
