"""On-disk side of the index directory (``lotus/vector_store/faiss_vs.py:27-41``).

The reference keeps two files per index: ``{dir}/vecs`` (a Python pickle of the embeddings, re-read IN FULL by every
``get_vectors_from_index`` and every ``ids``-branch search, ``faiss_vs.py:38-41,59``) and ``{dir}/index`` (faiss binary).
Both are still written, so a directory built here loads in stock LOTUS and vice versa.  On top of that:

* rows are served from a memory map, never from an unpickle: float32 embeddings are mapped in place inside
  ``{dir}/index`` (its code section IS the row-major matrix); other storage types (fp16, fp64) get a raw
  ``{dir}/rows.f16`` / ``rows.f64`` written next to it, described by ``{dir}/rows.json``;
* a rank that owns rows ``[lo, hi)`` touches only those pages (per-rank partial load);
* ``signature()`` = (size, mtime) of the files an index was loaded from, so a directory rewritten by another
  process is noticed instead of served stale; ``rows.json`` records size, mtime and a sampled content stamp of the two
  reference files it was written with: untouched files are accepted at once, a copied directory keeps its row store after
  ONE exact comparison with the faiss file, and a foreign rewrite - even of a few rows of the same shape - voids it.
"""
from __future__ import annotations

import json
import os
import pickle
import struct

import numpy as np

from . import faiss_io

RAW_VERSION = 1
_DTYPES = {"f16": np.float16, "f32": np.float32, "f64": np.float64}


def _dtype_tag(dt) -> str:
    for tag, t in _DTYPES.items():
        if np.dtype(dt) == np.dtype(t):
            return tag
    raise ValueError(f"unsupported storage dtype {dt}")


def write_dir(index_dir: str, embeddings, host: np.ndarray, metric: int, raw: bool = True) -> None:
    """Persist an index: the reference's two files + (``raw``) the mappable row store.  ``embeddings`` is what the caller
    handed to ``index()`` (pickled as is when it is an ndarray, exactly as ``faiss_vs.py:27-28``), ``host`` its 2-D host
    image."""
    os.makedirs(index_dir, exist_ok=True)
    meta_path = os.path.join(index_dir, "rows.json")
    if os.path.exists(meta_path):
        os.remove(meta_path)  # never leave a description of rows that are being replaced
    with open(os.path.join(index_dir, "vecs"), "wb") as fp:
        pickle.dump(embeddings if isinstance(embeddings, np.ndarray) else host, fp)
    faiss_io.write_index_flat(os.path.join(index_dir, "index"), host, metric)
    if not raw:
        return
    tag = _dtype_tag(host.dtype)
    meta = {"version": RAW_VERSION, "n": int(host.shape[0]), "d": int(host.shape[1]), "dtype": tag, "metric": int(metric),
            # the reference's two files as they are NOW: a writer that does not know about rows.json (stock LOTUS re-running
            # sem_index into this directory) changes them, and the description below is then void (_described_rows)
            "written_with": _file_stamps(index_dir)}
    if tag == "f32":
        meta["file"] = "index"  # mapped in place: the faiss file's code section is the float32 matrix
    else:
        meta["file"] = f"rows.{tag}"
        np.ascontiguousarray(host).tofile(os.path.join(index_dir, meta["file"]))
    tmp = meta_path + ".tmp"
    with open(tmp, "w") as fp:
        json.dump(meta, fp)
    os.replace(tmp, meta_path)  # the description appears only once the rows are complete


_STAMP_BLOCK = 4096
_STAMP_BLOCKS = 16


def _fingerprint(path: str, size: int) -> str:
    """Content stamp of a file that survives a copy: blake2b over its size and 16 evenly spaced 4 KB blocks (first and last
    included) - 64 KB read whatever the file's size.  A re-index with other embeddings changes the sampled rows; an mtime
    (round 3's stamp) changed with every `cp -r` / rsync and voided a perfectly good row store."""
    import hashlib

    h = hashlib.blake2b(digest_size=16)
    h.update(str(size).encode())
    with open(path, "rb") as fp:
        last = max(0, size - _STAMP_BLOCK)
        for i in range(_STAMP_BLOCKS):
            fp.seek(last * i // (_STAMP_BLOCKS - 1))
            h.update(fp.read(_STAMP_BLOCK))
    return h.hexdigest()


def _file_stamps(index_dir: str) -> dict:
    """[size, sampled fingerprint, mtime_ns] of the reference's two files."""
    out = {}
    for name in ("index", "vecs"):
        path = os.path.join(index_dir, name)
        try:
            st = os.stat(path)
            out[name] = [int(st.st_size), _fingerprint(path, int(st.st_size)), int(st.st_mtime_ns)]
        except FileNotFoundError:
            out[name] = [-1, "", -1]
    return out


def _row_store_matches_index(index_dir: str, meta: dict) -> bool:
    """EXACT check that ``rows.f16|f64`` still holds the embeddings ``{dir}/index`` was written from: the faiss file's code
    section is the float32 cast of the embeddings (``faiss_vs.py:24``), so the row store cast to float32 must equal it value
    for value.  One sequential pass over both files, only taken when the cheap (size, mtime) test could not decide."""
    codes = _mmap_flat_or_none(os.path.join(index_dir, "index"))
    n, d, tag = int(meta["n"]), int(meta["d"]), meta["dtype"]
    if codes is None or codes.shape != (n, d):
        return False
    path = os.path.join(index_dir, meta["file"])
    if not os.path.exists(path) or os.path.getsize(path) != n * d * np.dtype(_DTYPES[tag]).itemsize:
        return False
    if n == 0:
        return True
    rows = np.memmap(path, dtype=_DTYPES[tag], mode="r", shape=(n, d))
    step = max(1, (64 << 20) // (4 * d))
    for r0 in range(0, n, step):
        a = np.asarray(rows[r0:r0 + step]).astype(np.float32)
        if not np.array_equal(a.view(np.uint32), np.asarray(codes[r0:r0 + step]).view(np.uint32)):
            return False
    return True


def _stamps_still_valid(index_dir: str, meta: dict) -> bool:
    """Is the row store ``rows.json`` describes still the one the reference's two files were written with?

    * size of ``index`` / ``vecs`` changed                     -> no;
    * size and mtime both as recorded                          -> yes (nobody touched the files);
    * same size, other mtime (``cp -r`` / rsync / a re-index of the same shape by a writer that does not know rows.json):
      the sampled fingerprint must match - 64 KB of a file of gigabytes cannot rule out a rewrite of a few rows, so it only
      screens - and then the row store is compared with the faiss file's code section EXACTLY.  A float32 store is the faiss
      file itself (mapped in place): whatever it holds now is what a search must see, the fingerprint screen is enough.
    After a full comparison the new mtimes are recorded (best effort) so that the next open is cheap again."""
    stamps = meta.get("written_with")
    if stamps is None:
        return True
    now = _file_stamps(index_dir)
    same_time = True
    for name in ("index", "vecs"):
        was = list(stamps.get(name, [-2, ""]))
        if was[0] != now[name][0]:
            return False
        same_time &= len(was) >= 3 and was[2] == now[name][2]
    if same_time:
        return True
    if any(list(stamps[name])[1] != now[name][1] for name in ("index", "vecs")):
        return False
    if meta.get("file") == "index":
        return True
    if not _row_store_matches_index(index_dir, meta):
        return False
    try:
        meta = dict(meta, written_with=now)
        # several ranks may open the same copied directory at once: every writer gets a temporary file of its own, the rename
        # is atomic, the last one wins (they all write the same stamps)
        import tempfile

        fd, tmp = tempfile.mkstemp(prefix="rows.json.", suffix=".tmp", dir=index_dir)
        try:
            with os.fdopen(fd, "w") as fp:
                json.dump(meta, fp)
            os.replace(tmp, os.path.join(index_dir, "rows.json"))
        except OSError:
            try:
                os.unlink(tmp)
            except OSError:
                pass
    except OSError:
        pass
    return True


def _mmap_flat_or_none(path: str):
    """The code section of a faiss ``IndexFlat`` file as a memmap, or None when the file is something else -
    ``FaissVS(factory_string="IVF.." / "HNSW..")`` (``faiss_vs.py:14,23,30``) writes a non-flat index, whose rows can only
    come from the ``vecs`` pickle."""
    try:
        return faiss_io.mmap_index_flat(path)[0]
    except (ValueError, struct.error, OSError):
        return None


def signature(index_dir: str):
    sig = []
    for name in ("rows.json", "index", "vecs"):
        try:
            st = os.stat(os.path.join(index_dir, name))
            sig.append((name, st.st_size, st.st_mtime_ns))
        except FileNotFoundError:
            sig.append((name, -1, -1))
    return tuple(sig)


def _described_rows(index_dir: str):
    """The memmap ``rows.json`` describes, or None."""
    meta_path = os.path.join(index_dir, "rows.json")
    if not os.path.exists(meta_path):
        return None
    try:
        with open(meta_path) as fp:
            meta = json.load(fp)
        n, d, tag = int(meta["n"]), int(meta["d"]), meta["dtype"]
    except (OSError, ValueError, KeyError, TypeError):
        return None
    if meta.get("version") != RAW_VERSION or tag not in _DTYPES:
        return None
    if not _stamps_still_valid(index_dir, meta):
        return None  # `index` / `vecs` were rewritten by a writer that left this description (and rows.f16|f64) behind
    if meta["file"] == "index":
        rows = _mmap_flat_or_none(os.path.join(index_dir, "index"))
        return rows if (rows is not None and rows.shape == (n, d)) else None
    path = os.path.join(index_dir, meta["file"])
    if not os.path.exists(path) or os.path.getsize(path) != n * d * np.dtype(_DTYPES[tag]).itemsize:
        return None
    if n == 0:
        return np.zeros((0, d), _DTYPES[tag])
    return np.memmap(path, dtype=_DTYPES[tag], mode="r", shape=(n, d))


def open_stored_rows(index_dir: str):
    """-> (rows [n, d] in the STORED dtype, how).  What ``get_vectors_from_index`` serves (``faiss_vs.py:38-41``): a
    read-only memmap when the directory carries the row store ("raw" / "index-mmap"), else the unpickled ``vecs`` of a
    directory written by stock LOTUS ("pickle")."""
    rows = _described_rows(index_dir)
    if rows is not None:
        return rows, "mmap"
    with open(os.path.join(index_dir, "vecs"), "rb") as fp:
        return np.asarray(pickle.load(fp)), "pickle"


def open_device_rows(index_dir: str):
    """-> (rows [n, d], how) to build the device image from; always mappable: the row store when present, else the code
    section of ``{dir}/index`` - the float32 cast of the embeddings, i.e. exactly the values faiss itself searches
    (``faiss_vs.py:34,75``).  A rank slices ``rows[lo:hi]`` and reads only those pages; nothing is unpickled."""
    rows = _described_rows(index_dir)
    if rows is not None:
        return rows, "mmap"
    idx = os.path.join(index_dir, "index")
    if os.path.exists(idx):
        rows = _mmap_flat_or_none(idx)
        if rows is not None:
            return rows, "index-mmap"
    with open(os.path.join(index_dir, "vecs"), "rb") as fp:  # no mappable file: only the pickle has the rows
        return np.asarray(pickle.load(fp)), "pickle"
