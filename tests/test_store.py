"""On-disk side of the index directory: reference-compatible files, mappable row store, per-rank
partial loads, stale-directory detection, byte-level faiss layout."""
import os
import pickle

import numpy as np
import pytest

import synth
from lotus_amd import HipVS, _capi, faiss_io, store
from oracle_backend import OracleBackend

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_faiss_flat_layout_against_hand_packed_bytes(tmp_path):
    """Reader and writer against bytes assembled WITHOUT them (tests/golden/make_faiss_fixture.py)."""
    from golden.make_faiss_fixture import IP_ROWS, L2_ROWS

    for name, rows, metric in (("faiss_flat_ip_3x4.index", IP_ROWS, 0), ("faiss_flat_l2_2x3.index", L2_ROWS, 1)):
        path = os.path.join(GOLDEN, name)
        x, m = faiss_io.read_index_flat(path)
        assert m == metric and x.dtype == np.float32 and np.array_equal(x, np.asarray(rows, np.float32))
        mm, m2 = faiss_io.mmap_index_flat(path)
        assert m2 == metric and np.array_equal(np.asarray(mm), x) and not mm.flags.writeable
        out = tmp_path / name
        faiss_io.write_index_flat(str(out), np.asarray(rows, np.float64), metric)  # fp64 in: cast like faiss's wrapper
        assert out.read_bytes() == open(path, "rb").read()  # byte for byte
    blob = open(os.path.join(GOLDEN, "faiss_flat_ip_3x4.index"), "rb").read()
    assert blob[:4] == b"IxFI" and len(blob) == 45 + 3 * 4 * 4  # 45-byte header, then the row-major float32 matrix
    bad = tmp_path / "bad.index"
    bad.write_bytes(b"IwFl" + blob[4:])
    with pytest.raises(ValueError):
        faiss_io.read_index_flat(str(bad))


@pytest.mark.parametrize("dtype,file,how", [(np.float32, "index", "mmap"), (np.float16, "rows.f16", "mmap"),
                                            (np.float64, "rows.f64", "mmap")])
def test_row_store_is_mapped_not_unpickled(tmp_path, dtype, file, how, monkeypatch):
    x = synth.corpus(300, 24, seed=1).astype(dtype)
    d = str(tmp_path / "i")
    store.write_dir(d, x, x, 0)
    assert sorted(os.listdir(d)) == sorted({"vecs", "index", "rows.json", file})
    with open(os.path.join(d, "vecs"), "rb") as fp:
        assert np.array_equal(pickle.load(fp), x)  # the reference's pickle, as given (faiss_vs.py:27-28)
    assert np.array_equal(faiss_io.read_index_flat(os.path.join(d, "index"))[0], x.astype(np.float32))
    monkeypatch.setattr(pickle, "load", lambda *a, **k: (_ for _ in ()).throw(AssertionError("unpickled")))
    rows, h = store.open_stored_rows(d)
    assert h == how and isinstance(rows, np.memmap) and rows.dtype == dtype and np.array_equal(np.asarray(rows), x)
    rows, h = store.open_device_rows(d)
    assert isinstance(rows, np.memmap) and np.array_equal(np.asarray(rows[100:200]), x[100:200])


def test_directory_written_by_stock_lotus_loads_without_unpickling_for_the_device_image(tmp_path, monkeypatch):
    """FaissVS.index leaves `vecs` (pickle) + `index` (faiss): the device image comes from the faiss file's code
    section (the float32 values faiss itself searches), the pickle is only read if the stored rows are asked for."""
    x = synth.corpus(200, 16, seed=2).astype(np.float64)  # LiteLLM-style fp64 embeddings (litellm_rm.py:69)
    d = str(tmp_path / "stock")
    os.makedirs(d)
    with open(os.path.join(d, "vecs"), "wb") as fp:
        pickle.dump(x, fp)
    faiss_io.write_index_flat(os.path.join(d, "index"), x, 0)
    real_load = pickle.load
    monkeypatch.setattr(pickle, "load", lambda *a, **k: (_ for _ in ()).throw(AssertionError("unpickled")))
    vs = HipVS(backend=OracleBackend())
    vs.load_index(d)
    out = vs(x[:5], 3)
    assert out.indices[:, 0].tolist() == [0, 1, 2, 3, 4]
    monkeypatch.setattr(pickle, "load", real_load)
    got = vs.get_vectors_from_index(d, [3, 1])
    assert got.dtype == np.float64 and np.array_equal(got, x[[3, 1]])  # stored dtype, from the pickle, on demand


def test_each_rank_reads_and_packs_only_its_rows(tmp_path, monkeypatch):
    x = synth.corpus(1001, 8, seed=3).astype(np.float16)
    d = str(tmp_path / "i")
    HipVS(backend=OracleBackend()).index(None, x, d)
    for rank, (lo, hi) in enumerate(((0, 501), (501, 1001))):
        vs = HipVS(backend=OracleBackend(), shard=True)
        monkeypatch.setattr(vs, "_dist", lambda r=rank: (r, 2))
        vs.load_index(d)
        ent = vs._resident[d]
        assert (ent.lo, ent.hi, ent.packed.n) == (lo, hi, hi - lo)
        assert vs.backend.calls[0] == ("pack", (hi - lo, 8), _capi.PACK_F16)  # only the shard went to the device
        assert np.array_equal(ent.packed.rows.numpy(), x[lo:hi].astype(np.float32))
        assert ent.vecs is None  # no host copy of the matrix is kept
        assert np.array_equal(vs.get_vectors_from_index(d, [1000, 0]), x[[1000, 0]])  # served from the memory map


def test_rewritten_directory_is_noticed(tmp_path):
    """a resident index must not be served stale after another process re-ran sem_index on its directory."""
    a, b = synth.corpus(50, 8, seed=4), synth.corpus(60, 8, seed=5)
    d = str(tmp_path / "i")
    vs = HipVS(backend=OracleBackend())
    vs.index(None, a, d)
    assert vs(a[7:8], 1).indices[0, 0] == 7
    other = HipVS(backend=OracleBackend())
    other.index(None, b, d)  # "another process" rewrites the directory
    os.utime(os.path.join(d, "index"), ns=(1, 1))  # even with a clock that went backwards: the signature differs
    vs.load_index(d)
    assert vs._resident[d].n == 60 and vs(b[55:56], 1).indices[0, 0] == 55
    n_packs = len(vs.backend.calls)
    vs.load_index(d)  # unchanged: served from HBM
    assert len([c for c in vs.backend.calls[n_packs:] if c[0] == "pack"]) == 0


def test_never_persisted_index_serves_vectors_from_the_device_image(tmp_path):
    x = synth.corpus(40, 8, seed=6).astype(np.float16)
    vs = HipVS(backend=OracleBackend())
    d = str(tmp_path / "nowhere")
    vs.index(None, x, d, persist=False)
    assert not os.path.exists(os.path.join(d, "vecs"))
    vs._resident[d].vecs = None  # e.g. the embeddings were a device tensor: no host array to fall back on
    got = vs.get_vectors_from_index(d, [5, 2])
    assert got.dtype == np.float16 and np.array_equal(got, x[[5, 2]])
    assert vs(x[:3], 1).indices[:, 0].tolist() == [0, 1, 2]


def test_non_flat_faiss_index_falls_back_to_the_pickle(tmp_path):
    """FaissVS(factory_string="IVF.."/"HNSW..") (faiss_vs.py:14,23,30) writes a non-IxF* `index`; such a
    directory must load from `vecs` instead of raising 'not a faiss IndexFlat file'."""
    x = synth.corpus(120, 16, seed=7)
    d = str(tmp_path / "ivf")
    os.makedirs(d)
    with open(os.path.join(d, "vecs"), "wb") as fp:
        pickle.dump(x, fp)
    with open(os.path.join(d, "index"), "wb") as fp:
        fp.write(b"IwFl" + b"\x00" * 100)  # an IVF index: another fourcc, another layout
    rows, how = store.open_device_rows(d)
    assert how == "pickle" and np.array_equal(rows, x)
    vs = HipVS(backend=OracleBackend())
    vs.load_index(d)
    assert vs(x[11:12], 1).indices[0, 0] == 11
    assert np.array_equal(vs.get_vectors_from_index(d, [4, 2]), x[[4, 2]])
    # ... also when a row-store description that points INTO the index file survived the overwrite
    d2 = str(tmp_path / "was_flat")
    HipVS(backend=OracleBackend()).index(None, x, d2)  # float32: rows.json says file == "index"
    with open(os.path.join(d2, "index"), "wb") as fp:
        fp.write(b"IwFl" + b"\x00" * 100)
    rows, how = store.open_device_rows(d2)
    assert how == "pickle" and np.array_equal(rows, x)
    with open(os.path.join(d2, "index"), "wb") as fp:  # truncated header: struct.error territory
        fp.write(b"IxFI\x01")
    assert store.open_device_rows(d2)[1] == "pickle"


def test_stale_raw_rows_are_not_served_after_a_foreign_rewrite(tmp_path):
    """stock LOTUS re-running sem_index into a directory HipVS wrote with fp16 embeddings leaves rows.f16 and
    rows.json behind; same n and d must not be enough to trust them."""
    a = synth.corpus(80, 8, seed=8).astype(np.float16)
    b = synth.corpus(80, 8, seed=9).astype(np.float32)  # same shape, other rows
    d = str(tmp_path / "i")
    vs = HipVS(backend=OracleBackend())
    vs.index(None, a, d)
    assert store.open_stored_rows(d)[1] == "mmap"
    with open(os.path.join(d, "vecs"), "wb") as fp:  # what FaissVS.index does (faiss_vs.py:27-30): only its two files
        pickle.dump(b, fp)
    faiss_io.write_index_flat(os.path.join(d, "index"), b, 0)
    rows, how = store.open_stored_rows(d)
    assert how == "pickle" and np.array_equal(rows, b)
    rows, how = store.open_device_rows(d)
    assert how == "index-mmap" and np.array_equal(np.asarray(rows), b)
    vs.load_index(d)  # the resident entry is stale too (signature)
    assert vs(b[33:34], 1).indices[0, 0] == 33
    assert np.array_equal(vs.get_vectors_from_index(d, [5]), b[[5]])


def test_get_vectors_notices_a_rewritten_directory_without_load_index(tmp_path):
    a, b = synth.corpus(30, 8, seed=10), synth.corpus(30, 8, seed=11)
    d = str(tmp_path / "i")
    vs = HipVS(backend=OracleBackend())
    vs.index(None, a, d)
    assert np.array_equal(vs.get_vectors_from_index(d, [3]), a[[3]])
    HipVS(backend=OracleBackend()).index(None, b, d)
    os.utime(os.path.join(d, "vecs"), ns=(5, 5))
    assert np.array_equal(vs.get_vectors_from_index(d, [3]), b[[3]])


def test_copied_directory_keeps_its_row_store(tmp_path):
    """rows.json used to record the mtime of `vecs` / `index`, so `cp -r` / rsync of an index directory voided
    the fp16 row store and get_vectors_from_index silently changed dtype.  The stamp is a content fingerprint now: a copy
    (other mtimes, same bytes) keeps serving the stored dtype from the memory map; a foreign rewrite still voids it."""
    import shutil
    import time

    a = synth.corpus(300, 24, seed=3).astype(np.float16)
    d = str(tmp_path / "i")
    HipVS(backend=OracleBackend()).index(None, a, d)
    d2 = str(tmp_path / "copy")
    shutil.copytree(d, d2)
    for name in ("vecs", "index", "rows.f16", "rows.json"):
        os.utime(os.path.join(d2, name), ns=(time.time_ns(), time.time_ns() - 7_000_000_000))  # other mtimes
    rows, how = store.open_stored_rows(d2)
    assert how == "mmap" and rows.dtype == np.float16 and np.array_equal(np.asarray(rows), a)
    vs = HipVS(backend=OracleBackend())
    vs.load_index(d2)
    got = vs.get_vectors_from_index(d2, [7, 9])
    assert got.dtype == np.float16 and np.array_equal(got, a[[7, 9]])
    # the one exact comparison re-stamped the copy: the next open is decided by (size, mtime) alone
    import json

    with open(os.path.join(d2, "rows.json")) as fp:
        assert json.load(fp)["written_with"]["index"][2] == os.stat(os.path.join(d2, "index")).st_mtime_ns
    b = synth.corpus(300, 24, seed=4).astype(np.float32)
    with open(os.path.join(d2, "vecs"), "wb") as fp:
        pickle.dump(b, fp)
    faiss_io.write_index_flat(os.path.join(d2, "index"), b, 0)
    assert store.open_stored_rows(d2)[1] == "pickle"


def test_same_shape_rewrite_of_a_few_middle_rows_voids_the_row_store(tmp_path):
    """a re-index by stock FaissVS that keeps n x d and changes only some rows leaves the file sizes and - almost
    surely - the 16 sampled 4 KB blocks as they were.  The row store must not be served on the strength of 64 KB of samples:
    untouched files are accepted by (size, mtime); anything else is compared with the faiss file exactly."""
    n, dim = 20_000, 64  # index file 5 MB: 16 sampled blocks cover 1.3 % of it
    a = synth.corpus(n, dim, seed=21).astype(np.float16)
    d = str(tmp_path / "i")
    vs = HipVS(backend=OracleBackend())
    vs.index(None, a, d)
    assert store.open_stored_rows(d)[1] == "mmap"
    # pick rows whose bytes lie in none of the sampled blocks of `index` (float32 codes from byte 45) or of `vecs` (the fp16
    # pickle: a header of < 256 bytes, then the matrix)
    def sampled(name):
        last = os.path.getsize(os.path.join(d, name)) - store._STAMP_BLOCK
        return [(last * i // (store._STAMP_BLOCKS - 1), last * i // (store._STAMP_BLOCKS - 1) + store._STAMP_BLOCK)
                for i in range(store._STAMP_BLOCKS)]

    def clear(r, name, head, item):
        return not any(lo - item * dim - 256 <= head + r * item * dim < hi + 256 for lo, hi in sampled(name))

    rows = [r for r in range(n // 2 - 200, n // 2 + 200) if clear(r, "index", 45, 4) and clear(r, "vecs", 0, 2)][:5]
    assert len(rows) == 5
    b = a.copy()
    b[rows] = synth.corpus(5, dim, seed=22).astype(np.float16)
    before = store._file_stamps(d)
    with open(os.path.join(d, "vecs"), "wb") as fp:  # FaissVS.index (faiss_vs.py:27-30): its two files, same shape and dtype
        pickle.dump(b, fp)
    faiss_io.write_index_flat(os.path.join(d, "index"), b.astype(np.float32), 0)
    after = store._file_stamps(d)
    for name in ("index", "vecs"):  # sizes and sampled fingerprints did not see the rewrite; only the mtimes moved
        assert after[name][:2] == before[name][:2] and after[name][2] != before[name][2]
    stored, how = store.open_stored_rows(d)
    assert how == "pickle" and np.array_equal(stored, b)
    dev_rows, how = store.open_device_rows(d)
    assert how == "index-mmap" and np.array_equal(np.asarray(dev_rows), b.astype(np.float32))
    vs.load_index(d)
    assert np.array_equal(vs.get_vectors_from_index(d, rows), b[rows])
    assert vs(b[rows[2]:rows[2] + 1].astype(np.float32), 1).indices[0, 0] == rows[2]
