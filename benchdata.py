"""Synthetic inputs of the BASELINE configs, regenerable by anyone from (config, block) alone - SURVEY.md section 8(d).

Every 1 M-row block b of config c comes from ONE numpy stream, ``numpy.random.default_rng(SeedSequence([20260923, c, b]))``,
drawn in row order (the draws of a block may be split into sub-chunks: the stream, hence the values, is the same), on the
host, in float32, then cast to the config's storage dtype.  No GPU generator is involved, so the inputs behind a BENCH line
can be rebuilt on any machine with numpy; blocks are independent, so a rank (or a thread) generates only the blocks it needs.

    corpus(cfg, n, d)       rows i.i.d. N(0, 1) -> L2-normalised in float32 -> storage dtype            (configs[1], [2])
    queries(cfg, xb, nq)    q_i = normalize(0.7 x[j_i] + 0.7 u_i), j_i ~ U[0, n), u_i a random unit vector: a planted
                            neighbour at cos ~ 0.71 far above the random background                      (stream block 1000)
    dedup_rows(cfg, n)      base rows + planted near-duplicates normalize(x_j + 0.2 u) (cos ~ 0.98), chains dup-of-dup,
                            hard negatives normalize(x_j + 0.5 u) (cos ~ 0.89); returns the plant table   (configs[3])
    blobs(cfg, n, d, K)     mixture of K blobs normalize(c_m + 0.3 g), |g| ~ 1                           (configs[4])

This module is input generation for bench.py and tests/ only; nothing under lotus_amd/ imports it.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

SEED = 20260923
BLOCK_ROWS = 1_000_000
CHUNK_ROWS = 65_536  # rows drawn per call inside a block (bounds the float32 temporaries; does not change the values)
CFG_SEARCH, CFG_JOIN, CFG_DEDUP, CFG_KMEANS = 2, 3, 4, 5  # 1-based config numbers of BASELINE.md section 2
QUERY_STREAM = 1000   # stream index of a config's queries
AUX_STREAM = 2000     # ... of its auxiliary draws (blob centres, plant table)


def block_rng(cfg: int, block: int) -> np.random.Generator:
    return np.random.default_rng(np.random.SeedSequence([SEED, int(cfg), int(block)]))


def _normalize_rows(x: np.ndarray) -> np.ndarray:
    x /= np.sqrt(np.einsum("ij,ij->i", x, x))[:, None]
    return x


def default_threads() -> int:
    return max(1, min(32, (os.cpu_count() or 1)))


def _run_blocks(fn, nblocks: int, threads: int | None):
    threads = default_threads() if threads is None else threads
    if nblocks <= 1 or threads <= 1:
        return [fn(b) for b in range(nblocks)]
    with ThreadPoolExecutor(min(threads, nblocks)) as ex:  # numpy releases the GIL while it draws
        return list(ex.map(fn, range(nblocks)))


def corpus_block(cfg: int, block: int, rows: int, d: int, dtype=np.float16, out=None) -> np.ndarray:
    """Rows [block * 1M, block * 1M + rows) of config ``cfg``'s corpus (written into ``out`` when given)."""
    rng = block_rng(cfg, block)
    if out is None:
        out = np.empty((rows, d), dtype)
    for r0 in range(0, rows, CHUNK_ROWS):
        r1 = min(rows, r0 + CHUNK_ROWS)
        out[r0:r1] = _normalize_rows(rng.standard_normal((r1 - r0, d), dtype=np.float32))
    return out


def corpus(cfg: int, n: int, d: int, dtype=np.float16, threads: int | None = None, rows=None, out=None) -> np.ndarray:
    """The first ``n`` rows of the corpus (``rows=(lo, hi)``: only that slice - whole blocks are still drawn, a stream
    cannot be entered in the middle).  ``out``: a preallocated [n, d] array to fill (whole-corpus form only)."""
    lo, hi = (0, n) if rows is None else rows
    b0, b1 = lo // BLOCK_ROWS, -(-hi // BLOCK_ROWS)
    r_lo, r_hi = b0 * BLOCK_ROWS, min(n, b1 * BLOCK_ROWS)
    buf = out if (out is not None and rows is None) else np.empty((r_hi - r_lo, d), dtype)

    def run(i):
        s0 = i * BLOCK_ROWS
        s1 = min(r_hi - r_lo, s0 + BLOCK_ROWS)
        corpus_block(cfg, b0 + i, s1 - s0, d, dtype, out=buf[s0:s1])

    _run_blocks(run, b1 - b0, threads)
    return buf[lo - r_lo:hi - r_lo]


def queries(cfg: int, xb: np.ndarray, nq: int, n_total: int | None = None, dtype=np.float16):
    """-> (xq [nq, d], planted [nq]): query i is a noisy copy of corpus row planted[i].  ``xb`` must hold every row a
    query is planted on: the whole corpus (``n_total`` None) or any superset of rows [0, n_total)."""
    n = int(xb.shape[0] if n_total is None else n_total)
    d = int(xb.shape[1])
    rng = block_rng(cfg, QUERY_STREAM)
    j = rng.integers(0, n, nq)
    out = np.empty((nq, d), dtype)
    for r0 in range(0, nq, CHUNK_ROWS):
        r1 = min(nq, r0 + CHUNK_ROWS)
        u = _normalize_rows(rng.standard_normal((r1 - r0, d), dtype=np.float32))
        q = np.float32(0.7) * xb[j[r0:r1]].astype(np.float32) + np.float32(0.7) * u
        out[r0:r1] = _normalize_rows(q)
    return out, j


# ---- configs[3]: threshold self-join with planted near-duplicates -------------------------------------------------------
def dedup_layout(n: int):
    """Row ranges of the planted structure for an ``n``-row input (5 M at full size: 4.4 M base rows, 0.4 M direct
    near-duplicates of distinct base rows, 0.1 M duplicates of duplicates (chains), 0.1 M hard negatives)."""
    n_dup = n * 8 // 100
    n_chain = n * 2 // 100
    n_neg = n * 2 // 100
    n_base = n - n_dup - n_chain - n_neg
    return n_base, n_dup, n_chain, n_neg


def dedup_rows(cfg: int, n: int, d: int, dtype=np.float16, threads: int | None = None):
    """-> (x [n, d], plants): ``plants`` = dict(row, src, kind) for every planted row (kind 0 direct duplicate, 1 chain,
    2 hard negative); ``src`` is the row it was derived from (always a lower row).  Base rows are config ``cfg``'s
    corpus; planted rows are drawn from streams AUX_STREAM + block of the planted region."""
    n_base, n_dup, n_chain, n_neg = dedup_layout(n)
    x = np.empty((n, d), dtype)
    corpus(cfg, n_base, d, dtype, threads, out=x[:n_base])
    aux = block_rng(cfg, AUX_STREAM)
    src_dup = aux.choice(n_base, n_dup, replace=False)                 # distinct base rows: no sibling pairs
    src_chain = n_base + aux.choice(n_dup, n_chain, replace=False)     # distinct direct duplicates
    src_neg = aux.choice(n_base, n_neg, replace=False)
    row0 = [n_base, n_base + n_dup, n_base + n_dup + n_chain]
    specs = [(row0[0], src_dup, 0.2), (row0[1], src_chain, 0.2), (row0[2], src_neg, 0.5)]
    # planted rows in 1 M-row blocks of the planted region; chains need the direct duplicates first (two phases)
    for phase in (0, 1):
        jobs = []
        for si, (r0, src, eps) in enumerate(specs):
            if (si == 1) != (phase == 1):
                continue
            for b0 in range(0, len(src), BLOCK_ROWS):
                jobs.append((si, r0, src, eps, b0, min(len(src), b0 + BLOCK_ROWS)))

        def run(i):
            si, r0, src, eps, b0, b1 = jobs[i]
            rng = block_rng(cfg, AUX_STREAM + 1 + si * 100 + b0 // BLOCK_ROWS)
            for c0 in range(b0, b1, CHUNK_ROWS):
                c1 = min(b1, c0 + CHUNK_ROWS)
                u = _normalize_rows(rng.standard_normal((c1 - c0, d), dtype=np.float32))
                y = x[src[c0:c1]].astype(np.float32) + np.float32(eps) * u
                x[r0 + c0:r0 + c1] = _normalize_rows(y)

        _run_blocks(run, len(jobs), threads)
    plants = {"row": np.concatenate([r0 + np.arange(len(src)) for r0, src, _ in specs]).astype(np.int64),
              "src": np.concatenate([src for _, src, _ in specs]).astype(np.int64),
              "kind": np.concatenate([np.full(len(src), ki, np.int8) for ki, (_, src, _) in enumerate(specs)])}
    return x, plants


def dedup_expected_pairs(x: np.ndarray, plants: dict, threshold: float, band: float = 2e-5):
    """Pairs (i < j) the planted structure implies, from float32 cosines of the STORED values computed on the host:
    (planted row, its source) and, for chains, (planted row, its source's source).  -> (sure, maybe): ``sure`` pairs are
    above threshold + band, ``maybe`` pairs lie within the band (either answer is right there).  Sets of (i, j)."""
    row, src, kind = plants["row"], plants["src"], plants["kind"]
    cand_a, cand_b = [src], [row]
    chain = kind == 1
    src_of = dict(zip(row[kind == 0].tolist(), src[kind == 0].tolist()))
    grand = np.array([src_of[s] for s in src[chain].tolist()], dtype=np.int64)
    cand_a.append(grand)
    cand_b.append(row[chain])
    a, b = np.concatenate(cand_a), np.concatenate(cand_b)
    cos = np.empty(len(a), np.float32)
    for c0 in range(0, len(a), CHUNK_ROWS):
        c1 = min(len(a), c0 + CHUNK_ROWS)
        cos[c0:c1] = np.einsum("ij,ij->i", x[a[c0:c1]].astype(np.float32), x[b[c0:c1]].astype(np.float32))
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    sure = cos > threshold + band
    maybe = (~sure) & (cos > threshold - band)
    return set(zip(lo[sure].tolist(), hi[sure].tolist())), set(zip(lo[maybe].tolist(), hi[maybe].tolist()))


# ---- configs[4]: k-means on a mixture of blobs -------------------------------------------------------------------------
def blob_centres(cfg: int, k: int, d: int) -> np.ndarray:
    return _normalize_rows(block_rng(cfg, AUX_STREAM).standard_normal((k, d), dtype=np.float32))


def blobs_block(cfg: int, block: int, rows: int, d: int, centres: np.ndarray, out: np.ndarray, lab: np.ndarray) -> None:
    rng = block_rng(cfg, block)
    k = centres.shape[0]
    noise = np.float32(0.3 / np.sqrt(d))  # 0.3 g with |g| ~ 1
    for r0 in range(0, rows, CHUNK_ROWS):
        r1 = min(rows, r0 + CHUNK_ROWS)
        m = rng.integers(0, k, r1 - r0)
        g = rng.standard_normal((r1 - r0, d), dtype=np.float32)
        g *= noise
        g += centres[m]
        out[r0:r1] = _normalize_rows(g)
        lab[r0:r1] = m


def blobs(cfg: int, n: int, d: int, k: int, dtype=np.float16, threads: int | None = None):
    """-> (x [n, d], label [n]) - row i belongs to blob label[i] (cos to its centre ~ 0.96)."""
    centres = blob_centres(cfg, k, d)
    x = np.empty((n, d), dtype)
    lab = np.empty(n, np.int64)

    def run(b):
        s0, s1 = b * BLOCK_ROWS, min(n, (b + 1) * BLOCK_ROWS)
        blobs_block(cfg, b, s1 - s0, d, centres, x[s0:s1], lab[s0:s1])

    _run_blocks(run, -(-n // BLOCK_ROWS), threads)
    return x, lab
