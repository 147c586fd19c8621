"""GPU parity of the register-resident-queries kernels at join scale (round 6: `lvs_rj_kernel` - one wave per SIMD, 64 queries per
wave, deferred insertions - and the chunked launches of `lvs_rq_kernel` / `lvs_rj_kernel` beyond 4 096 queries) against the CPU
oracle.  Replaces the reference's `faiss_vs.py:67,75` search as reached from `sem_sim_join.py:132-134`.

Every group count 1 .. 16 of a single launch, the chunk shapes of larger calls (32 / 64 / 128 groups sharing a corpus range per XCD,
plus the 4 096-query and ragged last chunks), d = 256 / 384 / 512 / 768, inner product and squared L2, k = 1 .. 16, ragged corpora
(the kernel sees whole 32-row blocks, the last nb % 32 rows go through `lvs_rq_kernel`), exact duplicates across block and
range boundaries, and a block of hundreds of equal rows (the candidate buffer's overflow path).  The corpus is long (the
kernels need >= 32 768 rows per group), so the oracle checks a STRATIFIED SAMPLE of the queries - two per 32-query block, i.e.
every wave of every workgroup - while the device answers all of them; `timing_read_full()["kernel"]` proves which kernel family
served the call.  Bars: ids identical outside 2e-5 near-ties, scores within 1e-5 (inner product; squared L2 on unit rows)."""
import numpy as np
import pytest

import oracle
import synth
from lotus_amd import _capi

pytestmark = pytest.mark.gpu
F16 = _capi.PACK_F16
IP, L2 = _capi.METRIC_IP, _capi.METRIC_L2

_CORPUS = {}


def _corpus(be, nb, d):
    """(rows as stored: fp16 values in float32, packed device image) - cached: the sweep reuses a few long corpora."""
    key = (nb, d)
    if key not in _CORPUS:
        _CORPUS.clear()  # one resident corpus at a time (0.8 GB at 530 k x 768)
        xb = synth.corpus(nb, d, seed=nb % 89).astype(np.float16)
        xb[nb // 2] = xb[3]                                  # an exact duplicate far away
        xb[31], xb[32] = xb[40], xb[40]                      # ... and a pair across a block boundary
        xb[70_000:70_400] = xb[5]                            # 400 equal rows: every lane of a block holds candidates
        _CORPUS[key] = (xb.astype(np.float32), be.pack(xb, F16))
    return _CORPUS[key]


def _check(be, nq, nb, d, k, metric, expect_kernel, seed):
    xb, cb = _corpus(be, nb, d)
    xq, _ = synth.queries(xb, nq, seed=seed)
    xq = xq.astype(np.float16)
    xq[5], xq[nq - 1], xq[7 % nq] = xb[3], xb[nb - 1], xb[5]  # queries ON rows: the duplicates, the ragged tail, the equal block
    cq = be.pack(xq, F16)
    be.timing_enable(True)
    keys = be.search_keys(cb, cq, k, metric, id_offset=11)
    be.synchronize()
    t = be.timing_read_full()
    be.timing_enable(False)
    assert t["kernel"] == expect_kernel, t
    D, I = be.keys_to_result(keys, metric)
    # two queries of every 32-query block (every wave of every workgroup), the planted ones, the last
    rng = np.random.default_rng(seed)
    blocks = np.arange(0, nq, 32)
    rows = np.unique(np.clip(np.concatenate([blocks + rng.integers(0, 32, blocks.size), blocks + rng.integers(0, 32, blocks.size),
                                             [5, 7 % nq, nq - 1]]), 0, nq - 1))
    dev = be.to_device(rows.astype(np.int64))
    Dg, Ig = D[dev].cpu().numpy(), I[dev].cpu().numpy() - 11
    Dr, Ir = oracle.flat_search(xb, xq[rows].astype(np.float32), k, metric)
    err, hard, recall = synth.compare_topk(Dr, Ir, Dg, Ig, atol=1e-5)
    assert err <= 1e-5, f"score error {err}"
    assert hard == 0, f"{hard} id mismatches outside near-ties"
    assert recall >= 0.9999, recall
    # exact ties (the duplicated rows) come back in id order, as the oracle's
    q5 = int(np.nonzero(rows == 5)[0][0])
    assert np.array_equal(Ig[q5][:min(k, 2)], Ir[q5][:min(k, 2)])
    return t


# every group count of ONE launch (nq <= 4 096): d = 256 keeps the oracle's share small; nb >= 32 768 x 16 rows
@pytest.mark.parametrize("groups", list(range(1, 17)))
def test_every_group_count_of_one_launch(hip_backend, groups):
    nq = groups * 256 - (37 if groups % 2 else 0)  # (odd counts: a ragged last group)
    nq = max(nq, 129)
    metric = L2 if groups % 3 == 0 else IP
    k = (1, 10, 16, 7)[groups % 4]
    # (2 .. 8, 10, 14 .. 16 groups fill >= 224 of the 256 CUs; 9 and 11 .. 13 would idle a sixth of the chip and stay with the list
    # kernel - checked here all the same, against the same oracle)
    expect = "lvs_tile_kernel" if groups in (9, 11, 12, 13) else "lvs_rj_kernel"
    _check(hip_backend, nq, 530_003, 256, k, metric, expect, seed=100 + groups)


# chunked calls: 32 768-query chunks (128 groups x 2 ranges), 8 192 (32 x 8), 4 096 and ragged last chunks
@pytest.mark.parametrize("nq,nb,d,k,metric", [
    (9_000, 530_003, 256, 10, IP),      # 8 192 + 808 (the last chunk: four groups)
    (20_000, 140_001, 256, 10, L2),     # 8 192 + 8 192 + 3 616 on a SHORT corpus (a shard: two ranges of 2 187 blocks)
    (40_000, 140_001, 256, 16, IP),     # 32 768 + 4 096 + 3 136, k = 16 (the 16-slot lists: a shorter staging ring)
    (33_000, 70_001, 384, 1, IP),       # 32 768 + 232, k = 1, d = 384 (all B fragments in accumulation registers)
    (9_000, 140_001, 512, 10, L2),
    (9_000, 140_001, 768, 10, IP),      # the headline's operand shape
    (5_000, 530_003, 768, 10, L2),      # 4 096 + 904
])
def test_chunked_calls(hip_backend, nq, nb, d, k, metric):
    t = _check(hip_backend, nq, nb, d, k, metric, _last_kernel(nq, nb), seed=nq % 1000)
    assert t["launches"] >= 2 and t["calls"] == 1, t


def _last_kernel(nq, nb):
    """Kernel family of the call's LAST chunk (what `timing_read_full` reports): chunks of 32 768 / 8 192 / 4 096 queries, then the
    rest; a rest of <= 2 048 queries on a corpus shorter than 32 768 rows per group goes through lvs_rq_kernel."""
    left = nq
    for c in (32768, 8192, 4096):
        while left >= c and (c != 4096 or left > 4096):
            left -= c
    if left == 0:
        return "lvs_rj_kernel"
    groups = -(-left // 256)
    fits = left > 128 and (nb >= 32768 * groups and (groups == 1 or groups * (8 * (32 // groups)) >= 224) or (left > 2048 and nb >= 65536))
    return "lvs_rj_kernel" if fits else "lvs_rq_kernel"


def test_a_corpus_of_whole_blocks_and_one_with_a_single_tail_row(hip_backend):
    """nb % 32 == 0 (no tail launch) and nb % 32 == 1 (a one-row tail through lvs_rq_kernel): the last row wins for the query
    planted on it."""
    for nb in (65_536 + 32 * 7, 65_536 + 32 * 7 + 1):
        be = hip_backend
        xb = synth.corpus(nb, 256, seed=3).astype(np.float16)
        xq, _ = synth.queries(xb.astype(np.float32), 300, seed=4)
        xq = xq.astype(np.float16)
        xq[299] = xb[nb - 1]
        keys = be.search_keys(be.pack(xb, F16), be.pack(xq, F16), 5, IP)
        D, I = be.keys_to_result(keys, IP)
        Dr, Ir = oracle.flat_search(xb.astype(np.float32), xq.astype(np.float32), 5, IP)
        err, hard, recall = synth.compare_topk(Dr, Ir, D.cpu().numpy(), I.cpu().numpy(), atol=1e-5)
        assert err <= 1e-5 and hard == 0 and recall >= 0.9999
        assert int(I[299, 0]) == nb - 1


@pytest.mark.parametrize("metric", [IP, L2])
def test_fp32_embeddings_take_the_same_kernel_for_their_one_certified_pass(hip_backend, metric):
    """LOTUS's default embeddings are float32 (hi|lo rows here).  A join of 4 097 queries or more runs its ONE pass over the hi parts
    on lvs_rj_kernel (plain 15-deep lists instead of the list kernel's banded ones), rescoring and the certificate as before:
    the exact top k of the plain multi-segment search, and the float32 oracle's ids / scores."""
    SPLIT = _capi.PACK_SPLIT
    be = hip_backend
    nq, nb, d, k = 9_000, 140_001, 256, 10
    xb = synth.corpus(nb, d, seed=17) * np.float32(1.0 + 2.0 ** -12)  # (values that need their lo halves)
    xq, _ = synth.queries(xb, nq, seed=18)
    xb[nb // 3] = xb[9]
    xq[4], xq[nq - 1] = xb[9], xb[nb - 1]
    cb = be.pack(xb, SPLIT, exp="auto")
    cq = be.pack(xq, SPLIT, exp=cb.exp)  # (squared L2 needs one scale on both operands)
    stats = {}
    be.timing_enable(True)
    keys = be.search_keys(cb, cq, k, metric, stats=stats)
    be.synchronize()
    t = be.timing_read_full()
    be.timing_enable(False)
    assert t["kernel"] in ("lvs_rj_kernel", "lvs_rq_kernel", "lvs_tile_kernel"), t  # (the last timed launch may be the open queries' second round)
    assert t["launches"] >= 2, t                                                    # ... but the first pass ran in chunks
    plain = be.search_keys(cb, cq, k, metric, one_pass=False)
    sexp = be.score_exp_of(cb, cq)
    Dc, Ic = be.keys_to_result(keys, metric, score_exp=sexp)
    Dp, Ip = be.keys_to_result(plain, metric, score_exp=sexp)
    same = (Ic == Ip).float().mean().item()
    assert same >= 0.9995, same  # (exact scores from two summation orders: only near-ties may swap)
    assert stats["queries"] == nq and stats["uncertified"] <= 0.02 * nq, stats
    rows = np.unique(np.concatenate([np.arange(0, nq, 97), [4, nq - 1]]))
    Dr, Ir = oracle.flat_search(xb, xq[rows], k, metric)
    dev = be.to_device(rows.astype(np.int64))
    err, hard, recall = synth.compare_topk(Dr, Ir, Dc[dev].cpu().numpy(), Ic[dev].cpu().numpy(), atol=1e-5)
    assert err <= 1e-5 and hard == 0 and recall >= 0.9999


def test_threshold_self_join_on_the_same_kernel_equals_brute_force(hip_backend, monkeypatch):
    """sem_dedup's arithmetic (`sem_dedup.py:45-46`: all pairs with score > threshold, strict) at a size where `lvs_range_join` runs
    on lvs_rj_kernel's RANGE epilogue: chunks of 32 768 query rows, each against the rows above its first query (j > i), the
    corpus' ragged last rows and the short last chunks through the list kernel - one pair list, the planted pairs exactly."""
    be = hip_backend
    n, d = 150_011, 256
    rng = np.random.default_rng(5)
    x = synth.corpus(n, d, seed=41)
    src = rng.choice(n, 3000, replace=False)
    dst = rng.choice(np.setdiff1d(np.arange(n), src), 3000, replace=False)
    x[dst] = x[src] + 0.05 * synth.corpus(3000, d, seed=42)            # near-duplicates (cos ~ 0.998), spread over every chunk
    hard = rng.choice(np.setdiff1d(np.arange(n), np.concatenate([src, dst])), 2000, replace=False)
    x[hard] = x[(hard + 7) % n] + 0.5 * synth.corpus(2000, d, seed=43)  # hard negatives (cos ~ 0.89)
    x[n - 1] = x[n - 40_000]                                             # an exact duplicate in the ragged tail
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x16 = x.astype(np.float16)
    p = be.pack(x16, F16)
    be.timing_enable(True)
    q, j, sc = be.range_join(p, p, 0.95, IP, q_row0=0)
    be.synchronize()
    t = be.timing_read_full()
    be.timing_enable(False)
    assert t["launches"] >= 5, t  # chunks of the register-resident kernel + the list kernel's tails
    got = set(zip(q.cpu().numpy().tolist(), j.cpu().numpy().tolist()))
    assert len(got) == int(q.numel())  # no pair twice
    # brute force over the candidate rows only (every planted row against everything) - float32 on the stored fp16 values
    s = x16.astype(np.float32)
    cand = np.unique(np.concatenate([src, dst, hard, (hard + 7) % n, [n - 1, n - 40_000]]))
    sims = s[cand] @ s.T
    want, near = set(), set()
    for a_i, row in zip(cand.tolist(), sims):
        for b_i in np.nonzero(row > 0.95 - 2e-5)[0].tolist():
            if b_i == a_i:
                continue
            pair = (min(a_i, b_i), max(a_i, b_i))
            (want if row[b_i] > 0.95 + 2e-5 else near).add(pair)
    assert want <= got, sorted(want - got)[:5]
    assert got <= (want | near), sorted(got - want - near)[:5]
    assert len(want) >= 3000
    qi, ji = q.cpu().numpy(), j.cpu().numpy()
    assert np.allclose(sc.cpu().numpy(), np.einsum("ij,ij->i", s[qi], s[ji]), atol=1e-5)
    assert (ji > qi).all()
    # several ranks: whole chunks of query rows are dealt to them (snake order) - three "ranks" partition the result
    parts = [be.range_join(p, p, 0.95, IP, q_row0=0, stride=3, phase=r) for r in range(3)]
    sets = [set(zip(a.cpu().numpy().tolist(), b.cpu().numpy().tolist())) for a, b, _ in parts]
    assert all(sets) and not (sets[0] & sets[1]) and not (sets[0] & sets[2]) and not (sets[1] & sets[2])
    assert (sets[0] | sets[1] | sets[2]) == got
