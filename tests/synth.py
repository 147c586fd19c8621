"""Synthetic embeddings shaped like BASELINE.md section 2: unit-norm Gaussian corpus, queries with a planted
neighbour at cos ~ 0.71."""
import numpy as np


def corpus(n, d, seed=0, dtype=np.float32):
    rng = np.random.default_rng(np.random.SeedSequence([20260923, seed, n, d]))
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(dtype)


def queries(xb, nq, seed=1, dtype=np.float32):
    n, d = xb.shape
    rng = np.random.default_rng(np.random.SeedSequence([20260923, seed, nq, d, 7]))
    j = rng.integers(0, n, nq)
    u = rng.standard_normal((nq, d), dtype=np.float32)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    q = 0.7 * xb[j].astype(np.float32) + 0.7 * u
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(dtype), j


def compare_topk(D_ref, I_ref, D, I, atol=1e-5, tie_gap=2e-5):
    """Parity protocol of SURVEY.md section 8(c): scores within atol after aligning by rank; ids identical
    wherever the reference's neighbouring scores are further apart than tie_gap.  Returns (max score error,
    number of id mismatches outside near-tie groups, recall)."""
    D_ref, D = np.asarray(D_ref, np.float64), np.asarray(D, np.float64)
    I_ref, I = np.asarray(I_ref), np.asarray(I)
    valid = I_ref >= 0
    err = np.abs(np.where(valid, D_ref - D, 0.0)).max() if valid.any() else 0.0
    mism = (I_ref != I)
    hard = 0
    nq, k = I_ref.shape
    for q, r in zip(*np.nonzero(mism)):
        gaps = []
        if r > 0:
            gaps.append(abs(D_ref[q, r] - D_ref[q, r - 1]))
        if r + 1 < k:
            gaps.append(abs(D_ref[q, r] - D_ref[q, r + 1]))
        else:
            gaps.append(0.0)  # boundary: the (k+1)-th score is unknown here, treat as possible tie
        if min(gaps) > tie_gap:
            hard += 1
    inter = sum(len(set(a[a >= 0]) & set(b[b >= 0])) for a, b in zip(I_ref, I))
    denom = max(1, int(valid.sum()))
    return float(err), int(hard), inter / denom
