"""Regenerates tests/golden/*.npz.  Run from the repo root: `python tests/golden/make_golden.py`.

The reference holds no golden vectors for this path and its arithmetic (faiss-cpu 1.13.0) cannot run here
("parity unpinned", oracle/__init__.py), so these fixtures are produced by the CPU oracle from seeded inputs and
pin (i) the oracle against regressions and (ii) the HIP path on the GPU box, where neither /root/reference nor faiss
exist.  Inputs are stored as float16 so that fixtures stay small and the device sees bit-identical values."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
import synth  # noqa: E402


def flat_case(name, nq, nb, d, k, metric, seed, dup=False, ids=None, scale=1.0):
    xb = (synth.corpus(nb, d, seed=seed) * scale).astype(np.float16)
    if dup:
        xb[nb // 2:] = xb[: nb - nb // 2]  # exact duplicate rows -> exact score ties
    xq = synth.queries(xb.astype(np.float32), nq, seed=seed + 1)[0].astype(np.float16)
    idl = None
    if ids:
        idl = np.sort(np.random.default_rng(seed).choice(nb, ids, replace=False)).astype(np.int64)
    D, I = oracle.flat_search(xb.astype(np.float32), xq.astype(np.float32), k, metric, ids=idl)
    np.savez_compressed(os.path.join(HERE, f"flat_{name}.npz"), xb=xb, xq=xq, k=k, metric=metric,
                        ids=np.zeros(0, np.int64) if idl is None else idl, D=D, I=I)
    print(name, D.shape)


def main():
    flat_case("ip_small", 64, 500, 96, 5, 0, 1)
    flat_case("ip_ragged", 33, 777, 100, 10, 0, 2)
    flat_case("ip_dups", 40, 300, 64, 12, 0, 3, dup=True)
    flat_case("ip_k_gt_n", 7, 9, 32, 16, 0, 4)
    flat_case("ip_subset", 50, 900, 64, 8, 0, 5, ids=200)
    flat_case("l2_small", 64, 500, 96, 5, 1, 6, scale=1.7)
    flat_case("ip_k30", 20, 400, 64, 30, 0, 7)
    # k-means (faiss-parity mode): blobs; fp16-representable inputs
    rng = np.random.default_rng(11)
    k, d, n = 8, 32, 3000
    c = rng.standard_normal((k, d)).astype(np.float32) * 3
    x = (c[rng.integers(0, k, n)] + 0.4 * rng.standard_normal((n, d))).astype(np.float16)
    r = oracle.kmeans_faiss(x.astype(np.float32), k, niter=8, max_points_per_centroid=128)
    np.savez_compressed(os.path.join(HERE, "kmeans_blobs.npz"), x=x, k=k, niter=8, mppc=128, centroids=r.centroids,
                        assign=r.assign, obj=r.obj, train_ids=r.train_ids)
    # dedup: planted near-duplicates incl. a chain
    base = synth.corpus(120, 48, seed=9)
    u = synth.corpus(120, 48, seed=10)
    near = base[:30] + 0.15 * u[:30]
    chain = near[:10] + 0.15 * u[30:40]
    far = base[30:50] + 0.8 * u[40:60]
    xd = np.concatenate([base, near, chain, far])
    xd /= np.linalg.norm(xd, axis=1, keepdims=True)
    xd = xd.astype(np.float16)
    pi, pj, ps = oracle.range_self_join(xd.astype(np.float32), 0.95)
    labels = oracle.dedup_components(len(xd), pi, pj)
    np.savez_compressed(os.path.join(HERE, "dedup_pairs.npz"), x=xd, thr=0.95, pi=pi, pj=pj, ps=ps, labels=labels)
    print("pairs", len(pi), "components", len(set(labels.tolist())))


if __name__ == "__main__":
    main()
