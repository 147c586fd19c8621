"""Test double for ``lotus_amd.backend.HipBackend`` that answers every device call with the CPU oracle.

Lives under tests/ on purpose: it lets the host-side logic of ``HipVS`` (argument handling, padding, subset/ids
mapping, caching, sharding + all-gather + merge over gloo) run in a GPU-less container.  It is never importable
from the product package."""
from __future__ import annotations

import numpy as np
import torch

import oracle
from lotus_amd import _capi
from lotus_amd.backend import PackedRows
from oracle.flat import _finish


def _emulate_storage(x: np.ndarray, mode: int) -> np.ndarray:
    """Values the device would hold: fp16-rounded (PACK_F16) or hi+lo of an fp16 pair (PACK_SPLIT)."""
    x = np.asarray(x, dtype=np.float32)
    hi = x.astype(np.float16)
    if mode == _capi.PACK_F16:
        return hi.astype(np.float32)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32) + lo.astype(np.float32)


class OracleBackend:
    device = torch.device("cpu")

    def __init__(self):
        self.calls = []

    def synchronize(self):
        pass

    def to_device(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr))

    SCALE_TARGET_EXP = 6

    @staticmethod
    def absmax(x) -> float:
        x = x.numpy() if torch.is_tensor(x) else np.asarray(x)
        return float(np.abs(x.astype(np.float32)).max(initial=0.0))

    @classmethod
    def exp_for(cls, absmax: float) -> int:
        if not np.isfinite(absmax) or absmax <= 0.0:
            return 0
        return int(np.clip(cls.SCALE_TARGET_EXP - int(np.floor(np.log2(absmax))), -60, 60))

    def pack(self, x, mode, normalize=False, check=False, exp=0):
        """As the device: the image holds x * 2^exp ("auto": chosen from the data for hi|lo rows), flags report inf / NaN
        and magnitudes that leave fp16's range under that scale."""
        x = x.numpy() if torch.is_tensor(x) else np.asarray(x)
        x = x.astype(np.float32)
        if exp == "auto":
            if mode != _capi.PACK_SPLIT or x.shape[0] == 0:
                exp = 0
            elif normalize:
                exp = self.SCALE_TARGET_EXP + 2
            else:
                finite = x[np.isfinite(x)]
                exp = self.exp_for(self.absmax(finite)) if finite.size else 0
        exp = int(exp)
        f = 0
        if not np.isfinite(x).all():
            f |= _capi.PACK_FLAG_NONFINITE
        with np.errstate(all="ignore"):
            if normalize:
                x = x / np.linalg.norm(x, axis=1, keepdims=True)
            x = x * np.float32(2.0 ** exp)
            if np.abs(np.where(np.isfinite(x), x, 0)).max(initial=0.0) > 65504.0:
                f |= _capi.PACK_FLAG_RANGE
            if check is True and f:
                self.raise_for_flags(f)
            vals = _emulate_storage(x, mode)
        self.calls.append(("pack", vals.shape, mode))
        norms = np.einsum("ij,ij->i", vals, vals, dtype=np.float32)
        return PackedRows(rows=torch.from_numpy(vals), norms=torch.from_numpy(norms), n=vals.shape[0],
                          d=vals.shape[1] if vals.ndim == 2 else 0, mode=mode, exp=exp,
                          flags=torch.tensor([f], dtype=torch.int32) if check == "lazy" else None)

    @staticmethod
    def raise_for_flags(f, what="embeddings"):
        from lotus_amd.backend import HipBackend

        HipBackend.raise_for_flags(f, what)

    def gather(self, src, ids_dev):
        idx = ids_dev.numpy().astype(np.int64)
        self.calls.append(("gather", len(idx)))
        return PackedRows(rows=src.rows[idx], norms=src.norms[idx], n=len(idx), d=src.d, mode=src.mode, exp=src.exp)

    SEED_EXCHANGE_MIN_QUERIES = 2048

    def seed_tiles(self, nq, shard_rows, k, corpus_mode, query_mode):
        """Same rule as HipBackend.seed_tiles (the tile count comes from the library's host-side planner)."""
        if nq < self.SEED_EXCHANGE_MIN_QUERIES or corpus_mode != _capi.PACK_F16 or query_mode != _capi.PACK_F16:
            return 0
        if not 1 <= k <= 56:
            return 0
        return max(0, int(_capi.load().lvs_flat_search_seed_tiles(int(nq), int(shard_rows), int(k))))

    def seed_scores(self, corpus, queries, metric, tiles):
        """[tiles, nq]: best score of every query over each of the first `tiles` 256-row tiles (-inf past the shard)."""
        self.calls.append(("seed", queries.n, corpus.n, tiles))
        out = np.full((tiles, queries.n), -np.inf, np.float32)
        xb, xq = corpus.rows.numpy(), queries.rows.numpy()
        for t in range(min(tiles, corpus.n // 256)):
            s = xq @ xb[256 * t:256 * t + 256].T
            if metric == 1:
                s = -np.maximum((queries.norms.numpy()[:, None] + corpus.norms.numpy()[None, 256 * t:256 * t + 256]) - 2 * s, 0)
            out[t] = s.max(axis=1)
        return torch.from_numpy(out)

    def search_keys(self, corpus, queries, k, metric, id_offset=0, row_ids=None, one_pass=None, stats=None, seed_scores=None):
        self.calls.append(("search", queries.n, corpus.n, k, metric))
        xb, xq = corpus.rows.numpy(), queries.rows.numpy()
        if corpus.n == 0:
            return torch.zeros((queries.n, k), dtype=torch.int64)
        D, I = oracle.flat_search(xb, xq, min(k, corpus.n), metric)
        better = D if metric == 0 else -D
        valid = I >= 0
        if seed_scores is not None and seed_scores.shape[0] >= k:
            # as the device: rows below the k-th largest pooled sample score are never listed (a shard may come back short);
            # a float32 ulp of slack - the double's sample scores and search scores come from different sgemm shapes
            thr = np.sort(seed_scores.numpy(), axis=0)[::-1][k - 1]
            valid &= better >= (thr - 1e-6 * np.abs(thr))[:, None]
            self.calls.append(("seeded", int((~valid).sum())))
        if row_ids is not None:
            rid = row_ids.numpy().view(np.uint32).astype(np.int64)
            ids = np.where(valid, rid[np.where(valid, I, 0)], 0)
        else:
            ids = np.where(valid, I + id_offset, 0)
        keys = np.where(valid, oracle.pack_keys(better, ids), np.uint64(0))
        # a mapped id order may differ from the scan order inside exact ties: re-sort by key
        keys = np.sort(keys, axis=1)[:, ::-1]
        if keys.shape[1] < k:
            keys = np.concatenate([keys, np.zeros((keys.shape[0], k - keys.shape[1]), np.uint64)], axis=1)
        return torch.from_numpy(np.array(keys, dtype=np.uint64, order="C", copy=True).view(np.int64))

    def nearest(self, corpus, queries, metric, id_offset=0, stats=None, exact_scores=True, corpus_stats=None):
        return self.search_keys(corpus, queries, 1, metric, id_offset=id_offset)

    def merge_keys(self, parts):
        p = parts.numpy().view(np.uint64)
        P, nq, k = p.shape
        allk = np.transpose(p, (1, 0, 2)).reshape(nq, P * k)
        out = np.sort(allk, axis=1)[:, ::-1][:, :k]
        self.calls.append(("merge", P, nq, k))
        return torch.from_numpy(np.array(out, dtype=np.uint64, order="C", copy=True).view(np.int64))

    @staticmethod
    def score_exp_of(corpus, queries):
        return int(corpus.exp) + int(queries.exp)

    def keys_to_result(self, keys, metric, id_map=None, score_exp=0):
        k = keys.numpy().view(np.uint64)
        D, I = _finish(k, metric, None if id_map is None else id_map.numpy())
        if score_exp:
            D = np.where(I >= 0, D * np.float32(2.0 ** -score_exp), D).astype(np.float32)  # exact; pads stay +-FLT_MAX
        return torch.from_numpy(D), torch.from_numpy(I)

    def scores(self, corpus, queries, metric):
        xb, xq = corpus.rows.numpy(), queries.rows.numpy()
        if metric == 1 and corpus.exp != queries.exp:
            raise ValueError("squared L2 needs both operands packed with the same scale exponent")
        s = xq @ xb.T
        if metric == 1:
            s = -np.maximum((queries.norms.numpy()[:, None] + corpus.norms.numpy()[None, :]) - 2 * s, 0)
        s = s.astype(np.float32) * np.float32(2.0 ** -self.score_exp_of(corpus, queries))
        return torch.from_numpy(s.astype(np.float32))

    def rank_all(self, corpus, queries, metric, id_offset=0):
        return self.search_keys(corpus, queries, corpus.n, metric, id_offset=id_offset)

    def rank_scores(self, sc, id_offset=0):
        s = sc.numpy().astype(np.float32)
        self.calls.append(("rank", s.shape[0], s.shape[1]))
        ids = np.broadcast_to(np.arange(s.shape[1], dtype=np.int64) + id_offset, s.shape)
        keys = np.sort(oracle.pack_keys(s, ids), axis=1)[:, ::-1]
        return torch.from_numpy(np.array(keys, dtype=np.uint64, order="C", copy=True).view(np.int64))

    def unpack(self, src, ids_dev=None, raw=False):
        rows = src.rows if ids_dev is None else src.rows[ids_dev.numpy().astype(np.int64)]
        rows = rows.clone().to(torch.float32)
        return rows if (raw or not src.exp) else rows * float(2.0 ** -src.exp)

    @staticmethod
    def slice_rows(src, r0, r1):
        return PackedRows(rows=src.rows[r0:r1], norms=src.norms[r0:r1], n=r1 - r0, d=src.d, mode=src.mode, exp=src.exp)

    def kmeans_update_centroids(self, sums, counts, centroids):
        c, sm, cn = centroids.numpy(), sums.numpy(), counts.numpy()
        nz = cn > 0
        c[nz] = sm[nz] * (np.float32(1.0) / cn[nz])[:, None]  # faiss: c *= 1 / count

    # ---- threshold join ----
    def range_join(self, corpus, queries, threshold, metric=0, q_row0=-1, id_offset=0, stride=1, phase=0,
                   capacity=1 << 22):
        xb, xq = corpus.rows.numpy(), queries.rows.numpy()
        s = xq @ xb.T
        if metric == 1:
            s = -np.maximum((queries.norms.numpy()[:, None] + corpus.norms.numpy()[None, :]) - 2 * s, 0)
        s = (s.astype(np.float32) * np.float32(2.0 ** -self.score_exp_of(corpus, queries))).astype(np.float32)
        q, j = np.nonzero(s > np.float32(threshold))
        keep = ((q // 256) % stride) == phase  # 256-query tiles, as the kernel deals them
        if q_row0 >= 0:
            keep &= (j + id_offset) > (q + q_row0)
        q, j = q[keep], j[keep]
        return (torch.from_numpy(q.astype(np.int64)), torch.from_numpy((j + id_offset).astype(np.int64)),
                torch.from_numpy(s[q, j].astype(np.float32)))

    # ---- k-means pieces ----
    def kmeans_accumulate(self, x, assign, k):
        vals = x.rows.numpy()
        a = assign.numpy().astype(np.int64)
        sums = np.zeros((k, vals.shape[1]), np.float32)
        ok = (a >= 0) & (a < k)
        np.add.at(sums, a[ok], vals[ok])  # unbuffered: in row order, like the device kernel
        counts = np.bincount(a[ok], minlength=k).astype(np.float32)
        return torch.from_numpy(sums), torch.from_numpy(counts)

    def kmeans_accumulate_keys(self, x, keys, k, id_offset=0):
        kk = keys.numpy().view(np.uint64).reshape(-1)
        ids = (np.uint64(0xFFFFFFFF) - (kk & np.uint64(0xFFFFFFFF))).astype(np.int64) - id_offset
        ids[kk == 0] = -1
        return self.kmeans_accumulate(x, torch.from_numpy(ids), k)

    def kmeans_objective(self, centroids, sums, counts, x2, out):
        c = centroids.numpy().astype(np.float64)
        o = x2.numpy()[0] - 2.0 * (c * sums.numpy().astype(np.float64)).sum() + (counts.numpy().astype(np.float64) * (c * c).sum(1)).sum()
        out[0] = float(o)

    def kmeans_pack_centroids(self, centroids, mode, exp=0):
        pk = self.pack(centroids, mode)  # centroids live in the points' scaled domain already
        pk.exp = int(exp)
        return pk, torch.zeros(2)

    def kmeans_finish(self, sums, counts, centroids, n_train, mode, nsplit_out=None, exp=0):
        self.kmeans_update_centroids(sums, counts, centroids)
        hs = counts.numpy()
        ns = 0
        if n_train > 0 and (hs == 0).any():
            ns = self.split_clusters(n_train, hs, centroids.numpy())
        if nsplit_out is not None:
            nsplit_out[0] = ns
        return self.kmeans_pack_centroids(centroids, mode, exp)

    def rand_perm(self, n, seed, m=None):
        perm = oracle.rand_perm(n, seed)
        return perm if m is None else perm[:m]

    def split_clusters(self, n, hassign, centroids):
        from oracle.kmeans import _split_clusters

        return _split_clusters(n, hassign, centroids, None)
