/* CPU oracle, plain C restatement.  TEST INFRASTRUCTURE ONLY - never linked into the product library.
 *
 * Restates what the reference delegates to faiss-cpu 1.13.0 (reference uv.lock:573-574) at the call sites
 *   lotus/vector_store/faiss_vs.py:23-24,63-64  index_factory(d,"Flat",metric) + add
 *   lotus/vector_store/faiss_vs.py:67,75         search(query_vectors, K)
 *   lotus/utils.py:61-62,65                      Kmeans(d,k,niter).train(x); index.search(x,1)
 * following the published faiss behaviour summarised in SURVEY.md Appendix A (A.2-A.4).
 * PARITY UNPINNED: faiss itself is not available in this image; see oracle/__init__.py.
 *
 * Total order used everywhere: (score best-first, id ascending), expressed as one uint64 key
 *   key = ord32(score_where_larger_is_better) << 32 | (0xFFFFFFFF - id),   key 0 = empty slot.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint32_t ord32(float f) {
    f = f + 0.0f; /* fold -0 onto +0 */
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u >> 31) ? ~u : (u ^ 0x80000000u);
}

static inline uint64_t pack_key(float better, uint32_t id) {
    return ((uint64_t)ord32(better) << 32) | (uint64_t)(0xFFFFFFFFu - id);
}

/* keys[0..k) sorted descending; insert cand if it beats the last one (faiss HeapBlockResultHandler role,
 * Appendix A.3: "insert only if strictly better than the current k-th"). */
static inline void insert_sorted(uint64_t* keys, int k, uint64_t cand) {
    if (cand <= keys[k - 1]) return;
    int j = k - 1;
    while (j > 0 && keys[j - 1] < cand) {
        keys[j] = keys[j - 1];
        --j;
    }
    keys[j] = cand;
}

/* Feed one score block (nq x nb, larger = better, leading dimension ld) whose columns are ids id0..id0+nb-1
 * into the per-query sorted key lists keys[nq][k].  This is the "result handler" step of faiss's blocked
 * BLAS search (one call per 4096 x 1024 sgemm block). */
void oracle_topk_update(const float* better, int64_t nq, int64_t nb, int64_t ld, int64_t id0, int32_t k,
                        uint64_t* keys) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nq; ++q) {
        const float* row = better + q * ld;
        uint64_t* kq = keys + q * (int64_t)k;
        uint32_t thr_ord = (uint32_t)(kq[k - 1] >> 32);
        for (int64_t j = 0; j < nb; ++j) {
            uint32_t o = ord32(row[j]);
            if (o < thr_ord) continue; /* cheap reject on the score alone */
            insert_sorted(kq, k, ((uint64_t)o << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(id0 + j)));
            thr_ord = (uint32_t)(kq[k - 1] >> 32);
        }
    }
}

/* Whole Flat search in plain C: per-pair dot products (faiss's non-BLAS path, Appendix A.3) + k-best lists.
 * metric 0 = inner product (descending), 1 = squared L2 (ascending, direct sum (x-y)^2).
 * Output keys[nq][k] (0 = empty); the Python side decodes them exactly like the numpy oracle. */
void oracle_flat_search_naive(const float* xb, int64_t nb, const float* xq, int64_t nq, int32_t d, int32_t k,
                              int32_t metric, uint64_t* keys) {
    memset(keys, 0, sizeof(uint64_t) * (size_t)nq * (size_t)k);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t q = 0; q < nq; ++q) {
        const float* x = xq + q * (int64_t)d;
        uint64_t* kq = keys + q * (int64_t)k;
        for (int64_t j = 0; j < nb; ++j) {
            const float* y = xb + j * (int64_t)d;
            float acc = 0.f;
            if (metric == 0) {
                for (int32_t t = 0; t < d; ++t) acc += x[t] * y[t];
            } else {
                for (int32_t t = 0; t < d; ++t) {
                    float df = x[t] - y[t];
                    acc += df * df;
                }
                acc = -acc;
            }
            insert_sorted(kq, k, pack_key(acc, (uint32_t)j));
        }
    }
}

/* ---- std::mt19937 (the generator behind faiss RandomGenerator, Appendix A.4) ---- */
typedef struct {
    uint32_t mt[624];
    int idx;
} mt19937_t;

static void mt_seed(mt19937_t* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937_t* g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            g->mt[i] = v;
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

void oracle_mt19937_raw(uint32_t seed, int64_t n, uint32_t* out) {
    mt19937_t g;
    mt_seed(&g, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = mt_next(&g);
}

/* faiss rand_perm: Fisher-Yates with rand_int(max) = mt() % max (Appendix A.4). */
void oracle_rand_perm(int64_t n, int64_t seed, int64_t* perm) {
    mt19937_t g;
    mt_seed(&g, (uint32_t)seed);
    for (int64_t i = 0; i < n; ++i) perm[i] = i;
    for (int64_t i = 0; i + 1 < n; ++i) {
        int64_t i2 = i + (int64_t)(mt_next(&g) % (uint32_t)(n - i));
        int64_t t = perm[i];
        perm[i] = perm[i2];
        perm[i2] = t;
    }
}

/* faiss compute_centroids (Clustering.cpp): per-centroid float32 sums accumulated in point order, then
 * scaled by 1/count; centroids of empty clusters are left as they were.  hassign[k] receives the counts. */
void oracle_compute_centroids(const float* x, int64_t n, int32_t d, int32_t k, const int64_t* assign,
                              float* centroids, float* hassign) {
    float* sums = (float*)calloc((size_t)k * (size_t)d, sizeof(float));
    for (int32_t c = 0; c < k; ++c) hassign[c] = 0.f;
    for (int64_t i = 0; i < n; ++i) {
        int64_t c = assign[i];
        hassign[c] += 1.0f;
        float* s = sums + c * (int64_t)d;
        const float* xi = x + i * (int64_t)d;
        for (int32_t t = 0; t < d; ++t) s[t] += xi[t];
    }
    for (int32_t c = 0; c < k; ++c) {
        if (hassign[c] == 0.f) continue;
        float norm = 1.0f / hassign[c];
        float* cc = centroids + (int64_t)c * d;
        const float* s = sums + (int64_t)c * d;
        for (int32_t t = 0; t < d; ++t) cc[t] = s[t] * norm;
    }
    free(sums);
}

/* faiss split_clusters (Clustering.cpp): re-seed every empty cluster from a populated one chosen with
 * probability (count-1)/(n-k), RNG seeded 1234, symmetric (1 +- 1/1024) perturbation.  Returns #splits. */
int32_t oracle_split_clusters(int32_t d, int32_t k, int64_t n, float* hassign, float* centroids) {
    const float EPS = 1.0f / 1024.0f;
    mt19937_t g;
    mt_seed(&g, 1234u);
    int32_t nsplit = 0;
    for (int32_t ci = 0; ci < k; ++ci) {
        if (hassign[ci] != 0.f) continue;
        int32_t cj;
        for (cj = 0;; cj = (cj + 1) % k) {
            float p = (hassign[cj] - 1.0f) / (float)(n - k);
            float r = (float)mt_next(&g) / (float)4294967295u; /* rand_float = mt() / float(mt.max()) */
            if (r < p) break;
        }
        float* a = centroids + (int64_t)ci * d;
        float* b = centroids + (int64_t)cj * d;
        memcpy(a, b, sizeof(float) * (size_t)d);
        for (int32_t j = 0; j < d; ++j) {
            if (j % 2 == 0) {
                a[j] *= 1 + EPS;
                b[j] *= 1 - EPS;
            } else {
                a[j] *= 1 - EPS;
                b[j] *= 1 + EPS;
            }
        }
        hassign[ci] = hassign[cj] / 2;
        hassign[cj] -= hassign[ci];
        ++nsplit;
    }
    return nsplit;
}

int32_t oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
