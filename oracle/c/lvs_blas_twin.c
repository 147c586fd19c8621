/* CPU timing comparator in C + OpenMP: faiss's BLAS search path (query blocks x database blocks, one sgemm per block
 * pair, block scores fed to a k-best collector) written out as ONE fused loop nest - a packed,
 * register-blocked sgemm micro-kernel whose 12 x 32 score tile goes straight into the per-query k-best lists, so the
 * score matrix is never materialised.  TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py cpu_baseline; checked against
 * oracle/flat.py in tests/test_oracle.py).  Labelled "port", never "faiss": faiss-cpu is not installable here.
 *
 * Reference call sites this stands in for: lotus/vector_store/faiss_vs.py:67,75 (index.search on >= 20 queries).
 *
 * Parallelisation: every thread owns a contiguous slice of database rows and scans ALL queries against it (its own
 * k-best list per query), then the per-thread lists are merged per query - the same "each worker emits local top-k,
 * merge" shape as the multi-GPU path.  Vector code uses GCC vector extensions (64-byte vectors -> AVX-512 with
 * -march=x86-64-v4, which both this container's Xeon and the GPU box's EPYC 9575F execute).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float v16f __attribute__((vector_size(64)));

#define MR 12  /* queries per micro-tile   */
#define NR 32  /* database rows per strip  */
#define NB 256 /* database rows per packed block (NB x d floats: 768 KB at d = 768, L2-resident) */

static inline uint32_t t_ord32(float f) {
    f = f + 0.0f;
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u >> 31) ? ~u : (u ^ 0x80000000u);
}
static inline float t_unord32(uint32_t o) {
    uint32_t u = (o >> 31) ? (o ^ 0x80000000u) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline void t_insert(uint64_t* keys, int k, uint64_t cand) {
    if (cand <= keys[k - 1]) return;
    int j = k - 1;
    while (j > 0 && keys[j - 1] < cand) {
        keys[j] = keys[j - 1];
        --j;
    }
    keys[j] = cand;
}

/* C[MR][NR] = A[MR][0..d) . Bp[0..d)[NR]   (A: MR query rows, row stride lda; Bp: one packed NR-wide panel, read
 * sequentially: 128 B per k) */
static inline void micro(const float* A, int64_t lda, int mr, const float* Bp, int d, float* C) {
    v16f acc[MR][2];
    for (int i = 0; i < MR; ++i) acc[i][0] = acc[i][1] = (v16f){0};
    const float* a[MR];
    for (int i = 0; i < MR; ++i) a[i] = A + (int64_t)(i < mr ? i : mr - 1) * lda; /* short tiles repeat the last row */
    for (int kk = 0; kk < d; ++kk) {
        const v16f b0 = *(const v16f*)(Bp + (int64_t)kk * NR);
        const v16f b1 = *(const v16f*)(Bp + (int64_t)kk * NR + 16);
#pragma GCC unroll 12
        for (int i = 0; i < MR; ++i) {
            const float s = a[i][kk];
            const v16f av = {s, s, s, s, s, s, s, s, s, s, s, s, s, s, s, s};
            acc[i][0] += av * b0;
            acc[i][1] += av * b1;
        }
    }
    for (int i = 0; i < MR; ++i) {
        *(v16f*)(C + i * NR) = acc[i][0];
        *(v16f*)(C + i * NR + 16) = acc[i][1];
    }
}

/* metric 0: inner product (larger = better); 1: squared L2 via |x|^2 + |y|^2 - 2<x,y> clamped at 0 (better = -dist).
 * out_keys [nq][k] uint64 result keys (same encoding as the oracle), best first. */
void twin_flat_search(const float* xb, int64_t nb, const float* xq, int64_t nq, int32_t d, int32_t k, int32_t metric,
                      uint64_t* out_keys, int32_t nthreads) {
    if (nq <= 0 || k <= 0) return;
    memset(out_keys, 0, (size_t)nq * k * 8);
    if (nb <= 0) return;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    if ((int64_t)nthreads * NB > nb) nthreads = (int)((nb + NB - 1) / NB);
    uint64_t* lists = (uint64_t*)calloc((size_t)nthreads * nq * k, 8); /* per-thread k-best per query */
    float* qn = NULL;
    if (metric == 1) {
        qn = (float*)malloc((size_t)nq * 4);
        for (int64_t q = 0; q < nq; ++q) {
            float s = 0.f;
            for (int j = 0; j < d; ++j) s += xq[q * d + j] * xq[q * d + j];
            qn[q] = s;
        }
    }
#pragma omp parallel num_threads(nthreads)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t per = ((nb + nthreads - 1) / nthreads + NB - 1) / NB * NB;
        const int64_t r_lo = (int64_t)t * per, r_hi = r_lo + per < nb ? r_lo + per : nb;
        uint64_t* mine = lists + (size_t)t * nq * k;
        /* the k-th best score of every query, densely packed: the per-tile threshold test reads 12 adjacent floats instead
         * of one cache line per query out of the 80-byte-strided lists (8 192 queries x k = 10: 655 KB of lists would share
         * the 1 MB L2 with the 768 KB packed block; 32 KB of thresholds do not) */
        float* thrv = (float*)malloc((size_t)nq * 4);
        for (int64_t q = 0; q < nq; ++q) thrv[q] = -__builtin_inff();
        float* Bt = (float*)aligned_alloc(64, (size_t)d * NB * 4);
        float bnorm[NB];
        float C[MR * NR] __attribute__((aligned(64)));
        for (int64_t r0 = r_lo; r0 < r_hi; r0 += NB) {
            const int rows = (int)(r_hi - r0 < NB ? r_hi - r0 : NB);
            /* pack the block as NR-wide panels, transposed: panel p = j / NR holds Bp[p][kk][j % NR] = xb[r0 + j][kk];
             * rows past the end are zero */
            for (int j = 0; j < NB; ++j) {
                float s = 0.f;
                float* dst = Bt + (int64_t)(j / NR) * d * NR + (j % NR);
                if (j < rows) {
                    const float* src = xb + (r0 + j) * d;
                    for (int kk = 0; kk < d; ++kk) {
                        dst[(int64_t)kk * NR] = src[kk];
                        s += src[kk] * src[kk];
                    }
                } else {
                    for (int kk = 0; kk < d; ++kk) dst[(int64_t)kk * NR] = 0.f;
                }
                bnorm[j] = s;
            }
            for (int64_t q0 = 0; q0 < nq; q0 += MR) {
                const int mr = (int)(nq - q0 < MR ? nq - q0 : MR);
                for (int j0 = 0; j0 < rows; j0 += NR) {
                    micro(xq + q0 * d, d, mr, Bt + (int64_t)(j0 / NR) * d * NR, d, C);
                    const int nr = rows - j0 < NR ? rows - j0 : NR;
                    for (int i = 0; i < mr; ++i) {
                        float thr = thrv[q0 + i];
                        const float* c = C + i * NR;
                        for (int j = 0; j < nr; ++j) {
                            float s = c[j];
                            if (metric == 1) {
                                float dis = (qn[q0 + i] + bnorm[j0 + j]) - 2.0f * s;
                                s = -(dis > 0.f ? dis : 0.f);
                            }
                            if (s < thr) continue; /* almost always */
                            uint64_t* kq = mine + (q0 + i) * (int64_t)k;
                            t_insert(kq, k, ((uint64_t)t_ord32(s) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(r0 + j0 + j)));
                            thr = kq[k - 1] ? t_unord32((uint32_t)(kq[k - 1] >> 32)) : -__builtin_inff();
                            thrv[q0 + i] = thr;
                        }
                    }
                }
            }
        }
        free(Bt);
        free(thrv);
    }
    /* merge the per-thread lists of every query */
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nq; ++q) {
        uint64_t* o = out_keys + q * (int64_t)k;
        for (int t = 0; t < nthreads; ++t) {
            const uint64_t* l = lists + ((size_t)t * nq + q) * k;
            for (int j = 0; j < k && l[j]; ++j) t_insert(o, k, l[j]);
        }
    }
    free(lists);
    free(qn);
}
