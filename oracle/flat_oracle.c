/*
 * flat_oracle.c — CPU restatement of the flat inner-product search on MiniVectorDB's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the product
 * (minivectordb_amd/, libmvdb.so); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the CPU number printed beside the GPU's.
 *
 * PARITY UNPINNED at the faiss boundary: the reference's arithmetic for this path lives in the
 * un-vendored, unpinned `faiss-cpu` wheel (reference requirements.txt:7), which is absent from
 * the reference tree and from this image, and none of the reference's tests asserts a score
 * (SURVEY.md §4, §8c).  What is restated here is faiss' published algorithm as called by the
 * reference:
 *   - faiss.normalize_L2 (fvec_renorm_L2): nr = sum x_j^2; if nr > 0: x *= 1/sqrtf(nr)
 *       reference call sites: minivectordb/vector_database.py:45 (corpus, in place), :475 (query)
 *   - faiss.IndexFlatIP.search at nq < 20 (the sequential, non-BLAS path): one
 *     fvec_inner_product per stored row, a k-min-heap with replace-if-better, results sorted by
 *     score descending (heap_reorder); ties resolved by the lower id (faiss' cmp2 ordering)
 *       reference call sites: minivectordb/vector_database.py:497, :514;
 *                             minivectordb/sharded_vector_database.py:626, :642
 * faiss lets the compiler vectorise fvec_inner_product (summation order is build dependent); this
 * file fixes ONE order — 8 interleaved partial sums, combined pairwise — and the float64
 * adjudicator (oracle_flat_search_f64) classifies any id difference between two fp32
 * implementations as a near-tie or a real error.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_METRIC_IP 0
#define ORACLE_METRIC_L2 1

/* ---- synthetic stream (bit-identical to minivectordb_amd/csrc/util_kernels.hpp) -------------- */
static inline uint32_t pcg_hash32(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}
static inline float synth_element(uint32_t seed_lo, uint32_t seed_hi, uint64_t ctr) {
    const uint32_t lo = (uint32_t)ctr, hi = (uint32_t)(ctr >> 32);
    const uint32_t h0 = pcg_hash32(lo ^ pcg_hash32(hi ^ pcg_hash32(seed_lo ^ pcg_hash32(seed_hi))));
    const uint32_t h1 = pcg_hash32(h0 ^ 0x9E3779B9u);
    const int32_t sum = (int32_t)((h0 & 0xFFFFu) + (h0 >> 16) + (h1 & 0xFFFFu) + (h1 >> 16));
    return (float)(sum - 131070) * (1.0f / 131072.0f);
}
/* The three families of rows (seed >> 56): 0 zero-mean bell (every BASELINE config), 1 all-positive uniform [0, 1) (what the
 * reference's own tests store: numpy.random.rand, tests/test_sharded_multithreaded_operations.py:22), 2 clustered (4,096
 * centres shared by every seed + 2^-4 noise; 1 row in 256 noise-free = exact duplicates, 1 in 256 at 2^-13 = near
 * duplicates).  Bit for bit minivectordb_amd/csrc/util_kernels.hpp synth_value: integer hashes, exact conversions,
 * power-of-two scales, at most ONE rounding (the sum of family 2; volatile keeps the compiler from fusing it differently). */
static inline float synth_value(uint64_t seed, uint64_t row, uint32_t col, uint32_t d) {
    const uint32_t slo = (uint32_t)seed, shi = (uint32_t)(seed >> 32);
    const uint32_t family = (uint32_t)(seed >> 56);
    const uint64_t ctr = row * (uint64_t)d + col;
    if (family == 0) return synth_element(slo, shi, ctr);
    if (family == 1) {
        const uint32_t lo = (uint32_t)ctr, hi = (uint32_t)(ctr >> 32);
        const uint32_t h = pcg_hash32(lo ^ pcg_hash32(hi ^ pcg_hash32(slo ^ pcg_hash32(shi))));
        return (float)(h >> 8) * (1.0f / 16777216.0f);
    }
    const uint32_t rlo = (uint32_t)row, rhi = (uint32_t)(row >> 32);
    const uint32_t hr = pcg_hash32(rlo ^ pcg_hash32(rhi ^ pcg_hash32(slo ^ pcg_hash32(shi ^ 0x5BD1E995u))));
    const uint32_t centre = hr & 4095u, kind = (hr >> 12) & 255u;
    const float c = synth_element(0xC3A5C85Cu, 2u, (uint64_t)centre * d + col);
    if (kind == 0) return c;
    const float e = synth_element(slo, shi, ctr);
    const float scaled = e * (kind == 1 ? 1.0f / 8192.0f : 1.0f / 16.0f);  /* exact */
    return c + scaled;
}
void oracle_synth_fill(float* out, int64_t n, int d, uint64_t seed, int64_t first_row) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < d; ++j)
            out[i * d + j] = synth_value(seed, (uint64_t)(first_row + i), (uint32_t)j, (uint32_t)d);
}

/* ---- fp32 kernels with a fixed summation order ---------------------------------------------- */
static inline float dot_f32(const float* x, const float* y, int d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int j = 0;
    for (; j + 8 <= d; j += 8)
        for (int l = 0; l < 8; ++l) acc[l] += x[j + l] * y[j + l];
    float tail = 0.f;
    for (; j < d; ++j) tail += x[j] * y[j];
    const float s0 = (acc[0] + acc[4]) + (acc[2] + acc[6]);
    const float s1 = (acc[1] + acc[5]) + (acc[3] + acc[7]);
    return (s0 + s1) + tail;
}
static inline float l2sqr_f32(const float* x, const float* y, int d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int j = 0;
    for (; j + 8 <= d; j += 8)
        for (int l = 0; l < 8; ++l) {
            const float t = x[j + l] - y[j + l];
            acc[l] += t * t;
        }
    float tail = 0.f;
    for (; j < d; ++j) {
        const float t = x[j] - y[j];
        tail += t * t;
    }
    const float s0 = (acc[0] + acc[4]) + (acc[2] + acc[6]);
    const float s1 = (acc[1] + acc[5]) + (acc[3] + acc[7]);
    return (s0 + s1) + tail;
}

/* faiss fvec_renorm_L2 */
void oracle_normalize_l2(float* x, int64_t n, int d) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float* r = x + i * d;
        const float nr = dot_f32(r, r, d);
        if (nr > 0) {
            const float inorm = 1.0f / sqrtf(nr);
            for (int j = 0; j < d; ++j) r[j] *= inorm;
        }
    }
}

/* ---- k-heap on (score, id): "better" = higher score, then lower id ------------------------- */
typedef struct {
    double s; /* score as double so the same heap serves the fp64 adjudicator */
    int64_t id;
} item_t;
static inline int better(item_t a, item_t b) { return a.s > b.s || (a.s == b.s && a.id < b.id); }
/* min-heap: root = worst kept item */
static void heap_sift_down(item_t* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && better(h[w], h[l])) w = l;
        if (r < n && better(h[w], h[r])) w = r;
        if (w == i) return;
        item_t t = h[i];
        h[i] = h[w];
        h[w] = t;
        i = w;
    }
}
static void heap_sift_up(item_t* h, int i) {
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!better(h[p], h[i])) return;
        item_t t = h[i];
        h[i] = h[p];
        h[p] = t;
        i = p;
    }
}
static inline void heap_offer(item_t* h, int* cnt, int k, item_t it) {
    if (it.s != it.s) return; /* NaN never enters (faiss: the comparison with the root fails) */
    if (*cnt < k) {
        h[*cnt] = it;
        heap_sift_up(h, (*cnt)++);
    } else if (better(it, h[0])) {
        h[0] = it;
        heap_sift_down(h, k, 0);
    }
}
static int cmp_desc(const void* a, const void* b) {
    const item_t *x = (const item_t*)a, *y = (const item_t*)b;
    if (better(*x, *y)) return -1;
    if (better(*y, *x)) return 1;
    return 0;
}

static void scan_range(const float* x, int64_t r0, int64_t r1, int d, const float* q, int metric,
                       const int64_t* rows, int f64, item_t* heap, int* cnt, int k) {
    for (int64_t i = r0; i < r1; ++i) {
        const float* row = x + (rows ? rows[i] : i) * (int64_t)d;
        item_t it;
        it.id = i;
        if (f64) {
            double acc = 0.0;
            if (metric == ORACLE_METRIC_IP)
                for (int j = 0; j < d; ++j) acc += (double)row[j] * (double)q[j];
            else
                for (int j = 0; j < d; ++j) {
                    const double t = (double)q[j] - (double)row[j];
                    acc -= t * t;
                }
            it.s = acc;
        } else {
            it.s = metric == ORACLE_METRIC_IP ? (double)dot_f32(q, row, d) : -(double)l2sqr_f32(q, row, d);
        }
        heap_offer(heap, cnt, k, it);
    }
}

/*
 * x[n_phys,d], q[nq,d]; rows (optional, length n) selects/permutes the rows that form the searched
 * set — labels are positions in rows[], as with the reference's throw-away sub-index
 * (vector_database.py:510-523).  normalize_q applies oracle_normalize_l2 to a copy of each query.
 * D[nq,k] fp32 (or D64[nq,k] when f64 != 0), I[nq,k]; missing slots: -1 and -FLT_MAX / +FLT_MAX.
 * nthreads > 1 partitions the rows over OpenMP threads (per-thread heaps, merged) — results are
 * identical to the sequential scan because the (score,id) order is total.
 */
static void search_impl(const float* x, int64_t n, int d, const float* q, int nq, int k, int metric,
                        int normalize_q, const int64_t* rows, int f64, int nthreads, float* D,
                        double* D64, int64_t* I) {
    if (nthreads < 1) nthreads = 1;
    float* qn = (float*)malloc((size_t)nq * d * sizeof(float));
    memcpy(qn, q, (size_t)nq * d * sizeof(float));
    if (normalize_q) {
        int saved = 1;
#ifdef _OPENMP
        saved = omp_get_max_threads();
        omp_set_num_threads(1);
#endif
        oracle_normalize_l2(qn, nq, d);
#ifdef _OPENMP
        omp_set_num_threads(saved);
#endif
    }
    item_t* heaps = (item_t*)malloc((size_t)nthreads * k * sizeof(item_t));
    int* cnts = (int*)malloc((size_t)nthreads * sizeof(int));
    for (int qi = 0; qi < nq; ++qi) {
        const float* qq = qn + (size_t)qi * d;
        for (int t = 0; t < nthreads; ++t) cnts[t] = 0;
        if (nthreads == 1) {
            scan_range(x, 0, n, d, qq, metric, rows, f64, heaps, &cnts[0], k);
        } else {
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
            for (int t = 0; t < nthreads; ++t) {
                const int64_t r0 = n * t / nthreads, r1 = n * (t + 1) / nthreads;
                scan_range(x, r0, r1, d, qq, metric, rows, f64, heaps + (size_t)t * k, &cnts[t], k);
            }
            for (int t = 1; t < nthreads; ++t)
                for (int j = 0; j < cnts[t]; ++j)
                    heap_offer(heaps, &cnts[0], k, heaps[(size_t)t * k + j]);
        }
        qsort(heaps, cnts[0], sizeof(item_t), cmp_desc);
        for (int j = 0; j < k; ++j) {
            double s;
            int64_t id;
            if (j < cnts[0]) {
                s = metric == ORACLE_METRIC_IP ? heaps[j].s : -heaps[j].s;
                id = heaps[j].id;
            } else {
                s = metric == ORACLE_METRIC_IP ? -3.402823466e+38 : 3.402823466e+38;
                id = -1;
            }
            if (D) D[(size_t)qi * k + j] = (float)s;
            if (D64) D64[(size_t)qi * k + j] = s;
            I[(size_t)qi * k + j] = id;
        }
    }
    free(heaps);
    free(cnts);
    free(qn);
}

void oracle_flat_search(const float* x, int64_t n, int d, const float* q, int nq, int k, int metric,
                        int normalize_q, const int64_t* rows, int nthreads, float* D, int64_t* I) {
    search_impl(x, n, d, q, nq, k, metric, normalize_q, rows, 0, nthreads, D, NULL, I);
}

void oracle_flat_search_f64(const float* x, int64_t n, int d, const float* q, int nq, int k,
                            int metric, int normalize_q, const int64_t* rows, int nthreads,
                            double* D64, int64_t* I) {
    search_impl(x, n, d, q, nq, k, metric, normalize_q, rows, 1, nthreads, NULL, D64, I);
}

/*
 * Many queries over ONE block of rows, for the full-size parity tests (tests/bigcheck.py streams a 10M / 80M-row device
 * corpus through here in 1M-row blocks).  The SAME arithmetic and order as oracle_flat_search — every (query, row) pair is
 * one dot_f32 / l2sqr_f32, candidates are ordered by (score desc, id asc) — only the loop nest differs: threads own row
 * ranges, walk them in tiles that stay in their cache and score every query against a tile before moving on (the
 * per-query scan above re-streams the block from DRAM once per query).  Labels are id_base + row, so per-block results
 * merge on the host by the same total order (oracle_merge_topk).  Rows with NaN scores never enter, as above.
 * keep (optional, one byte per row of the block): rows whose byte is 0 are not part of the searched set (a filter).
 * D[nq,k], I[nq,k]; missing slots -1 / -FLT_MAX (IP) or +FLT_MAX (L2).
 */
void oracle_flat_search_block(const float* x, int64_t n, int d, const float* q, int nq, int k, int metric,
                              int64_t id_base, const uint8_t* keep, int nthreads, float* D, int64_t* I) {
    if (nthreads < 1) nthreads = 1;
    const int64_t tile = 128;
    item_t* heaps = (item_t*)malloc((size_t)nthreads * nq * k * sizeof(item_t));
    int* cnts = (int*)calloc((size_t)nthreads * nq, sizeof(int));
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; ++t) {
        const int64_t r0 = n * t / nthreads, r1 = n * (t + 1) / nthreads;
        item_t* hp = heaps + (size_t)t * nq * k;
        int* cp = cnts + (size_t)t * nq;
        for (int64_t b = r0; b < r1; b += tile) {
            const int64_t e = b + tile < r1 ? b + tile : r1;
            for (int qi = 0; qi < nq; ++qi) {
                const float* qq = q + (size_t)qi * d;
                for (int64_t i = b; i < e; ++i) {
                    if (keep && !keep[i]) continue;
                    const float* row = x + i * (int64_t)d;
                    item_t it;
                    it.id = id_base + i;
                    it.s = metric == ORACLE_METRIC_IP ? (double)dot_f32(qq, row, d) : -(double)l2sqr_f32(qq, row, d);
                    heap_offer(hp + (size_t)qi * k, &cp[qi], k, it);
                }
            }
        }
    }
    for (int qi = 0; qi < nq; ++qi) {
        item_t* h0 = heaps + (size_t)qi * k;
        for (int t = 1; t < nthreads; ++t) {
            const item_t* ht = heaps + ((size_t)t * nq + qi) * k;
            for (int j = 0; j < cnts[(size_t)t * nq + qi]; ++j) heap_offer(h0, &cnts[qi], k, ht[j]);
        }
        qsort(h0, cnts[qi], sizeof(item_t), cmp_desc);
        for (int j = 0; j < k; ++j) {
            const int have = j < cnts[qi];
            D[(size_t)qi * k + j] = have ? (float)(metric == ORACLE_METRIC_IP ? h0[j].s : -h0[j].s)
                                         : (metric == ORACLE_METRIC_IP ? -3.402823466e+38f : 3.402823466e+38f);
            I[(size_t)qi * k + j] = have ? h0[j].id : -1;
        }
    }
    free(heaps);
    free(cnts);
}

/* Merge `parts` per-block result lists (Dp[parts][nq][k], Ip likewise; -1 = empty slot) into the top-k of their union by
 * (score desc for IP / distance asc for L2, id asc): exact, because that order is total. */
void oracle_merge_topk(const float* Dp, const int64_t* Ip, int parts, int nq, int k, int metric, float* D, int64_t* I) {
    item_t* all = (item_t*)malloc((size_t)parts * k * sizeof(item_t));
    for (int qi = 0; qi < nq; ++qi) {
        int m = 0;
        for (int p = 0; p < parts; ++p)
            for (int j = 0; j < k; ++j) {
                const size_t at = ((size_t)p * nq + qi) * k + j;
                if (Ip[at] < 0) continue;
                all[m].s = metric == ORACLE_METRIC_IP ? (double)Dp[at] : -(double)Dp[at];
                all[m].id = Ip[at];
                ++m;
            }
        qsort(all, m, sizeof(item_t), cmp_desc);
        for (int j = 0; j < k; ++j) {
            const int have = j < m;
            D[(size_t)qi * k + j] = have ? (float)(metric == ORACLE_METRIC_IP ? all[j].s : -all[j].s)
                                         : (metric == ORACLE_METRIC_IP ? -3.402823466e+38f : 3.402823466e+38f);
            I[(size_t)qi * k + j] = have ? all[j].id : -1;
        }
    }
    free(all);
}

/* float64 scores of listed rows against ONE (already prepared) query: out[m] */
void oracle_scores_f64(const float* x, int d, const float* q, int metric, const int64_t* rows,
                       int64_t m, double* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < m; ++i) {
        const float* row = x + rows[i] * (int64_t)d;
        double acc = 0.0;
        if (metric == ORACLE_METRIC_IP)
            for (int j = 0; j < d; ++j) acc += (double)row[j] * (double)q[j];
        else
            for (int j = 0; j < d; ++j) {
                const double t = (double)q[j] - (double)row[j];
                acc += t * t;
            }
        out[i] = acc;
    }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* NUMA-friendly copy for the multi-threaded leg of bench.py's cpu_baseline: thread t of `nthreads` copies (first
 * touches) exactly the row range [n t / nthreads, n (t + 1) / nthreads) that it will scan in oracle_flat_search with
 * the same thread count, so each thread's rows sit in memory local to it (one thread touching the whole matrix puts
 * it behind a single memory controller). */
/* The same placement, one block at a time: src[0, m) becomes rows [row0, row0 + m) of dst[n_total, d], each row written (first
 * touched) by the thread that scans it when dst is searched with `nthreads` threads — a corpus streamed from the GPU in
 * blocks lands NUMA-placed without ever existing twice on the host. */
void oracle_first_touch_copy_block(float* dst, const float* src, int64_t n_total, int d, int nthreads, int64_t row0, int64_t m) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; ++t) {
        int64_t r0 = n_total * t / nthreads, r1 = n_total * (t + 1) / nthreads;
        if (r0 < row0) r0 = row0;
        if (r1 > row0 + m) r1 = row0 + m;
        if (r1 > r0) memcpy(dst + r0 * d, src + (r0 - row0) * d, (size_t)(r1 - r0) * d * sizeof(float));
    }
}

void oracle_first_touch_copy(float* dst, const float* src, int64_t n, int d, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; ++t) {
        const int64_t r0 = n * t / nthreads, r1 = n * (t + 1) / nthreads;
        memcpy(dst + r0 * d, src + r0 * d, (size_t)(r1 - r0) * d * sizeof(float));
    }
}

