// scan_kernels.hpp — the nq-small flat scan: score every stored row against one query and keep
// the best k, in ONE pass over the corpus.  Replaces faiss.IndexFlatIP.search at nq = 1
// (reference call sites minivectordb/vector_database.py:497, :514;
//  minivectordb/sharded_vector_database.py:626, :642).
//
// Roofline: HBM.  Algorithmic bytes per launch = n * ld * 4 (every stored row once).
//
// Shape of the work (GEMV, M <= 16 row of the guide's staging table: "load straight to VGPRs, deep
// unroll, late vmcnt", no LDS round trip for the streamed operand):
//   * a row is cut into 16-byte chunks; G = 2^g lanes share one row, lane t takes chunks
//     t, t+G, ... (C of them), so one wave-instruction reads 64/G rows x (G*16) contiguous bytes —
//     for d = 512: G = 64, C = 2, each global_load_dwordx4 is one 1-KiB contiguous half row;
//   * U row-groups are in flight per wave (U*C independent 16-B loads per lane) before the first
//     FMA needs its data; many such waves per CU keep >= 64 KiB in flight per CU;
//   * the query lives in registers (C float4 per lane), optionally L2-normalised in the prologue
//     (faiss.normalize_L2 on the query, vector_database.py:475);
//   * per row: C*4 FMAs per lane, then a log2(G)-step xor-butterfly over the row's lanes;
//   * selection: threshold-gated sorted insert per wave (topk_device.hpp), merged per block via
//     LDS, one sorted k-list per block written out; a tiny second kernel merges the block lists.
#pragma once
#include "topk_device.hpp"

namespace mvdb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kScanThreads = 256;
constexpr int kScanWaves = kScanThreads / kWave;

enum ScanMode { kModeTopK = 0, kModeScores = 1 };

struct ScanArgs {
    const float* X;       // [n_phys, ld] corpus
    int64_t n;            // rows to score (m for subset searches)
    int64_t ld;           // row stride in floats (multiple of 4)
    int d4;               // valid 16-B chunks per row (= ld / 4)
    const float* q;       // [nq, ldq] queries (device), ldq = ld, zero padded
    int normalize_q;      // L2-normalise the query in the prologue
    int k;                // <= kMaxFusedK in kModeTopK
    const int64_t* rows;  // optional subset: physical row of logical row r (NULL = identity)
    const uint64_t* mask; // optional subset as a BITMAP over the physical rows (bit r & 63 of word r >> 6): every row is
                          // scored at the full scan's rate, rows whose bit is clear are never offered; labels = row numbers
    uint64_t* cand;       // kModeTopK : [nq, gridDim.x, k] sorted block lists
    float* scores;        // kModeScores: [nq, n]
    const int* gate = nullptr;  // GATED instantiations only: query blockIdx.y is scanned iff *gate > gate_lo + blockIdx.y
    int gate_lo = 0;
    const int* need = nullptr;  // GATED, optional: one word per query of the launch — a query whose word is 0 is skipped (the
                                // rescue pass has answered it, mvdb.hip)
};

template <int G>
__device__ __forceinline__ float group_reduce_add(float v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// METRIC 0: inner product (score = q.x).  METRIC 1: squared L2 (score = -|q-x|^2).
// SEL    : 0 every row; 1 rows[] indirection (compile-time, so the identity path carries no branch and — crucially —
//          no `s_waitcnt vmcnt(0)` between the row loads of a batch); 2 bitmap (a.mask): rows in corpus order, the bit of
//          each row of the NEXT batch fetched behind the current batch's row loads; a clear bit only keeps the row out of
//          the top-k gate (and writes -inf in score mode) — dense filters (exclude-filters keep most rows) then cost one
//          full scan instead of a gather that is slower than the scan it avoids.
// MASKED : the row has fewer than G*C chunks (lanes with chunk >= d4 load nothing).  When the row
//          fills every lane the loads are unconditional: no exec-mask branches in the loop.
// (Explicit register double-buffering of batches was measured and dropped: 3-5 % slower than relying
//  on the other resident waves, profiles/r01_sweep_scan_variants.txt lineage.)
// GATED  : the launch is enabled per query ON THE DEVICE (the exact re-run of the L2 queries a certified batch pass could
//          not certify, mvdb.hip): a separate instantiation — the ungated kernels carry no test.
template <int G, int C, int U, int METRIC, int MODE, bool NT = true, int SEL = 0, bool MASKED = true, bool GATED = false>
__global__ __launch_bounds__(kScanThreads) void flat_scan_kernel(ScanArgs a) {
    if (GATED && (*a.gate <= a.gate_lo + (int)blockIdx.y || (a.need && a.need[blockIdx.y] == 0))) return;
    constexpr bool SUBSET = SEL == 1;
    constexpr int RPI = kWave / G;  // rows per wave-instruction
    constexpr int RB = RPI * U;     // rows per wave batch
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int t = lane % G;  // chunk lane within the row
    const int g = lane / G;  // row slot within the instruction
    const int qi = blockIdx.y;

    // ---- query -> registers ------------------------------------------------------------------
    f32x4 qv[C];
    bool cvalid[C];
    const float* qptr = a.q + (int64_t)qi * a.ld;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int chunk = c * G + t;
        cvalid[c] = !MASKED || chunk < a.d4;
        qv[c] = cvalid[c] ? *reinterpret_cast<const f32x4*>(qptr + chunk * 4) : f32x4{0, 0, 0, 0};
    }
    if (a.normalize_q) {
        // faiss fvec_renorm_L2: nr = |q|^2; if (nr > 0) q *= 1/sqrt(nr)
        float nr = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c)
            nr += qv[c].x * qv[c].x + qv[c].y * qv[c].y + qv[c].z * qv[c].z + qv[c].w * qv[c].w;
        nr = group_reduce_add<G>(nr);
        if (nr > 0.f) {
            const float inorm = 1.0f / sqrtf(nr);
#pragma unroll
            for (int c = 0; c < C; ++c) qv[c] *= inorm;
        }
    }

    WaveTopK tk;
    tk.init(a.k);

    const int64_t nwaves_total = (int64_t)gridDim.x * kScanWaves;
    const int64_t gw = (int64_t)blockIdx.x * kScanWaves + wave;
    const int64_t nbatches = (a.n + RB - 1) / RB;
    const int64_t last = a.n - 1;

    // physical rows of batch b (SUBSET: one 8-byte load per row through rows[])
    auto batch_rows = [&](int64_t b, int64_t (&pr)[U]) {
        const int64_t row0 = b * RB + g;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t r = row0 + (int64_t)u * RPI;
            r = r < last ? r : last;  // clamp: tail lanes re-read the last row, result discarded
            pr[u] = SUBSET ? a.rows[r] : r;
        }
    };
    // issue the U*C loads of a batch (nothing waits here)
    auto load_batch = [&](const int64_t (&pr)[U], f32x4 (&x)[U][C]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* p = a.X + pr[u] * a.ld + t * 4;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const f32x4* src = reinterpret_cast<const f32x4*>(p + c * G * 4);
                if (MASKED)
                    x[u][c] = cvalid[c] ? (NT ? __builtin_nontemporal_load(src) : *src) : f32x4{0, 0, 0, 0};
                else
                    x[u][c] = NT ? __builtin_nontemporal_load(src) : *src;
            }
        }
    };

    // SEL == 2: is row r selected?  (one 8-byte load per row; the lanes of a row read the same word)
    auto batch_bits = [&](int64_t b, bool (&sel)[U]) {
        const int64_t row0 = b * RB + g;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t r = row0 + (int64_t)u * RPI;
            r = r < last ? r : last;
            sel[u] = (a.mask[r >> 6] >> (r & 63)) & 1ull;
        }
    };
    // reduce batch b and offer its rows to the top-k list / write its scores
    auto consume_batch = [&](int64_t b, f32x4 (&x)[U][C], const bool (&sel)[U]) {
        const int64_t row0 = b * RB + g;
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (METRIC == 0) {
                    acc = fmaf(x[u][c].x, qv[c].x, acc);
                    acc = fmaf(x[u][c].y, qv[c].y, acc);
                    acc = fmaf(x[u][c].z, qv[c].z, acc);
                    acc = fmaf(x[u][c].w, qv[c].w, acc);
                } else {
                    const f32x4 df = qv[c] - x[u][c];
                    acc = fmaf(df.x, df.x, acc);
                    acc = fmaf(df.y, df.y, acc);
                    acc = fmaf(df.z, df.z, acc);
                    acc = fmaf(df.w, df.w, acc);
                }
            }
            s[u] = acc;
        }
        // the U butterflies are independent: interleave them step by step
#pragma unroll
        for (int m = G / 2; m >= 1; m >>= 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] += __shfl_xor(s[u], m);
        }
        if (METRIC != 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] = -s[u];
        }
        if (MODE == kModeScores) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = row0 + (int64_t)u * RPI;
                if (t == 0 && r < a.n) {
                    const float v = s[u];
                    a.scores[(int64_t)qi * a.n + r] = (v == v && (SEL != 2 || sel[u])) ? v : -INFINITY;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = row0 + (int64_t)u * RPI;
                // gate on the score alone (NaN fails); exact 64-bit order decided inside offer()
                const bool pass = (t == 0) && (r < a.n) && (SEL != 2 || sel[u]) && (s[u] >= tk.thr_score);
                if (__ballot(pass)) tk.offer(pass ? make_key(s[u], (uint32_t)r) : 0ull);
            }
        }
    };

    bool all_rows[U];
#pragma unroll
    for (int u = 0; u < U; ++u) all_rows[u] = true;
    if (SEL == 2) {
        bool sel[U], seln[U];
        if (gw < nbatches) batch_bits(gw, sel);
        for (int64_t b = gw; b < nbatches; b += nwaves_total) {
            f32x4 x[U][C];
            int64_t pr[U];
            batch_rows(b, pr);
            load_batch(pr, x);
            const int64_t bn = b + nwaves_total;
            batch_bits(bn < nbatches ? bn : b, seln);  // behind the row loads: never waited for by its own batch
            consume_batch(b, x, sel);
#pragma unroll
            for (int u = 0; u < U; ++u) sel[u] = seln[u];
        }
    } else if (SUBSET) {
        // the row ids of the NEXT batch are fetched behind this batch's row loads, so that a batch never waits for
        // its own indirection (id load -> vmcnt(0) -> row loads serialised two round trips per batch: 2.0-2.3 TB/s of
        // rows touched at 10M x 512)
        int64_t pr[U], pn[U];
        if (gw < nbatches) batch_rows(gw, pr);
        for (int64_t b = gw; b < nbatches; b += nwaves_total) {
            f32x4 x[U][C];
            load_batch(pr, x);
            const int64_t bn = b + nwaves_total;
            batch_rows(bn < nbatches ? bn : b, pn);
            consume_batch(b, x, all_rows);
#pragma unroll
            for (int u = 0; u < U; ++u) pr[u] = pn[u];
        }
    } else {
        for (int64_t b = gw; b < nbatches; b += nwaves_total) {
            f32x4 x[U][C];
            int64_t pr[U];
            batch_rows(b, pr);
            load_batch(pr, x);
            consume_batch(b, x, all_rows);
        }
    }

    if (MODE == kModeTopK) {
        __shared__ uint64_t sh[(kScanWaves - 1) * kWave];
        block_merge_topk(tk, sh, kScanWaves);
        if (wave == 0 && lane < a.k)
            a.cand[((int64_t)qi * gridDim.x + blockIdx.x) * a.k + lane] = tk.key;
    }
}

// ---- second stage: merge `nlists` sorted k-lists per query and emit (D, I) --------------------
// grid = (nq), block = 1024.  keys[nq, nlists, k].  Labels: position -> label_offset + row.
// METRIC 1 stores -dist in the key; D gets +dist back.
struct MergeArgs {
    const uint64_t* keys;
    int nlists;
    int k;
    int metric;
    int64_t label_offset;
    float* D;    // [nq, k]
    int64_t* I;  // [nq, k]
    const int* gate = nullptr;  // optional device-side enable: query blockIdx.x is merged iff *gate > gate_lo + blockIdx.x
    int gate_lo = 0;
    const int* need = nullptr;  // optional second enable, one word per PASS of per_pass queries: nothing of a pass is merged while its word is 0
    int per_pass = 0;           // > 0: the launch covers several passes — query blockIdx.x belongs to pass blockIdx.x / per_pass, whose
    int64_t pass_stride = 0;    //      lists start pass_stride keys after the pass before
};

constexpr int kMergeThreads = 1024;
constexpr int kMergeWaves = kMergeThreads / kWave;
constexpr int kMergeUnroll = 8;

__global__ __launch_bounds__(kMergeThreads) void merge_keys_kernel(MergeArgs a) {
    if (a.gate && *a.gate <= a.gate_lo + (int)blockIdx.x) return;
    const int pass = a.per_pass > 0 ? (int)blockIdx.x / a.per_pass : 0;
    if (a.need && a.need[pass] == 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int qi = blockIdx.x;
    const int64_t total = (int64_t)a.nlists * a.k;
    const uint64_t* src = a.keys + (int64_t)pass * a.pass_stride + (int64_t)(qi - pass * a.per_pass) * total;
    WaveTopK tk;
    tk.init(a.k);
    // each wave walks its slice in steps of 64 x kMergeUnroll keys: the loads of a step are
    // independent (issued back to back), only then are the candidates offered to the list
    for (int64_t base = (int64_t)wave * kWave * kMergeUnroll; base < total;
         base += (int64_t)kMergeThreads * kMergeUnroll) {
        uint64_t c[kMergeUnroll];
#pragma unroll
        for (int j = 0; j < kMergeUnroll; ++j) {
            const int64_t i = base + j * kWave + lane;
            c[j] = i < total ? src[i] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < kMergeUnroll; ++j) tk.offer(c[j]);
    }
    __shared__ uint64_t sh[(kMergeWaves - 1) * kWave];
    block_merge_topk(tk, sh, kMergeWaves);
    if (wave == 0 && lane < a.k) {
        const uint64_t key = tk.key;
        float d;
        int64_t id;
        if (key) {
            const float s = key_score(key);
            d = a.metric == 0 ? s : -s;
            id = a.label_offset + (int64_t)key_row(key);
        } else {  // faiss convention for missing results
            d = a.metric == 0 ? -3.402823466e+38f : 3.402823466e+38f;
            id = -1;
        }
        a.D[(int64_t)qi * a.k + lane] = d;
        a.I[(int64_t)qi * a.k + lane] = id;
    }
}

// Merge already-materialised (D, I) lists (all-gather layout [nlists, nq, k]) — the exchange
// step of the row-partitioned multi-GPU search.  Ties: lower label first.
struct MergeDIArgs {
    const float* D;
    const int64_t* I;
    int64_t strideD, strideI;  // elements between consecutive lists
    int nlists, nq, k, metric;
    float* Dout;
    int64_t* Iout;
};

__global__ __launch_bounds__(kWave) void merge_di_kernel(MergeDIArgs a) {
    // one wave per query; nlists*k candidates.  Keys need a 32-bit tiebreak: use the candidate's
    // rank by label among equal scores = position in (list, slot) order, which is ascending in
    // label because shard bases ascend with list index and slots are sorted (score desc, row asc).
    const int lane = threadIdx.x;
    const int qi = blockIdx.x;
    const int total = a.nlists * a.k;
    WaveTopK tk;
    tk.init(a.k);
    for (int base = 0; base < total; base += kWave) {
        const int i = base + lane;
        uint64_t key = 0;
        if (i < total) {
            const int l = i / a.k, j = i - l * a.k;
            const int64_t src = ((int64_t)qi) * a.k + j;
            if (a.I[l * a.strideI + src] >= 0) {
                const float d = a.D[l * a.strideD + src];
                key = make_key(a.metric == 0 ? d : -d, (uint32_t)i);
            }
        }
        tk.offer(key);
    }
    if (lane < a.k) {
        float d = a.metric == 0 ? -3.402823466e+38f : 3.402823466e+38f;
        int64_t id = -1;
        if (tk.key) {
            const int i = (int)key_row(tk.key);
            const int l = i / a.k, j = i - l * a.k;
            const int64_t src = ((int64_t)qi) * a.k + j;
            d = a.D[l * a.strideD + src];
            id = a.I[l * a.strideI + src];
        }
        a.Dout[(int64_t)qi * a.k + lane] = d;
        a.Iout[(int64_t)qi * a.k + lane] = id;
    }
}

}  // namespace mvdb
