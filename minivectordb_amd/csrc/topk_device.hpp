// topk_device.hpp — wave-level top-k selection primitives for gfx950 (64-lane wavefronts).
//
// A candidate is one 64-bit key:  (order-preserving image of the fp32 score) << 32 | (~row).
// Larger key == better candidate: higher score first, then LOWER row number — the deterministic
// restatement of faiss' (score, id) heap ordering used by IndexFlatIP.search
// (reference call site: minivectordb/vector_database.py:497).  Key 0 is the "empty slot" sentinel:
// every real key is > 0 because the image of any non-NaN float is > 0.
//
// A wave keeps its best k <= 64 keys SORTED DESCENDING, one per lane (lane i = i-th best).  Rows
// are admitted through a threshold gate (the k-th best so far), so after warm-up almost no row
// reaches the insert path: expected inserts per wave ~ k*ln(rows_per_wave/k).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvdb {

constexpr int kWave = 64;
constexpr int kMaxFusedK = 64;  // fused select keeps one candidate per lane

__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float score, uint32_t row) {
    return ((uint64_t)f2ord(score) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ float key_score(uint64_t key) { return ord2f((uint32_t)(key >> 32)); }
__device__ __forceinline__ uint32_t key_row(uint64_t key) { return 0xFFFFFFFFu - (uint32_t)key; }

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// Sorted per-wave candidate list.
struct WaveTopK {
    uint64_t key;    // this lane's slot (lanes >= k stay 0 and never take part)
    uint64_t thr;    // wave-uniform: key of slot k-1 (0 while the list is not full)
    float thr_score; // wave-uniform gate: a score below this cannot enter
    int k;

    __device__ __forceinline__ void init(int k_) {
        key = 0;
        thr = 0;
        thr_score = -INFINITY;
        k = k_;
    }

    // kn is wave-uniform.  Inserts kn if it beats the current k-th best.
    __device__ __forceinline__ void insert_uniform(uint64_t kn) {
        if (kn <= thr) return;
        const int lane = threadIdx.x & (kWave - 1);
        const int better = __popcll(__ballot(key > kn));  // sorted => lanes [0,better) are better
        const uint64_t up = __shfl_up(key, 1);
        if (lane == better)
            key = kn;
        else if (lane > better && lane < k)
            key = up;
        thr = readlane_u64(key, k - 1);
        thr_score = thr ? key_score(thr) : -INFINITY;
    }

    // Every lane offers one candidate (cand == 0 means "none").  Wave-uniform control flow.
    __device__ __forceinline__ void offer(uint64_t cand) {
        uint64_t mask = __ballot(cand > thr);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            insert_uniform(readlane_u64(cand, src));
        }
    }
};

// Certificate of the L2 metric's batch passes, which nominate by INNER PRODUCT (rows of nearly equal norm: the IP ranking is
// the L2 ranking up to the spread of |x|^2).  u bounds the approximate inner product of every row that was dropped, eps_q =
// eps |q| its error: such a row's true distance is at least |q|^2 + n2lo - 2 (u + eps_q).  rk is the fp32 re-score
// sum (q - x)^2 of the k-th result: a sum of non-negative terms, relative error <= (depth + 2) 2^-23 < 2e-5 for d <= 4096.
// |q| carries <= 4e-6 relative error (fp32 tree sum + sqrt); the last term covers the rounding of this expression itself.
__device__ __forceinline__ bool l2_certified(float rk, float u, float eps_q, float qn, float n2lo) {
    const float lower = fmaf(qn * 0.99999f, qn, n2lo) - 2.0f * (u + eps_q);
    return rk * 1.00002f < lower - 1e-6f * (qn * qn + fabsf(n2lo) + 2.0f * fabsf(u));
}

// The gate in front of the list inserts of the multi-query passes.  A lane keeps, per query, the SCORE its list's k-th key holds
// (or an admission floor) and the ROW of that key: a candidate goes to the insert only if its (score, row) key beats it —
// a higher score, or the same score and a lower row.  Until round 5 the gate compared scores alone (`s >= thr`) and left the
// tie to the insert: on a corpus of duplicates every row tied with the k-th score and paid a wave-cooperative insert attempt
// that the key order then refused — 64 queries over 1M identical rows took 21.8 ms instead of 0.33
// (benchmarks/ties_probe.py).  A floor (no list key behind the score) carries row 0: nothing ties its way past a floor —
// floors come from rows of earlier phases, i.e. lower rows.
__device__ __forceinline__ bool beats_key(float s, uint32_t row, float thr, uint32_t thr_row) {
    return s > thr || (s == thr && row < thr_row);
}
__device__ __forceinline__ void set_threshold(uint64_t kth, float floor0, float& thr, uint32_t& thr_row) {
    const float ks = kth ? key_score(kth) : -INFINITY;
    const bool from_list = kth != 0ull && ks >= floor0;
    thr = from_list ? ks : floor0;
    thr_row = from_list ? key_row(kth) : 0u;
}

// wave-cooperative sorted insert into an LDS list; returns the list's new k-th key
__device__ __forceinline__ uint64_t lds_list_insert(uint64_t* list, int k, uint64_t key, int lane) {
    const uint64_t cur = lane < k ? list[lane] : 0ull;
    const int better = __popcll(__ballot(cur > key));
    const uint64_t up = __shfl_up(cur, 1);
    uint64_t nv = cur;
    if (lane == better)
        nv = key;
    else if (lane > better)
        nv = up;
    if (lane < k && better < k) list[lane] = nv;
    return readlane_u64(better < k ? nv : cur, k - 1);
}

// Merge the sorted lists of all waves of the block into wave 0's list.
// sh must hold (nwaves-1)*64 keys.  Must be called by every thread of the block.
__device__ __forceinline__ void block_merge_topk(WaveTopK& tk, uint64_t* sh, int nwaves) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    if (wave > 0) sh[(wave - 1) * kWave + lane] = tk.key;
    __syncthreads();
    if (wave == 0) {
        for (int w = 0; w < nwaves - 1; ++w) tk.offer(sh[w * kWave + lane]);
    }
}

}  // namespace mvdb
