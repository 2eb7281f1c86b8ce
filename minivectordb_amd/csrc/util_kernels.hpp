// util_kernels.hpp — row normalisation, synthetic corpus generator, row gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvdb {

typedef float f32x4u __attribute__((ext_vector_type(4)));

// ---- faiss.normalize_L2 on device rows (reference: minivectordb/vector_database.py:45) --------
// One group of G lanes (runtime power of two <= 64) per row; pass 1 sums squares, pass 2 re-reads
// the row (L1/L2 hit: a row is <= 16 KiB) and scales it.  Zero-norm rows stay untouched, as in
// faiss fvec_renorm_L2 (`if (nr > 0)`).  HBM-bound: 1 read + 1 write of n*ld*4 bytes.
__global__ __launch_bounds__(256) void normalize_rows_kernel(float* __restrict__ X, int64_t n,
                                                             int64_t ld, int d4, int G) {
    const int lane = threadIdx.x & 63;
    const int t = lane & (G - 1);
    const int g = lane / G;
    const int rpi = 64 / G;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r0 = gw * rpi; r0 < n; r0 += nw * rpi) {
        const int64_t r = r0 + g;
        const bool rv = r < n;
        float* row = X + (rv ? r : n - 1) * ld;
        float nr = 0.f;
        for (int c = t; c < d4; c += G) {
            const f32x4u v = *reinterpret_cast<const f32x4u*>(row + c * 4);
            nr += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int m = G >> 1; m >= 1; m >>= 1) nr += __shfl_xor(nr, m);
        if (rv && nr > 0.f) {
            const float inorm = 1.0f / sqrtf(nr);
            for (int c = t; c < d4; c += G) {
                f32x4u v = *reinterpret_cast<const f32x4u*>(row + c * 4);
                v *= inorm;
                *reinterpret_cast<f32x4u*>(row + c * 4) = v;
            }
        }
    }
}

// max over rows of |row|^2 (raw adds: the split-precision batch pass needs a row-norm bound).  One
// wave per row; atomicMax on the bit pattern (non-negative floats order like their bits; a NaN row
// leaves a NaN pattern behind, which disables that pass).
__global__ __launch_bounds__(256) void max_row_norm2_kernel(const float* __restrict__ X, int64_t n, int64_t ld,
                                                            int d4, unsigned int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float best = 0.f;
    bool bad = false;
    for (int64_t r = gw; r < n; r += nw) {
        const float* row = X + r * ld;
        float nr = 0.f;
        for (int c = lane; c < d4; c += 64) {
            const f32x4u v = *reinterpret_cast<const f32x4u*>(row + c * 4);
            nr += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int m = 32; m >= 1; m >>= 1) nr += __shfl_xor(nr, m);
        bad |= !(nr >= 0.f);  // NaN
        best = fmaxf(best, nr);
    }
    if (lane == 0) atomicMax(out, bad ? 0x7FC00000u : __float_as_uint(best));
}

// min AND max over rows of |row|^2: out[0] = max (as above), out[1] = min (atomicMin on the bit pattern; the caller sets it
// to +inf first).  The L2 metric's certified batch passes nominate by inner product and need both ends of the norm range.
__global__ __launch_bounds__(256) void row_norm2_range_kernel(const float* __restrict__ X, int64_t n, int64_t ld,
                                                              int d4, unsigned int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float hi = 0.f, lo = INFINITY;
    bool bad = false;
    for (int64_t r = gw; r < n; r += nw) {
        const float* row = X + r * ld;
        float nr = 0.f;
        for (int c = lane; c < d4; c += 64) {
            const f32x4u v = *reinterpret_cast<const f32x4u*>(row + c * 4);
            nr += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int m = 32; m >= 1; m >>= 1) nr += __shfl_xor(nr, m);
        bad |= !(nr >= 0.f);  // NaN
        hi = fmaxf(hi, nr);
        lo = fminf(lo, nr);
    }
    if (lane == 0 && gw < n) {
        atomicMax(out, bad ? 0x7FC00000u : __float_as_uint(hi));
        atomicMin(out + 1, bad ? 0u : __float_as_uint(lo));
    }
}

// ---- synthetic stream -------------------------------------------------------------------------
// element(seed, i, j): a counter-based hash of (seed, i*d + j) split into four 16-bit uniforms,
// summed (Irwin-Hall, bell-shaped, zero mean) and scaled by an exact power of two.  Integer
// arithmetic + one exact int->float conversion + one exact scaling: host (oracle/synth.py) and
// device produce identical bits.
__host__ __device__ inline uint32_t pcg_hash32(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}
__host__ __device__ inline float synth_element(uint32_t seed_lo, uint32_t seed_hi, uint64_t ctr) {
    const uint32_t lo = (uint32_t)ctr, hi = (uint32_t)(ctr >> 32);
    const uint32_t h0 = pcg_hash32(lo ^ pcg_hash32(hi ^ pcg_hash32(seed_lo ^ pcg_hash32(seed_hi))));
    const uint32_t h1 = pcg_hash32(h0 ^ 0x9E3779B9u);
    const int32_t sum = (int32_t)((h0 & 0xFFFFu) + (h0 >> 16) + (h1 & 0xFFFFu) + (h1 >> 16));
    return (float)(sum - 131070) * (1.0f / 131072.0f);  // in (-1, 1), exact
}

// Three FAMILIES of rows, chosen by the seed's top byte (seed >> 56), all exact in fp32 on host and device alike
// (oracle/flat_oracle.c restates this function bit for bit):
//   0  zero-mean bell-shaped elements (above): the stream of every BASELINE config.  After normalisation the scores of a
//      query spread widely: the certified batch passes always certify on it.
//   1  all-positive rows, uniform in [0, 1) — what the reference's own tests store (numpy.random.rand:
//      tests/test_sharded_multithreaded_operations.py:22).  Normalised, every pair of rows has cosine ~0.75: a narrow cone.
//   2  clustered: row = centre[c] + noise, c = one of 4,096 centres SHARED by every seed (queries drawn from the family fall
//      beside the corpus' centres, as sentence embeddings of one domain do); noise = 2^-4 x a family-0 element (~6 % of the
//      centre's norm), except 1 row in 256 with none at all (EXACT duplicates of one another) and 1 in 256 at 2^-13 (near
//      duplicates).  Thousands of rows per centre score within 1e-3 of each other: certification fails by design.
constexpr uint64_t kSynthFamilyShift = 56;
constexpr uint32_t kSynthCentreSeed = 0xC3A5C85Cu;  // the centres' own stream (independent of the row seed)
__host__ __device__ inline float synth_value(uint64_t seed, uint64_t row, uint32_t col, uint32_t d) {
    const uint32_t slo = (uint32_t)seed, shi = (uint32_t)(seed >> 32);
    const uint32_t family = (uint32_t)(seed >> kSynthFamilyShift);
    const uint64_t ctr = row * (uint64_t)d + col;
    if (family == 0) return synth_element(slo, shi, ctr);
    if (family == 1) {
        const uint32_t lo = (uint32_t)ctr, hi = (uint32_t)(ctr >> 32);
        const uint32_t h = pcg_hash32(lo ^ pcg_hash32(hi ^ pcg_hash32(slo ^ pcg_hash32(shi))));
        return (float)(h >> 8) * (1.0f / 16777216.0f);  // 24 bits: exact, in [0, 1)
    }
    const uint32_t rlo = (uint32_t)row, rhi = (uint32_t)(row >> 32);
    const uint32_t hr = pcg_hash32(rlo ^ pcg_hash32(rhi ^ pcg_hash32(slo ^ pcg_hash32(shi ^ 0x5BD1E995u))));
    const uint32_t centre = hr & 4095u, kind = (hr >> 12) & 255u;
    const float c = synth_element(kSynthCentreSeed, 2u, (uint64_t)centre * d + col);
    if (kind == 0) return c;                                   // exact duplicate of every other noise-free row of the centre
    const float e = synth_element(slo, shi, ctr);
    return c + e * (kind == 1 ? 1.0f / 8192.0f : 1.0f / 16.0f);  // power-of-two scale: the product is exact, ONE rounding in the sum
}

__global__ __launch_bounds__(256) void synth_fill_kernel(float* __restrict__ X, int64_t n,
                                                         int64_t ld, int d, uint64_t seed,
                                                         int64_t first_row) {
    const int64_t d4 = ld / 4;
    const int64_t total = n * d4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d4;
        const int c = (int)(i - r * d4);
        const uint64_t row = (uint64_t)(first_row + r);
        f32x4u v;
        v.x = (c * 4 + 0 < d) ? synth_value(seed, row, c * 4 + 0, d) : 0.f;
        v.y = (c * 4 + 1 < d) ? synth_value(seed, row, c * 4 + 1, d) : 0.f;
        v.z = (c * 4 + 2 < d) ? synth_value(seed, row, c * 4 + 2, d) : 0.f;
        v.w = (c * 4 + 3 < d) ? synth_value(seed, row, c * 4 + 3, d) : 0.f;
        *reinterpret_cast<f32x4u*>(X + r * ld + c * 4) = v;
    }
}

// results of a compact re-run back into the rows of the queries they belong to: D[map[r], :] = Dt[r, :] (same for I)
__global__ __launch_bounds__(256) void scatter_results_kernel(const float* __restrict__ Dt, const int64_t* __restrict__ It,
                                                              const int64_t* __restrict__ map, int64_t nb, int k,
                                                              float* __restrict__ D, int64_t* __restrict__ I) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb * k) return;
    const int64_t r = i / k, c = i - r * k;
    D[map[r] * k + c] = Dt[i];
    I[map[r] * k + c] = It[i];
}

// ---- device-side bookkeeping of a certified batch pass (mvdb.hip: search_core) -----------------------------------------
// The host never reads the certification flags: one wave turns the per-query flags into the compact, ascending list of the
// queries that failed (map[0 .. nb)), publishes nb and adds the number of chunks that held one to the re-run counter.
__global__ __launch_bounds__(64) void split_plan_kernel(const int* __restrict__ chunk_flags, int nchunks,
                                                        const int* __restrict__ qfail, int nq, int64_t* __restrict__ map,
                                                        int* __restrict__ nb_out, unsigned long long* __restrict__ rerun_ctr,
                                                        unsigned int* __restrict__ fail_dev = nullptr,
                                                        volatile unsigned int* fail_host = nullptr,
                                                        unsigned int* __restrict__ any_dev = nullptr,
                                                        volatile unsigned int* any_host = nullptr) {
    const int lane = threadIdx.x;
    int bad = 0;
    for (int c = lane; c < nchunks; c += 64) bad += chunk_flags[c] != 0;
    for (int m = 32; m >= 1; m >>= 1) bad += __shfl_xor(bad, m);
    int run = 0;
    for (int base = 0; base < nq; base += 64) {
        const int i = base + lane;
        const bool f = i < nq && qfail[i] != 0;
        const unsigned long long mask = __ballot(f);
        if (f) map[run + __popcll(mask & ((1ull << lane) - 1ull))] = i;
        run += __popcll(mask);
    }
    if (lane == 0) {
        *nb_out = run;
        if (bad) atomicAdd(rerun_ctr, (unsigned long long)bad);
        // the opt-in single-query route: a running count of refused certificates, mirrored into a host-mapped word that the
        // routing reads WITHOUT synchronising (mvdb.hip: single_route_suspended)
        if (fail_dev && run) *fail_host = atomicAdd(fail_dev, (unsigned int)run) + (unsigned int)run;
        // every certified call: the same for the adaptive tile flags (mvdb.hip: tile_flags_wanted)
        if (any_dev && run) *any_host = atomicAdd(any_dev, (unsigned int)run) + (unsigned int)run;
    }
}

// dst[r, :] = src[map[r], :] for r < *nb; the other rows of [0, rows) (the gated exact kernels read whole query tiles)
// repeat the LAST failed query.  (Until round 5 they were zeros — and a zero query scores 0 against every row: once its list
// held k rows, every further row tied with the k-th score, passed the score gate and went through a wave-cooperative insert
// attempt that the 64-bit key order then refused: 10M attempts per padded slot.  A refused 32-query chunk at 10M x 512 cost
// 63 ms instead of the ~3.5 ms of one fp32-MFMA pass — found by benchmarks/scratch/refusal_probe.py on the clustered corpus.)
// floors (NULL, or one admission floor per ORIGINAL query) travel with the queries into floors_c (one per compact slot)
__global__ __launch_bounds__(256) void gather_failed_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                            const int64_t* __restrict__ map, const int* __restrict__ nb,
                                                            int64_t rows, int64_t ld, const float* __restrict__ floors,
                                                            float* __restrict__ floors_c) {
    const int64_t d4 = ld / 4;
    const int64_t total = rows * d4;
    const int64_t live = *nb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d4, c = i - r * d4;
        f32x4u v = {0.f, 0.f, 0.f, 0.f};
        const int64_t from = live > 0 ? map[r < live ? r : live - 1] : 0;
        if (live > 0) v = *reinterpret_cast<const f32x4u*>(src + from * ld + c * 4);
        *reinterpret_cast<f32x4u*>(dst + r * ld + c * 4) = v;
        if (c == 0 && floors_c) floors_c[r] = live > 0 && floors ? floors[from] : -INFINITY;
    }
}

// D[map[r], :] = Dt[r, :] (same for I) for r < *nb
__global__ __launch_bounds__(256) void scatter_failed_kernel(const float* __restrict__ Dt, const int64_t* __restrict__ It,
                                                             const int64_t* __restrict__ map, const int* __restrict__ nb, int64_t rows,
                                                             int k, float* __restrict__ D, int64_t* __restrict__ I) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * k) return;
    const int64_t r = i / k, c = i - r * k;
    if (r >= *nb) return;
    D[map[r] * k + c] = Dt[i];
    I[map[r] * k + c] = It[i];
}

// Row compaction after deletes, one chunk: dst[i, :] = src[r + lo(r), :] for the new rows r = r0 + i, i < rows, where lo(r) =
// the number of deleted rows at or below the old position = the smallest j in [0, m] with del[j] - j > r (del[] ascending,
// unique; del[j] - j is non-decreasing).  A wave per row: every lane runs the same search (one broadcast load per step),
// then the lanes move the row in 16-byte pieces.
__global__ __launch_bounds__(256) void gather_kept_rows_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                               const int64_t* __restrict__ del, int64_t m, int64_t r0,
                                                               int64_t rows, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t d4 = ld / 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < rows; i += (int64_t)gridDim.x * 4) {
        const int64_t r = r0 + i;
        int64_t lo = 0, hi = m;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (del[mid] - mid > r)
                hi = mid;
            else
                lo = mid + 1;
        }
        const f32x4u* s = reinterpret_cast<const f32x4u*>(src + (r + lo) * ld);
        f32x4u* t = reinterpret_cast<f32x4u*>(dst + i * ld);
        for (int64_t c = lane; c < d4; c += 64) t[c] = s[c];
    }
}

// ---- row compaction in ONE pass for a handful of deleted rows (round 6) ---------------------------------------------------
// The staging path above moves every byte of the tail twice (gather into the buffer, copy back).  For the reference's own
// delete — ONE row (`delete_embedding`, vector_database.py:119) — or any few rows, the tail is shifted in place instead, in
// 16-byte units: new unit u of the tail comes from old unit u + lo(row of u) x (units per row), a source that always lies
// ABOVE its destination by at most m rows.  Every workgroup owns one contiguous range of the new tail and walks it upwards in
// slices of 256 x U units: load the slice's sources into registers, barrier (the slice's destinations overlap its own
// sources), store.  A later slice's sources lie above everything stored so far, so slices need no other ordering.  The one
// cross-workgroup hazard — the sources of the last rows of a range lie in the NEXT range, which its owner overwrites at its own
// pace — is removed before the launch: shift_save_kernel copies the first m rows' worth of units of every range but the first
// into `side`, and sources at or above the owner's upper boundary are read from there.  No workgroup ever waits for another.
constexpr int kShiftUnits = 8;  // 16-byte units per thread and slice (32 KiB per workgroup and slice)

__global__ __launch_bounds__(256) void shift_save_kernel(f32x4u* __restrict__ side, const f32x4u* __restrict__ tail, int64_t range,
                                                         int64_t side_units, int64_t old_units) {
    // side[b][j] = tail[(b + 1) range + j], j < side_units (clamped to the old tail): boundary b = blockIdx.y
    const int64_t base = ((int64_t)blockIdx.y + 1) * range;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < side_units; j += (int64_t)gridDim.x * 256)
        if (base + j < old_units) side[(int64_t)blockIdx.y * side_units + j] = tail[base + j];
}

template <bool UNIFORM>
__global__ __launch_bounds__(256) void shift_rows_kernel(f32x4u* tail, const f32x4u* __restrict__ side,
                                                         const int64_t* __restrict__ del, int64_t m, int64_t new_units,
                                                         int64_t range, int64_t rowunits, int64_t side_units) {
    constexpr int U = kShiftUnits;
    const int64_t a = (int64_t)blockIdx.x * range;
    const int64_t b = a + range < new_units ? a + range : new_units;
    const bool has_next = blockIdx.x + 1 < gridDim.x;
    const int64_t bound = a + range;  // the next range's first unit: sources from here on come from `side`
    const f32x4u* myside = side + (int64_t)blockIdx.x * side_units;
    auto shift_of = [&](int64_t r) {  // smallest j with del[j] - j > r (gather_kept_rows_kernel's rule)
        int64_t l = 0, h = m;
        while (l < h) {
            const int64_t mid = (l + h) >> 1;
            if (del[mid] - mid > r)
                h = mid;
            else
                l = mid + 1;
        }
        return l;
    };
    for (int64_t p = a; p < b; p += 256 * U) {
        f32x4u v[U];
        // the shift is non-decreasing along the tail: equal at both ends of the slice (all but m slices) = one value for the slice;
        // UNIFORM: one run of m deleted rows at the head of the tail, every kept row moves up by m
        int64_t lo0 = m, lo1 = m;
        if (!UNIFORM) {
            const int64_t pe = (p + 256 * U < b ? p + 256 * U : b) - 1;
            lo0 = shift_of(p / rowunits);
            lo1 = shift_of(pe / rowunits);
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int64_t u = p + i * 256 + threadIdx.x;
            if (u < b) {
                const int64_t lo = lo0 == lo1 ? lo0 : shift_of(u / rowunits);
                const int64_t s = u + lo * rowunits;
                v[i] = (has_next && s >= bound) ? myside[s - bound] : tail[s];
            }
        }
        __syncthreads();  // (s_waitcnt vmcnt(0) + barrier: every source of the slice is in registers)
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int64_t u = p + i * 256 + threadIdx.x;
            if (u < b) tail[u] = v[i];
        }
    }
}

// dense [n,d] -> padded [n,ld] copy (device to device), zero padding
__global__ __launch_bounds__(256) void pad_rows_kernel(float* __restrict__ dst,
                                                       const float* __restrict__ src, int64_t n,
                                                       int d, int64_t ld) {
    const int64_t total = n * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        dst[i] = c < d ? src[r * d + c] : 0.f;
    }
}

}  // namespace mvdb
