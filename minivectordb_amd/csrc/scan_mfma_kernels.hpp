// scan_mfma_kernels.hpp — multi-query flat scan on the exact-fp32 matrix cores.
//
// One pass over the corpus serves up to 32 queries: scores[query, row] = Q · Xᵀ is a skinny GEMM
// (M = queries, N = corpus rows, K = d).  faiss.IndexFlatIP.search with nq queries
// (reference call site minivectordb/vector_database.py:497 is nq = 1; a batch API has no
// reference counterpart — results per query are identical to nq separate searches).
//
// Roofline: HBM up to nq ~ 39 (fp32 MFMA 157.3 TF / 8 TB/s = 19.7 FLOP/B; intensity = nq/2
// FLOP/B).  Algorithmic bytes per launch = n * ld * 4 (the corpus once, for all queries).
//
// v_mfma_f32_16x16x4_f32:  D[16x16] += A[16x4] · B[4x16], exact fp32 (an fmaf chain in k order).
//   A = queries      lane l holds A[i = l&15][k = l>>4]
//   B = corpus rows  lane l holds B[k = l>>4][j = l&15] = X[row0 + (l&15)][k]
//   D                lane l holds D[i = 4*(l>>4) + r][j = l&15], r = 0..3:
//                    the scores of corpus row (l&15) against queries 4*(l>>4) .. +3
// Corpus operand straight from HBM into VGPRs: lane l loads ONE float4 =
// X[row0 + (l&15)][16*kb + 4*(l>>4) .. +3]; its 4 components feed 4 MFMAs, the j-th of which
// contracts k in {16kb + j, +4, +8, +12}.  The query fragments are pre-arranged in LDS in exactly
// that order ("fragment-major", lane-linear => conflict-free ds_read_b32), so no lane ever
// shuffles.  One load instruction covers 16 rows x 64 contiguous bytes; consecutive instructions of
// a wave walk the same 16 rows along k, so every 128-B line is consumed within two instructions.
//
// Selection: per wave and query a sorted k-list lives in LDS (rarely touched); lanes gate their
// 4 scores against per-query thresholds held in registers; inserts are wave-cooperative.
#pragma once
#include "scan_kernels.hpp"

namespace mvdb {

typedef float f32x4m __attribute__((ext_vector_type(4)));

struct MfmaScanArgs {
    const float* X;
    int64_t n;
    int64_t ld;
    const float* q;  // [nq, ld] (already normalised if requested)
    int nq;          // <= 16 * NG
    int k;           // <= kMaxFusedK
    uint64_t* cand;  // [nq, gridDim.x, k]
    const uint32_t* mask = nullptr;  // NULL, or one bit per row: only rows whose bit is set are offered to the lists
    const float* thr0 = nullptr;     // NULL, or [nq] admission floors: a row scoring below its query's floor is never offered (the
                                     // re-run of a refused query starts from the k-th exact score of its nominees, less a rounding
                                     // margin: without it every wave's list fills from -inf — k ln(rows per wave / k) inserts per query
                                     // and wave, 2.5 per ROW at 1M rows x 32 queries, where the pass took 0.82 ms instead of 0.4)
};

// Offer one 16-row tile's scores to the per-wave, per-query LDS lists.  acc[g][r] of lane l is the score
// of corpus row (tile*16 + (l&15)) against query g*16 + 4*(l>>4) + r.  thr[g][r] gates (score only);
// the exact 64-bit order is decided by the insert.  Wave-uniform control flow.
template <int NG>
__device__ __forceinline__ void mfma_tile_select(const f32x4m (&acc)[NG], float (&thr)[NG][4], uint32_t (&thr_row)[NG][4], bool rvalid,
                                                 uint32_t rowid, int nq, int k, uint64_t* mylists, int lane, const float (&floor0)[NG][4]) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = acc[g][r];
            uint64_t mask = __ballot(rvalid && beats_key(s, rowid, thr[g][r], thr_row[g][r]));
            while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int qq = g * 16 + 4 * (src >> 4) + r;
                if (qq >= nq) continue;  // padded query slot
                const float sv = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(s), src));
                const uint32_t rv = (uint32_t)__builtin_amdgcn_readlane((int)rowid, src);
                const uint64_t kth = lds_list_insert(mylists + (size_t)qq * k, k, make_key(sv, rv), lane);
                if ((lane >> 4) == (src >> 4)) set_threshold(kth, floor0[g][r], thr[g][r], thr_row[g][r]);
            }
        }
    }
}

// After the scan: wave w merges, for queries w, w+4, ..., the lists of all waves and writes the block's list.
template <int NG>
__device__ __forceinline__ void mfma_block_merge(const uint64_t* lists, int nq, int k, uint64_t* cand, int lane,
                                                 int wave) {
    for (int qq = wave; qq < nq; qq += kScanWaves) {
        WaveTopK tk;
        tk.init(k);
#pragma unroll 1
        for (int w = 0; w < kScanWaves; ++w) {
            const uint64_t* l = lists + ((size_t)w * NG * 16 + qq) * k;
            tk.offer(lane < k ? l[lane] : 0ull);
        }
        if (lane < k) cand[((int64_t)qq * gridDim.x + blockIdx.x) * k + lane] = tk.key;
    }
}

template <int KB, int NG>
__device__ __forceinline__ void flat_scan_mfma_body(const MfmaScanArgs& a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Qf = reinterpret_cast<float*>(smem);                                   // [NG][KB*4][64]
    uint64_t* lists = reinterpret_cast<uint64_t*>(smem + (size_t)NG * KB * 4 * 64 * 4);  // [4][NG*16][k]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = a.k;

    // ---- queries -> LDS, fragment-major: Qf[g][kb*4 + j][l] = Q[g*16 + (l&15)][16kb + 4(l>>4) + j]
    for (int e = threadIdx.x; e < NG * KB * 4 * 64; e += kScanThreads) {
        const int l = e & 63, f = (e >> 6) % (KB * 4), g = e / (KB * 4 * 64);
        const int kb = f >> 2, j = f & 3;
        const int qi = g * 16 + (l & 15);
        Qf[e] = qi < a.nq ? a.q[(int64_t)qi * a.ld + 16 * kb + 4 * (l >> 4) + j] : 0.f;
    }
    uint64_t* mylists = lists + (size_t)wave * NG * 16 * k;
    for (int e = lane; e < NG * 16 * k; e += 64) mylists[e] = 0ull;
    __syncthreads();

    float thr[NG][4], floor0[NG][4];
    uint32_t thr_row[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // padded query slots never pass the gate
            thr[g][r] = (g * 16 + 4 * (lane >> 4) + r) < a.nq ? -INFINITY : INFINITY;
            thr_row[g][r] = 0u;
            floor0[g][r] = -INFINITY;
        }

    const int64_t ntiles = (a.n + 15) / 16;
    const int64_t nwaves_total = (int64_t)gridDim.x * kScanWaves;
    const int64_t last = a.n - 1;
    constexpr int PD = KB < 8 ? KB : 8;  // loads kept in flight ahead of the MFMAs (x 1 KiB each)

    for (int64_t tile = (int64_t)blockIdx.x * kScanWaves + wave; tile < ntiles; tile += nwaves_total) {
        int64_t row = tile * 16 + (lane & 15);
        const bool rvalid = row <= last;
        row = rvalid ? row : last;
        const f32x4m* p = reinterpret_cast<const f32x4m*>(a.X + row * a.ld + 4 * (lane >> 4));
        f32x4m x[KB];
#pragma unroll
        for (int kb = 0; kb < PD; ++kb) x[kb] = __builtin_nontemporal_load(p + kb * 4);
        f32x4m acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = f32x4m{0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb + PD < KB) x[kb + PD] = __builtin_nontemporal_load(p + (kb + PD) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float qa = Qf[(g * KB * 4 + kb * 4 + j) * 64 + lane];
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, x[kb][j], acc[g], 0, 0, 0);
                }
            }
        }
        mfma_tile_select<NG>(acc, thr, thr_row, rvalid, (uint32_t)(tile * 16 + (lane & 15)), a.nq, k, mylists, lane, floor0);
    }

    __syncthreads();
    mfma_block_merge<NG>(lists, a.nq, k, a.cand, lane, wave);
}

template <int KB, int NG>
__global__ __launch_bounds__(kScanThreads) void flat_scan_mfma_kernel(MfmaScanArgs a) {
    flat_scan_mfma_body<KB, NG>(a);
}
// enabled by a device-side count (the whole pass runs iff *gate > gate_lo): see flat_scan_gated_kernel
template <int KB, int NG>
__global__ __launch_bounds__(kScanThreads) void flat_scan_mfma_gated_kernel(MfmaScanArgs a, const int* __restrict__ gate, int gate_lo) {
    if (*gate <= gate_lo) return;
    flat_scan_mfma_body<KB, NG>(a);
}

}  // namespace mvdb

// =================================================================================================
// v2: corpus tiles streamed COALESCED into LDS by LDS-DMA, fragments read back conflict-free
// =================================================================================================
// The v1 kernel above feeds the MFMA B operand with fragment-shaped global loads (16 rows x 64 B per
// instruction) and reaches ~5.7 TB/s at 10M x 512.  Here every wave streams its 16-row tile in
// stages of SK = 128 floats per row with `global_load_lds_dwordx4` (LDS-DMA, no VGPR staging): one
// instruction moves 2 rows x 512 contiguous bytes, 8 instructions one 8-KiB stage, double buffered
// per wave behind a counted `s_waitcnt vmcnt(8)` (the next stage stays in flight).  The LDS image is
// lane-linear, so the XOR swizzle that makes the fragment reads conflict-free is applied to the
// per-lane SOURCE address (chunk p of row r holds global chunk p ^ (r & 15)), and again on the
// `ds_read_b128` side.  Query fragments live in registers (KB float4 per lane).
namespace mvdb {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// SKB = 16-float k-blocks per row and stage: 8 -> 512 B per row, 8-KiB stage, a DMA instruction moves
// 2 rows x 512 B; 16 -> 1 KiB per row, 16-KiB stage, a DMA instruction moves ONE row's contiguous KiB
// (the GEMV kernel's access shape).
constexpr int mfma2_stage_bytes(int skb) { return 16 * skb * 64; }
constexpr int mfma2_wave_lds_bytes(int skb) { return 2 * mfma2_stage_bytes(skb); }

template <int KB, int NG, int SKB, bool MASKED = false, int METRIC = 0>
__device__ __forceinline__ void flat_scan_mfma2_body(const MfmaScanArgs& a) {
    static_assert(KB % SKB == 0 && (SKB == 8 || SKB == 16), "d must be a multiple of the stage depth");
    constexpr int NS = KB / SKB;  // stages per tile
    constexpr int kStageFloats = SKB * 16, kStageBytes = mfma2_stage_bytes(SKB), kWaveLdsBytes = mfma2_wave_lds_bytes(SKB);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = a.k;
    unsigned char* wbuf = smem + (size_t)wave * kWaveLdsBytes;
    uint64_t* lists = reinterpret_cast<uint64_t*>(smem + (size_t)kScanWaves * kWaveLdsBytes);  // [4][NG*16][k]
    uint64_t* mylists = lists + (size_t)wave * NG * 16 * k;
    for (int e = lane; e < NG * 16 * k; e += 64) mylists[e] = 0ull;

    // ---- query fragments -> registers: qa[kb] = Q[l&15][16kb + 4(l>>4) .. +3] -----------------------
    f32x4m qa[NG][KB];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int qi = g * 16 + (lane & 15);
        const float* qp = a.q + (int64_t)qi * a.ld + 4 * (lane >> 4);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            qa[g][kb] = qi < a.nq ? *reinterpret_cast<const f32x4m*>(qp + 16 * kb) : f32x4m{0, 0, 0, 0};
    }
    float thr[NG][4], floor0[NG][4];
    uint32_t thr_row[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = g * 16 + 4 * (lane >> 4) + r;
            floor0[g][r] = qi < a.nq && a.thr0 ? a.thr0[qi] : -INFINITY;   // (a floor carries row 0: nothing ties its way past it)
            thr[g][r] = qi < a.nq ? floor0[g][r] : INFINITY;
            thr_row[g][r] = 0u;
        }
    // METRIC 1 (squared L2 by |q|^2 + |x|^2 - 2 q.x; the lists keep -distance as everywhere): |q|^2 of the four queries
    // whose scores this lane's accumulator registers hold — query g 16 + 4 (lane >> 4) + r
    float qn[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float qs = 0.f;
        if (METRIC == 1) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int j = 0; j < 4; ++j) qs = fmaf(qa[g][kb][j], qa[g][kb][j], qs);
            qs += __shfl_xor(qs, 16);
            qs += __shfl_xor(qs, 32);  // |q|^2 of query g 16 + (lane & 15), on all four k-group lanes
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) qn[g][r] = METRIC == 1 ? __shfl(qs, 4 * (lane >> 4) + r) : 0.f;
    }

    const int64_t ntiles = (a.n + 15) / 16;
    const int64_t nwaves_total = (int64_t)gridDim.x * kScanWaves;
    const int64_t last = a.n - 1;
    // DMA lane roles.  SKB = 8: instruction i (0..7) moves rows 2i and 2i+1; lane ln -> row 2i + (ln>>5),
    // LDS chunk position p = ln & 31.  SKB = 16: instruction i (0..15) moves row i, p = ln.  Position p of
    // row r must receive global chunk p ^ (r & 15).
    const int dma_rsel = SKB == 8 ? lane >> 5 : 0, dma_p = SKB == 8 ? (lane & 31) : lane;
    // fragment read: row (l&15), k-group (l>>4)
    const int fr = lane & 15, fkg = lane >> 4;

    auto issue_stage = [&](int64_t tile, int ks, int buf) {
#pragma unroll
        for (int i = 0; i < SKB; ++i) {
            const int r = (SKB == 8 ? 2 * i : i) + dma_rsel;
            int64_t row = tile * 16 + r;
            row = row <= last ? row : last;
            const float* src = a.X + row * a.ld + ks * kStageFloats + 4 * (dma_p ^ (r & 15));
            unsigned char* dst = wbuf + buf * kStageBytes + i * 1024;  // wave-uniform base; lane*16 implied
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 2 /* nt */);
        }
    };

    // Stage pipeline (per wave, private buffers, no block barrier): stage s = (tile, ks).
    //   wait until stage s has landed (stage s+1's 8 DMAs may stay in flight: vmcnt(8))
    //   read its 8 B fragments LDS -> registers, wait for them (lgkmcnt(0))
    //   -> buffer (s & 1) is free again: issue the DMAs of stage s+2 into it
    //   64*NG... MFMAs on the fragments (the DMA issue slots hide among them)
    // The sched_barriers pin that order: hipcc otherwise hoists LDS-DMA instructions into the reads of
    // the buffer they overwrite (seen in the ISA; nondeterministic wrong scores).
    int64_t tile = (int64_t)blockIdx.x * kScanWaves + wave;
    unsigned cnt = 0;  // stages consumed so far (buffer = cnt & 1)
    auto stage_tile = [&](int64_t t, int ks_abs) { return t + (int64_t)(ks_abs / NS) * nwaves_total; };
    if (tile < ntiles) {
        issue_stage(tile, 0, 0);
        const int64_t t1 = stage_tile(tile, 1);
        if (t1 < ntiles) issue_stage(t1, 1 % NS, 1);
    }
    while (tile < ntiles) {
        const int64_t next_tile = tile + nwaves_total;
        uint32_t mw = 0xffffffffu;  // MASKED: the bitmap word holding this tile's 16 rows, requested before the tile's stages
        if (MASKED) mw = a.mask[tile >> 1];
        f32x4m acc0[NG], acc1[NG];
        float xs = 0.f;  // METRIC 1: this lane's share of |x|^2 of row tile 16 + (lane & 15)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc0[g] = acc1[g] = f32x4m{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            const int buf = cnt & 1;
            const int64_t t1 = stage_tile(tile, ks + 1), t2 = stage_tile(tile, ks + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (t1 < ntiles) {  // this stage landed, the SKB DMAs of stage s+1 still in flight
                if (SKB == 8)
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* sb = wbuf + buf * kStageBytes + fr * (kStageFloats * 4);
            f32x4m xb[SKB];
#pragma unroll
            for (int kbl = 0; kbl < SKB; ++kbl)
                xb[kbl] = *reinterpret_cast<const f32x4m*>(sb + (((4 * kbl + fkg) ^ fr) << 4));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (t2 < ntiles) issue_stage(t2, (ks + 2) % NS, buf);  // refill the buffer just drained
            if (METRIC == 1) {
#pragma unroll
                for (int kbl = 0; kbl < SKB; ++kbl)
#pragma unroll
                    for (int j = 0; j < 4; ++j) xs = fmaf(xb[kbl][j], xb[kbl][j], xs);
            }
#pragma unroll
            for (int kbl = 0; kbl < SKB; ++kbl) {
                const int kb = ks * SKB + kbl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (kbl & 1)
                            acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[g][kb][j], xb[kbl][j], acc1[g], 0, 0, 0);
                        else
                            acc0[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[g][kb][j], xb[kbl][j], acc0[g], 0, 0, 0);
                    }
                }
            }
            ++cnt;
        }
        f32x4m acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = acc0[g] + acc1[g];
        if (METRIC == 1) {
            xs += __shfl_xor(xs, 16);
            xs += __shfl_xor(xs, 32);  // |x|^2 of this lane's row
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[g][r] = (2.f * acc[g][r] - xs) - qn[g][r];  // -(|q|^2 + |x|^2 - 2 q.x)
        }
        mfma_tile_select<NG>(acc, thr, thr_row, tile * 16 + fr <= last && ((mw >> ((int)(tile & 1) * 16 + fr)) & 1u), (uint32_t)(tile * 16 + fr), a.nq, k,
                             mylists, lane, floor0);
        tile = next_tile;
    }

    __syncthreads();
    mfma_block_merge<NG>(lists, a.nq, k, a.cand, lane, wave);
}

template <int KB, int NG, int SKB, int METRIC = 0>
__global__ __launch_bounds__(kScanThreads) void flat_scan_mfma2_kernel(MfmaScanArgs a) {
    flat_scan_mfma2_body<KB, NG, SKB, false, METRIC>(a);
}
// enabled by a device-side count (the whole pass runs iff *gate > gate_lo; gate == NULL: always): see flat_scan_gated_kernel.
// MASKED: only rows whose bit is set in a.mask are offered to the lists (bitmap-selected batches and their exact re-runs).
template <int KB, int NG, int SKB, bool MASKED = false>
// npasses > 1: ONE launch walks the passes of a compact batch of `rtot` queries, per_pass at a time (pass p: queries
// [p per_pass, ...), floors and lists likewise, lists cand_stride keys apart) — eight gated launches that return at once cost a
// certified 256-query call ~60 us of device time, 10 % of the call at 1M rows.
__global__ __launch_bounds__(kScanThreads) void flat_scan_mfma2_gated_kernel(MfmaScanArgs a, const int* __restrict__ gate, int gate_lo,
                                                                             const int* __restrict__ need, int npasses, int per_pass,
                                                                             int rtot, int64_t cand_stride) {
    for (int p = 0; p < npasses; ++p) {
        if (gate && *gate <= gate_lo + p * per_pass) return;
        if (need && need[p] == 0) continue;  // every refused query of this pass was answered by the rescue pass (half_scan.hip)
        MfmaScanArgs b = a;
        b.q = a.q + (int64_t)p * per_pass * a.ld;
        b.nq = min(per_pass, rtot - p * per_pass);
        b.cand = a.cand + (int64_t)p * cand_stride;
        if (a.thr0) b.thr0 = a.thr0 + p * per_pass;
        flat_scan_mfma2_body<KB, NG, SKB, MASKED>(b);
        __syncthreads();  // the lists are re-initialised by the next pass
    }
}

}  // namespace mvdb

// =================================================================================================
// large batches (nq >= 64): compute-bound tiled GEMM + in-register top-k gate
// =================================================================================================
// Beyond ~39 queries per pass the scan leaves the HBM roofline for the fp32 MFMA one (157.3 TFLOP/s),
// so the corpus is tiled like a GEMM: block tile = 128 corpus rows x 128 queries, K streamed in
// 16-deep steps through LDS (k-major tiles, conflict-free ds_read_b32 fragments, register-staged
// double buffering — the encoder's GEMM main loop), v_mfma_f32_32x32x2_f32 (exact fp32).  The
// accumulator tile D[row][query] keeps the QUERY on the lane (col = lane&31), so each lane gates its
// 64 scores against two per-query thresholds held in registers; the rare survivors are inserted
// wave-cooperatively into per-wave, per-query sorted lists in LDS (k <= 16).  Corpus traffic =
// ceil(nq/128) passes; algorithmic FLOPs per launch = 2 * n * d * 128.
namespace mvdb {

typedef float f32x16m __attribute__((ext_vector_type(16)));
constexpr int kGemmScanMaxK = 16;

struct GemmScanArgs {
    const float* X;
    int64_t n;
    int64_t ld;
    int K;           // = d (multiple of 16)
    const float* q;  // [nq, ld]
    int nq;
    int k;           // <= kGemmScanMaxK
    uint64_t* cand;  // [nq, gridDim.x, k]
};

__device__ __forceinline__ void flat_scan_gemm_body(const GemmScanArgs& a) {
    constexpr int BM = 128, BN = 128, BK = 16, LD = BM + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tiles = reinterpret_cast<float*>(smem);                                   // [2][A|B][BK][LD]
    uint64_t* lists = reinterpret_cast<uint64_t*>(smem + 2 * 2 * BK * LD * 4);       // [4 waves][64 queries][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fk = lane >> 5;
    const int k = a.k;
    const int n0 = blockIdx.y * BN;  // first query of this block's query tile
    uint64_t* mylists = lists + (size_t)wave * 64 * k;
    for (int e = lane; e < 64 * k; e += 64) mylists[e] = 0ull;

    // thresholds of this lane's two queries (query = n0 + wn*64 + j*32 + fr)
    float thr[2];
    uint32_t thr_row[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 2; ++j) thr[j] = (n0 + wn * 64 + j * 32 + fr) < a.nq ? -INFINITY : INFINITY;

    // staging roles (as the encoder GEMM): thread -> rows lr, lr+64 of each tile, 4 consecutive k
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const float* w_ptr[2];
    bool w_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int qrow = n0 + lr + i * 64;
        w_ok[i] = qrow < a.nq;
        w_ptr[i] = a.q + (int64_t)(w_ok[i] ? qrow : 0) * a.ld + lk;
    }
    const int64_t ntiles = (a.n + BM - 1) / BM;
    const int nk = a.K / BK;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t m0 = tile * BM;
        const float* a_ptr[2];
        bool a_ok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t row = m0 + lr + i * 64;
            a_ok[i] = row < a.n;
            a_ptr[i] = a.X + (a_ok[i] ? row : 0) * a.ld + lk;
        }
        f32x4m ra[2], rw[2];
        auto stage_load = [&](int k0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[i] = a_ok[i] ? __builtin_nontemporal_load(reinterpret_cast<const f32x4m*>(a_ptr[i] + k0))
                                : f32x4m{0, 0, 0, 0};
                rw[i] = w_ok[i] ? *reinterpret_cast<const f32x4m*>(w_ptr[i] + k0) : f32x4m{0, 0, 0, 0};
            }
        };
        auto stage_write = [&](int buf) {
            float* As = tiles + buf * (2 * BK * LD);
            float* Bs = As + BK * LD;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = lr + i * 64;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    As[(lk + j) * LD + r] = ra[i][j];
                    Bs[(lk + j) * LD + r] = rw[i][j];
                }
            }
        };
        f32x16m acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        __syncthreads();  // previous tile's last LDS reads are done before buffer 0 is overwritten
        stage_load(0);
        stage_write(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) stage_load((kt + 1) * BK);
            const float* As = tiles + buf * (2 * BK * LD);
            const float* Bs = As + BK * LD;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float av[2], bv[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    av[i] = As[(kk + fk) * LD + wm * 64 + i * 32 + fr];  // A[i = row fr][k = fk]
                    bv[i] = Bs[(kk + fk) * LD + wn * 64 + i * 32 + fr];  // B[k = fk][j = query fr]
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            if (kt + 1 < nk) {
                stage_write(buf ^ 1);
                __syncthreads();
            }
        }
        // ---- selection: D[row][query], query on the lane -------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ql = j * 32 + fr;  // this lane's query inside the wave's 64
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const float s = acc[i][j][r];
                    uint64_t mask = __ballot(row < a.n && beats_key(s, (uint32_t)row, thr[j], thr_row[j]));
                    while (mask) {
                        const int src = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        const int sq = j * 32 + (src & 31);
                        const float sv = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(s), src));
                        const uint32_t rv = (uint32_t)(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (src >> 5));
                        const uint64_t kth = lds_list_insert(mylists + (size_t)sq * k, k, make_key(sv, rv), lane);
                        if (ql == sq) set_threshold(kth, -INFINITY, thr[j], thr_row[j]);  // both lane halves of that query
                    }
                }
            }
        }
    }

    // ---- block merge: query column c (0..127) has one list in each of the two waves with wn = c / 64 ------
    __syncthreads();
    for (int c = wave; c < BN; c += 4) {
        const int qq = n0 + c;
        if (qq >= a.nq) continue;
        WaveTopK tk;
        tk.init(k);
#pragma unroll 1
        for (int m = 0; m < 2; ++m) {
            const uint64_t* l = lists + ((size_t)(m * 2 + c / 64) * 64 + (c % 64)) * k;
            tk.offer(lane < k ? l[lane] : 0ull);
        }
        if (lane < k) a.cand[((int64_t)qq * gridDim.x + blockIdx.x) * k + lane] = tk.key;
    }
}

__global__ __launch_bounds__(256) void flat_scan_gemm_kernel(GemmScanArgs a) { flat_scan_gemm_body(a); }
// enabled by a device-side count (the whole launch runs iff *gate > gate_lo): see flat_scan_gated_kernel
__global__ __launch_bounds__(256) void flat_scan_gemm_gated_kernel(GemmScanArgs a, const int* __restrict__ gate, int gate_lo) {
    if (*gate <= gate_lo) return;
    flat_scan_gemm_body(a);
}

}  // namespace mvdb
