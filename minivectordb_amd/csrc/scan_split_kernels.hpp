// scan_split_kernels.hpp — query batches (>= 14 per call) on the bf16 matrix cores, results certified exact.
//
// The fp32 MFMA rate (157 TFLOP/s) caps a 128-query corpus pass at ~13 ms on 10M x 512.  The bf16
// MFMA rate is 16x higher, so the pass is run as a SPLIT-PRECISION product instead:
//
//      x = xh + xl + rx,  q = qh + ql + rq      (xh, xl, qh, ql bf16 by RNE; |r| <= 2^-16 |.|)
//      q.x ~ qh.xh + qh.xl + ql.xh              (three v_mfma_f32_32x32x16_bf16 per tile, fp32 accumulate)
//
// whose operand error is bounded by 3 * 2^-16 * sum|q_i x_i| <= 4.6e-5 * |q| * |x| (Cauchy-Schwarz).
// The corpus stays fp32 in HBM (one copy, shared with the single-query scan); rows are split into
// bf16 pairs in registers while they are staged into LDS, queries are split once per call.
//
// The approximate scores only NOMINATE: each query keeps its best kSplitKeep = 16 rows, the
// nominees are re-scored in exact fp32 (`split_certify_kernel`) and the final top-k (k <= 12) is
// taken from those exact scores.  The result is certified per query by
//
//      exact_score(k-th) > approx_score(16th nominee) + eps * |q| * max|x|
//
// — every row that was not nominated has an approximate score <= the 16th nominee's, hence a true
// score below the k-th result.  A query that fails the test (more than 16 - k rows within eps of
// its k-th score: duplicate-heavy corpora) raises a flag and its chunk is re-run on the exact fp32
// kernels (mvdb.hip: search_core), so the path never returns an uncertified answer.
//
// Kernels of a pass (host side: mvdb.hip, launch_split_scan):
//   split_queries_kernel        q -> K-step-major bf16 (hi, lo) images + |q|
//   flat_scan_split_kernel      33..128 queries: block = 128 rows x 128 queries, 8 waves (2 per SIMD), wave tile
//                               32 rows x 64 queries, K streamed 32 deep through a 3-stage LDS-DMA ring (raw fp32
//                               rows + query images, bank swizzle on the DMA source), one bare s_barrier per step,
//                               MFMAs of step g-1 and the hi/lo split of step g in one scheduling region.
//                               <0, true> = the SEED launch every pass starts with (first min(tiles, CUs) tiles,
//                               one per block, scores dumped and merged instead of inserted)
//   flat_scan_split32_kernel    14..32 queries: per-wave 32-row tiles through private LDS-DMA rings (no block
//                               barriers), query fragments in registers; HBM-bound
//   split_seed_kernel           merges the seed launch's per-block lists, publishes the admission floors
//   split_certify_kernel        merges the nominees, exact fp32 re-scores, top-k, certificate
// In all of them D[row][query] keeps the query on the lane: one threshold register per lane gates the scores,
// survivors are inserted wave-cooperatively into per-wave, per-query sorted LDS lists.
#pragma once
#include "scan_mfma_kernels.hpp"

namespace mvdb {

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8m __attribute__((ext_vector_type(8)));

constexpr int kSplitKeep = 16;   // nominees per query
constexpr int kSplitMaxK = 12;   // results per query this path certifies
constexpr int kSplitThreads = 512;
constexpr int kSplitStages = 3;                          // LDS ring depth (K-steps)
constexpr int kSplitStageBytes = 32768;                  // 128 rows x 32 fp32 + 128 queries x 32 x (bf16 hi + lo)
constexpr size_t kSplitLds = (size_t)kSplitStages * kSplitStageBytes;  // dynamic part (the ring); the lists are static

struct SplitScanArgs {
    const float* X;
    int64_t n;
    int64_t ld;
    int K;              // = d, multiple of 32
    const __bf16* qh;   // [K / 32][128][32] K-step-major, rows >= nq zero (one 128-query chunk per launch)
    const __bf16* ql;
    int nq;
    uint64_t* cand;     // [nq, gridDim.x, kSplitKeep]
    int64_t tile0;      // this launch scans the 128-row tiles [tile0, tile1)
    int64_t tile1;
    const float* thr0;  // [nq] admission floor per query (seed pass, see launch_split_scan) or NULL
    unsigned int* stats; // NULL, or [2]: list inserts, tiles that reached the slow path (diagnostics)
};

typedef __bf16 sbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2m __attribute__((ext_vector_type(2)));
// (lo 16 bits: bf16(a), hi 16 bits: bf16(b)), RNE — one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    union { sbf16x2 v; uint32_t u; } c;
    c.v = __builtin_convertvector(f32x2m{a, b}, sbf16x2);
    return c.u;
}

// q -> (qh, ql) + |q|; rows >= nq of the 128-padded arrays are zero-filled
__global__ __launch_bounds__(256) void split_queries_kernel(const float* __restrict__ q, int64_t ld, int d, int nq,
                                                            __bf16* __restrict__ qh, __bf16* __restrict__ ql,
                                                            float* __restrict__ qnorm) {
    const int row = blockIdx.x;
    float ss = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) {
        const float v = row < nq ? q[(int64_t)row * ld + c] : 0.f;
        const __bf16 h = (__bf16)v;
        // K-step-major image [d / 32][128 queries][32]: the 16 KiB a K-step stages are contiguous (every block
        // of the scan reads the same lines at about the same time; a row-major image puts them 2*d bytes apart)
        const int64_t o = ((int64_t)(c >> 5) * 128 + row) * 32 + (c & 31);
        qh[o] = h;
        ql[o] = (__bf16)(v - (float)h);
        ss += v * v;
    }
    __shared__ float red[4];
    for (int off = 32; off; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0 && row < nq) qnorm[row] = sqrtf(red[0] + red[1] + red[2] + red[3]);
}

// DBG != 0: timing ablations only (benchmarks/split_probe.py, MVDB_SPLIT_DBG) — results are NOT valid.
//   2 no hi/lo split, 4 no MFMA, 8 no query DMA, 16 no corpus DMA, 32 no barrier.  (Bit 1, "no nomination", is not
//   offered: without an observable use of the scores hipcc deletes the MFMAs and the split as dead code, so the
//   variant times the DMA skeleton, not "everything but nomination".)
// SEED only names the instantiation: the seed launch shows up under its own kernel name in rocprofv3 summaries.
template <int DBG, bool SEED = false>
__global__ __launch_bounds__(kSplitThreads) void flat_scan_split_kernel(SplitScanArgs a) {
    constexpr int BM = 128, BN = 128, BK = 32, NST = kSplitStages, SB = kSplitStageBytes;
    // The DMA ring and the nominee lists are SEPARATE LDS objects: with both carved from one array hipcc cannot
    // prove that a list read does not alias an in-flight LDS-DMA and drains the ring (s_waitcnt vmcnt(0)) at
    // every tile end — measured +50 % kernel time.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // NST stages of [A raw fp32 128 x 128 B | Bh 128 x 64 B | Bl 128 x 64 B]
    __shared__ uint64_t lists[8 * 64 * kSplitKeep];                         // [8 waves][64 queries][16] keys
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // SGPR: LDS-DMA bases (M0) stay scalar
    const int wr = wave & 3, wq = wave >> 2;  // row group (32 rows), query half (64 queries)
    const int fr = lane & 31, fk = lane >> 5;
    const int n0 = blockIdx.y * BN;
    uint64_t* mylists = lists + (size_t)wave * 64 * kSplitKeep;
    for (int e = lane; e < 64 * kSplitKeep; e += 64) mylists[e] = 0ull;
    float thr[2], floor0[2];
    uint32_t thr_row[2] = {0u, 0u};  // row of the k-th key while thr is its score (beats_key), else 0
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qq = n0 + wq * 64 + j * 32 + fr;
        floor0[j] = qq < a.nq ? (a.thr0 ? a.thr0[qq] : -INFINITY) : INFINITY;
        thr[j] = floor0[j];
    }
    // Consume the floor loads HERE: hipcc cannot see the hand-placed vmcnt waits below, so a first use inside the
    // loop would get its own s_waitcnt vmcnt(0) — which also drains the DMA ring at every tile end (+50 % time).
    asm volatile("" : "+v"(floor0[0]), "+v"(floor0[1]), "+v"(thr[0]), "+v"(thr[1]));

    const int64_t ntiles = a.tile1 - a.tile0;
    const int nk = a.K / BK;
    const int64_t my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int64_t total = my_tiles * nk;  // K-steps of this block over all its tiles: ONE flat pipeline

    // ---- LDS-DMA roles.  The LDS image of a DMA instruction is lane-linear (lane l -> base + 16 l), so the
    // bank swizzle is applied to the SOURCE: 16-byte slot p of a row receives the row's logical slot p ^ g(row).
    //   A (fp32, 8 slots per row):  wave w moves rows 16w .. 16w+15 (two instructions of 8 rows), g = (row >> 1) & 7
    //   Bh / Bl (bf16, 4 slots):    wave w moves rows 16w .. 16w+15 (one instruction each),       g = (row >> 2) & 3
    const int a_row0 = 16 * wave + (lane >> 3), a_row1 = a_row0 + 8;
    const int a_s0 = (lane & 7) ^ ((a_row0 >> 1) & 7), a_s1 = (lane & 7) ^ ((a_row1 >> 1) & 7);
    const int b_row = 16 * wave + (lane >> 2);
    const int b_s = (lane & 3) ^ ((b_row >> 2) & 3);
    const __bf16* bh_src = a.qh + b_row * 32 + b_s * 8;  // K-step-major image: + kt * 128 * 32
    const __bf16* bl_src = a.ql + b_row * 32 + b_s * 8;
    int64_t ld_tile = a.tile0 + blockIdx.x;
    int ld_kt = 0;
    const int64_t last = a.n - 1;
    auto row_ptr = [&](int r, int slot) {
        const int64_t row = ld_tile * BM + r;
        return a.X + (row <= last ? row : last) * a.ld + slot * 4;  // rows past the end: clamped, never nominated
    };
    const float* a_src0 = row_ptr(a_row0, a_s0);
    const float* a_src1 = row_ptr(a_row1, a_s1);
    auto issue = [&](int stage) {
        unsigned char* sa = smem + stage * SB;
        if (!(DBG & 16)) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src0 + ld_kt * BK), (lds_ptr_t)(sa + wave * 2048), 16, 0, 2 /* nt */);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src1 + ld_kt * BK), (lds_ptr_t)(sa + wave * 2048 + 1024), 16, 0, 2);
        }
        if (!(DBG & 8)) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bh_src + ld_kt * (128 * BK)), (lds_ptr_t)(sa + 16384 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bl_src + ld_kt * (128 * BK)), (lds_ptr_t)(sa + 24576 + wave * 1024), 16, 0, 0);
        }
        if (++ld_kt == nk) {
            ld_kt = 0;
            ld_tile += gridDim.x;
            a_src0 = row_ptr(a_row0, a_s0);
            a_src1 = row_ptr(a_row1, a_s1);
        }
    };

    // ---- fragment addresses (bytes inside a stage) ----------------------------------------------------------------
    // A: row wr*32 + fr, k = kk*16 + fk*8 .. +7  -> logical slots kk*4 + fk*2 (+1)
    const int arow = wr * 32 + fr, ag = (arow >> 1) & 7;
    int a_off[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) a_off[kk][h] = arow * 128 + (((kk * 4 + fk * 2 + h) ^ ag) << 4);
    // B: query row wq*64 + j*32 + fr, k = kk*16 + fk*8 .. +7 -> logical slot kk*2 + fk
    int b_off[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qrow = wq * 64 + j * 32 + fr;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b_off[j][kk] = 16384 + qrow * 64 + (((kk * 2 + fk) ^ ((qrow >> 2) & 3)) << 4);
    }

    f32x16m acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // Operand registers of one K-step, two sets (ping-pong): the fragments of step g are read and split while
    // the matrix cores work on step g - 1.
    sbf16x8 ah[2][2], al[2][2], bh[2][2][2], bl[2][2][2];  // [set][kk] / [set][j][kk]

    unsigned int n_ins = 0, n_slow = 0;
    auto nominate = [&](int64_t m0) {
        if (DBG & 1) {
            if (acc[0][0] + acc[1][3] == 1.2345f) thr[0] = 0.f;
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ql = j * 32 + fr;  // this lane's query inside the wave's 64
            // fast reject: after warm-up almost no tile holds a score above the query's 16th best
            float mx = acc[j][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[j][r]);
            if (__ballot(mx >= thr[j]) != 0ull && !(DBG & 64)) {
                ++n_slow;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wr * 32 + (r & 3) + 8 * (r >> 2);
                    const float s = acc[j][r];
                    uint64_t mask = __ballot(m0 + rl + 4 * fk <= last && beats_key(s, (uint32_t)(m0 + rl + 4 * fk), thr[j], thr_row[j]));
                    if (DBG & 128) {
                        if (mask == 0x123456789ull) thr[j] = 1.f;
                        mask = 0;
                    }
                    while (mask) {
                        const int src = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        ++n_ins;
                        const int sq = j * 32 + (src & 31);
                        const float sv = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(s), src));
                        const uint32_t rv = (uint32_t)(m0 + rl + 4 * (src >> 5));
                        uint64_t kth;
                        if (DBG & 256)
                            kth = make_key(sv * 0.5f, rv);  // ablation: no LDS round trips
                        else
                            kth = lds_list_insert(mylists + (size_t)sq * kSplitKeep, kSplitKeep, make_key(sv, rv), lane);
                        if (ql == sq) set_threshold(kth, floor0[j], thr[j], thr_row[j]);  // both lane halves
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
    };
    auto mfma_step = [&](int set) {
        if (DBG & 4) {
            acc[0][0] += (float)ah[set][0][0] + (float)al[set][1][1] + (float)bh[set][0][0][2] + (float)bl[set][0][1][3];
            acc[1][0] += (float)ah[set][1][0] + (float)al[set][0][1] + (float)bh[set][1][1][2] + (float)bl[set][1][0][3];
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // small cross terms first, the leading product last
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[set][kk], bh[set][j][kk], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][kk], bl[set][j][kk], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][kk], bh[set][j][kk], acc[j], 0, 0, 0);
        }
    };

    // Pipeline: step g lives in stage g % NST.  At step g every wave waits for its own DMAs of step g, the block
    // barrier then says "stage g complete, stage g-1 drained"; the DMAs of step g + NST - 1 go into the stage that
    // step g - 1 left; the wave reads step g's fragments, runs the MFMAs of step g - 1 under those reads, then
    // splits step g's corpus fragment.  The sched_barriers pin the DMA order (hipcc otherwise hoists LDS-DMA
    // above LDS reads).
    if (total > 0) {
#pragma unroll
        for (int u = 0; u < NST - 1; ++u) issue(u);
    }
    int cs_kt = 0, stage = 0, fill = NST - 1;
    int64_t cs_tile = a.tile0 + blockIdx.x;
    f32x4m xa[2][2];
    auto enter_step = [&](int set) {  // stage `stage` is complete: refill the drained stage, read this step's fragments
        __builtin_amdgcn_sched_barrier(0);
        if (DBG & 24)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NST == 3)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        // bare s_barrier: __syncthreads() would add a release fence = vmcnt(0), draining the look-ahead DMAs
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(DBG & 32)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(fill);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sb = smem + stage * SB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int h = 0; h < 2; ++h) xa[kk][h] = *reinterpret_cast<const f32x4m*>(sb + a_off[kk][h]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[set][j][kk] = *reinterpret_cast<const sbf16x8*>(sb + b_off[j][kk]);
                bl[set][j][kk] = *reinterpret_cast<const sbf16x8*>(sb + 8192 + b_off[j][kk]);
            }
        }
        stage = stage + 1 == NST ? 0 : stage + 1;
        fill = fill + 1 == NST ? 0 : fill + 1;
    };
    auto split_step = [&](int set) {  // x -> (hi, lo) bf16 pairs, two elements per v_cvt_pk_bf16_f32: 6 VALU ops per pair
        if (DBG & 2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                union { sbf16x8 v; f32x4m f; } r0, r1;
                r0.f = xa[kk][0];
                r1.f = xa[kk][1];
                ah[set][kk] = r0.v;
                al[set][kk] = r1.v;
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            union { sbf16x8 v; uint32_t w[4]; } ahu, alu;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = xa[kk][e >> 1][(e & 1) * 2], x1 = xa[kk][e >> 1][(e & 1) * 2 + 1];
                const uint32_t hp = pack_bf16x2(x0, x1);
                ahu.w[e] = hp;
                alu.w[e] = pack_bf16x2(x0 - __uint_as_float(hp << 16), x1 - __uint_as_float(hp & 0xFFFF0000u));
            }
            ah[set][kk] = ahu.v;
            al[set][kk] = alu.v;
        }
    };
    if (total > 0) {
        enter_step(0);
        split_step(0);
    }
    for (int64_t g0 = 1; g0 < total; g0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t g = g0 + u;
            if (g >= total) break;
            const int set = (u + 1) & 1;  // = g & 1
            enter_step(set);
            // the MFMAs of step g - 1 and the split of step g in ONE scheduling region: 4 VALU ops ride in every
            // MFMA's shadow (an MFMA holds the vector issue port 8 of its 32 cycles)
            mfma_step(set ^ 1);
            split_step(set);
            // keep the split in this block (hipcc otherwise sinks it behind the nomination branch)
            asm volatile("" : "+v"(ah[set][0]), "+v"(ah[set][1]), "+v"(al[set][0]), "+v"(al[set][1]));
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // 4 VALU
            }
            __builtin_amdgcn_sched_barrier(0);
            if (++cs_kt == nk) {
                nominate(cs_tile * BM);
                cs_kt = 0;
                cs_tile += gridDim.x;
            }
        }
    }
    // Seed launch with ONE tile per block (the normal case): no thresholds to learn and nothing to merge into, so
    // instead of ~1,700 serial list inserts per wave the 32 scores every wave holds per query are dumped to LDS
    // (the idle ring + the list array) and the block merge below picks each query's 16 best of 4 x 32.
    const bool dump = SEED && my_tiles == 1;
    if (total > 0) {  // the last step's products
        if ((total - 1) & 1)
            mfma_step(1);
        else
            mfma_step(0);
        if (!dump) nominate(cs_tile * BM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped look-ahead DMAs must land before the LDS is released
    if (dump) {
        auto scratch = [&](int w) {  // [64 queries][32 keys] of wave w
            return w < 6 ? reinterpret_cast<uint64_t*>(smem) + (size_t)w * 2048 : lists + (size_t)(w - 6) * 2048;
        };
        __syncthreads();  // every wave is done with the ring and the lists
        // Lane-owned selection, no wave-cooperative inserts: the wave dumps its keys TRANSPOSED ([key 0..31][query
        // 0..63], conflict-free both ways), then lane q keeps the 16 best of query q's 32 keys in registers
        // (branch-free insertion, skipped when no lane's key beats its 16th); one wave per query half merges the
        // four row groups' lists the same way.
        uint64_t* mine = scratch(wave);
        const int64_t m0 = cs_tile * BM;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                mine[(fk * 16 + r) * 64 + j * 32 + fr] = row <= last ? make_key(acc[j][r], (uint32_t)row) : 0ull;
            }
        uint64_t best[kSplitKeep];
        auto keep_best = [&](uint64_t c) {
            if (c > best[kSplitKeep - 1]) {
#pragma unroll
                for (int t = 0; t < kSplitKeep; ++t) {
                    const bool up = best[t] > c;
                    const uint64_t hi = up ? best[t] : c;
                    c = up ? c : best[t];
                    best[t] = hi;
                }
            }
        };
#pragma unroll
        for (int t = 0; t < kSplitKeep; ++t) best[t] = 0ull;
#pragma unroll 1
        for (int i = 0; i < 32; ++i) keep_best(mine[i * 64 + lane]);
#pragma unroll
        for (int t = 0; t < kSplitKeep; ++t) mine[t * 64 + lane] = best[t];  // over the keys already consumed
        __syncthreads();
        if (wr == 0) {
#pragma unroll
            for (int t = 0; t < kSplitKeep; ++t) best[t] = 0ull;
#pragma unroll 1
            for (int m = 0; m < 4; ++m) {
                const uint64_t* l = scratch(wq * 4 + m);
#pragma unroll 1
                for (int t = 0; t < kSplitKeep; ++t) keep_best(l[t * 64 + lane]);
            }
            const int qq = n0 + wq * 64 + lane;
            if (qq < a.nq) {
#pragma unroll
                for (int t = 0; t < kSplitKeep; ++t)
                    a.cand[((int64_t)qq * gridDim.x + blockIdx.x) * kSplitKeep + t] = best[t];
            }
        }
        return;
    }

    if (a.stats && lane == 0) {
        atomicAdd(a.stats, n_ins);
        atomicAdd(a.stats + 1, n_slow);
    }
    // ---- block merge: query column c has one list in each of the four waves (wr = 0..3) with wq = c / 64 ----------
    __syncthreads();
    for (int c = wave; c < BN; c += 8) {
        const int qq = n0 + c;
        if (qq >= a.nq) continue;
        WaveTopK tk;
        tk.init(kSplitKeep);
#pragma unroll 1
        for (int m = 0; m < 4; ++m) {
            const uint64_t* l = lists + ((size_t)((c / 64) * 4 + m) * 64 + (c % 64)) * kSplitKeep;
            tk.offer(lane < kSplitKeep ? l[lane] : 0ull);
        }
        if (lane < kSplitKeep) a.cand[((int64_t)qq * gridDim.x + blockIdx.x) * kSplitKeep + lane] = tk.key;
    }
}

// ---- 14..32 queries per pass: per-wave rings, query fragments in registers --------------------------------------
// The fp32 32-query pass (flat_scan_mfma2_kernel<KB, 2>) is MFMA-issue-bound at one wave per SIMD (4.0 ms at
// 10M x 512 against a 2.9 ms HBM floor).  Same nominate-and-certify scheme as above on the bf16 cores, in that
// kernel's shape: every wave owns 32-row tiles and streams them through a private two-stage LDS-DMA ring (no block
// barriers), its 32 queries' (hi, lo) bf16 fragments for the whole K live in registers (8 VGPRs per 16-k block),
// D[row][query] keeps the query on the lane.  Stage = 32 rows x 64 floats (8 KiB, eight 1-KiB DMA instructions of
// 4 rows x 256 B); 16-byte slot p of row r receives the row's logical slot p ^ (r & 15) so that the fragment reads
// (row on the lane, 32-byte k-slices) are conflict-free.
struct Split32Args {
    const float* X;
    int64_t n;
    int64_t ld;
    const __bf16* qh;   // K-step-major image of split_queries_kernel: [K / 32][128][32]
    const __bf16* ql;
    int nq;             // <= 32
    uint64_t* cand;     // [nq, gridDim.x, kSplitKeep]
    int64_t tile0;      // 32-row tiles [tile0, tile1)
    int64_t tile1;
    const float* thr0;  // admission floors of the seed launch, or NULL
};

template <int KB>  // K / 16
__global__ __launch_bounds__(kScanThreads) void flat_scan_split32_kernel(Split32Args a) {
    static_assert(KB % 4 == 0, "d must be a multiple of 64");
    constexpr int NS = KB / 4, kStageBytes = 8192, kWaveLds = 2 * kStageBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [4 waves][2 stages][8 KiB]
    __shared__ uint64_t lists[kScanWaves * 32 * kSplitKeep];               // [4 waves][32 queries][16] keys
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 31, fk = lane >> 5;
    unsigned char* wbuf = smem + (size_t)wave * kWaveLds;
    uint64_t* mylists = lists + (size_t)wave * 32 * kSplitKeep;
    for (int e = lane; e < 32 * kSplitKeep; e += 64) mylists[e] = 0ull;

    // ---- query fragments: B[k = 16 kb + 8 fk + j][query fr], (hi, lo) ------------------------------------------------
    sbf16x8 qh[KB], ql[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int64_t o = ((int64_t)(kb >> 1) * 128 + fr) * 32 + (kb & 1) * 16 + fk * 8;
        qh[kb] = *reinterpret_cast<const sbf16x8*>(a.qh + o);
        ql[kb] = *reinterpret_cast<const sbf16x8*>(a.ql + o);
    }
    float floor0 = fr < a.nq ? (a.thr0 ? a.thr0[fr] : -INFINITY) : INFINITY;
    float thr = floor0;
    uint32_t thr_row = 0u;
    // consume every load here: the hand-placed vmcnt waits below are invisible to hipcc (see the kernel above)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+v"(qh[kb]), "+v"(ql[kb]));
    asm volatile("" : "+v"(floor0), "+v"(thr));

    const int64_t ntiles = a.tile1 - a.tile0;
    const int64_t nwaves_total = (int64_t)gridDim.x * kScanWaves;
    const int64_t last = a.n - 1;
    // DMA roles: instruction i (0..7) of a stage moves rows 4i .. 4i+3; lane -> row 4i + (lane >> 4), slot lane & 15
    const int dma_r = lane >> 4, dma_p = lane & 15;
    auto issue_stage = [&](int64_t tile, int ks, int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 4 * i + dma_r;
            int64_t row = (a.tile0 + tile) * 32 + r;
            row = row <= last ? row : last;
            const float* src = a.X + row * a.ld + ks * 64 + 4 * (dma_p ^ (r & 15));
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(wbuf + buf * kStageBytes + i * 1024), 16, 0, 2 /* nt */);
        }
    };
    // fragment read: row fr, 16-k block b of the stage (k = 16 b + 8 fk .. + 7) -> logical slots 4 b + 2 fk (+1)
    int f_off[4][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) f_off[b][h] = fr * 256 + (((4 * b + 2 * fk + h) ^ (fr & 15)) << 4);

    int64_t tile = (int64_t)blockIdx.x * kScanWaves + wave;
    unsigned cnt = 0;
    auto stage_tile = [&](int64_t t, int ks_abs) { return t + (int64_t)(ks_abs / NS) * nwaves_total; };
    if (tile < ntiles) {
        issue_stage(tile, 0, 0);
        const int64_t t1 = stage_tile(tile, 1);
        if (t1 < ntiles) issue_stage(t1, 1 % NS, 1);
    }
    while (tile < ntiles) {
        f32x16m acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            const int buf = cnt & 1;
            const int64_t t1 = stage_tile(tile, ks + 1), t2 = stage_tile(tile, ks + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (t1 < ntiles)
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this stage landed, the next one stays in flight
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* sb = wbuf + buf * kStageBytes;
            f32x4m xa[4][2];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) xa[b][h] = *reinterpret_cast<const f32x4m*>(sb + f_off[b][h]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (t2 < ntiles) issue_stage(t2, (ks + 2) % NS, buf);  // refill the buffer just drained
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int kb = ks * 4 + b;
                union { sbf16x8 v; uint32_t w[4]; } ahu, alu;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = xa[b][e >> 1][(e & 1) * 2], x1 = xa[b][e >> 1][(e & 1) * 2 + 1];
                    const uint32_t hp = pack_bf16x2(x0, x1);
                    ahu.w[e] = hp;
                    alu.w[e] = pack_bf16x2(x0 - __uint_as_float(hp << 16), x1 - __uint_as_float(hp & 0xFFFF0000u));
                }
                // small cross terms first, the leading product last (the order of the 128-query kernel)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alu.v, qh[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahu.v, ql[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahu.v, qh[kb], acc, 0, 0, 0);
            }
            ++cnt;
        }
        // ---- nomination: D[row][query], query on the lane (fr), rows in the 16 registers --------------------------
        {
            const int64_t m0 = (a.tile0 + tile) * 32;
            float mx = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
            if (__ballot(mx >= thr) != 0ull) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = (r & 3) + 8 * (r >> 2);
                    const float s = acc[r];
                    uint64_t mask = __ballot(m0 + rl + 4 * fk <= last && beats_key(s, (uint32_t)(m0 + rl + 4 * fk), thr, thr_row));
                    while (mask) {
                        const int src = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        const int sq = src & 31;
                        const float sv = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(s), src));
                        const uint32_t rv = (uint32_t)(m0 + rl + 4 * (src >> 5));
                        const uint64_t kth = lds_list_insert(mylists + (size_t)sq * kSplitKeep, kSplitKeep,
                                                             make_key(sv, rv), lane);
                        if (fr == sq) set_threshold(kth, floor0, thr, thr_row);  // both lane halves
                    }
                }
            }
        }
        tile += nwaves_total;
    }
    __syncthreads();
    for (int qq = wave; qq < a.nq; qq += kScanWaves) {
        WaveTopK tk;
        tk.init(kSplitKeep);
#pragma unroll 1
        for (int w = 0; w < kScanWaves; ++w) {
            const uint64_t* l = lists + ((size_t)w * 32 + qq) * kSplitKeep;
            tk.offer(lane < kSplitKeep ? l[lane] : 0ull);
        }
        if (lane < kSplitKeep) a.cand[((int64_t)qq * gridDim.x + blockIdx.x) * kSplitKeep + lane] = tk.key;
    }
}

// Seed pass epilogue: merge the per-block lists of the first launch (one tile per block) into each query's 16
// best approximate keys, and publish the 16th score as the admission floor of the main launch — every row of the
// global approximate top-16 scores at least that much, so the main pass only has to insert the few rows above it
// (without the floor each wave re-learns its threshold from scratch: ~7,000 LDS list inserts per wave at 10M rows,
// more time than the MFMAs).
__global__ __launch_bounds__(1024) void split_seed_kernel(const uint64_t* __restrict__ keys, int nlists,
                                                          const uint64_t* prev, uint64_t* seed, float* __restrict__ thr0) {
    // prev: NULL, or the [nq][16] running nominees of the earlier phases (may alias `seed`: read before written)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = blockIdx.x;
    const int64_t total = (int64_t)nlists * kSplitKeep;
    const uint64_t* src = keys + (int64_t)qi * total;
    WaveTopK tk;
    tk.init(kSplitKeep);
    for (int64_t base = (int64_t)wave * 64; base < total; base += 1024) {
        const int64_t i = base + lane;
        tk.offer(i < total ? src[i] : 0ull);
    }
    if (prev && wave == 0) tk.offer(lane < kSplitKeep ? prev[(int64_t)qi * kSplitKeep + lane] : 0ull);
    __shared__ uint64_t sh[15 * 64];
    block_merge_topk(tk, sh, 16);
    if (wave == 0) {
        if (lane < kSplitKeep) seed[(int64_t)qi * kSplitKeep + lane] = tk.key;
        const uint64_t last = readlane_u64(tk.key, kSplitKeep - 1);
        if (lane == 0) thr0[qi] = last ? key_score(last) : -INFINITY;
    }
}

// One block per query: merge the per-block nominee lists to the 16 best approximate keys, re-score
// those rows in exact fp32 (one wave per nominee), emit the exact top-k and certify it.
struct SplitCertifyArgs {
    const uint64_t* keys;  // [nq, nlists, kSplitKeep]
    int nlists;
    const uint64_t* seed;  // [nq, kSplitKeep] nominees of the seed pass, or NULL
    const float* X;
    int64_t ld;
    int d4;                // d / 4
    const float* q;        // [nq, ld] fp32 (normalised if requested)
    const float* qnorm;    // [nq]
    float eps;             // certified operand-error bound per unit |q| |x|, times the row-norm bound
    int k;
    int64_t label_offset;
    float* D;
    int64_t* I;
    int* uncertified;      // incremented once per query that fails the test
    int* failed;           // [nq] set to 1 for a query that fails it (the host re-runs exactly those)
    int l2 = 0;            // 1: the index' metric is squared L2.  Nomination is STILL by inner product (the keys hold approximate
                           // q.x); the nominees are re-scored as sum (q - x)^2 in fp32, ordered by smallest distance, and the
                           // certificate bounds every dropped row's distance from below through |x|^2 >= n2lo:
                           //   d(y) >= |q|^2 + n2lo - 2 (U + eps |q|)  >  r(k-th)        (topk_device.hpp: l2_certified)
    float n2lo = 0.f;      // lower bound of |x|^2 over the stored rows (l2 only)
};

__global__ __launch_bounds__(1024) void split_certify_kernel(SplitCertifyArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = blockIdx.x;
    const int64_t total = (int64_t)a.nlists * kSplitKeep;
    const uint64_t* src = a.keys + (int64_t)qi * total;
    WaveTopK tk;
    tk.init(kSplitKeep);
    for (int64_t base = (int64_t)wave * 64; base < total; base += 1024) {
        const int64_t i = base + lane;
        tk.offer(i < total ? src[i] : 0ull);
    }
    if (a.seed && wave == 0) tk.offer(lane < kSplitKeep ? a.seed[(int64_t)qi * kSplitKeep + lane] : 0ull);
    __shared__ uint64_t sh[15 * 64];
    __shared__ uint64_t nominee[kSplitKeep];
    __shared__ uint64_t exact[kSplitKeep];
    block_merge_topk(tk, sh, 16);
    if (wave == 0 && lane < kSplitKeep) nominee[lane] = tk.key;
    __syncthreads();
    // wave w re-scores nominee w: lane-strided 16-byte loads, butterfly sum (deterministic)
    {
        const uint64_t key = nominee[wave];
        uint64_t ek = 0ull;
        if (key) {
            const uint32_t row = key_row(key);
            const f32x4m* xr = reinterpret_cast<const f32x4m*>(a.X + (int64_t)row * a.ld);
            const f32x4m* qr = reinterpret_cast<const f32x4m*>(a.q + (int64_t)qi * a.ld);
            float s = 0.f;
            if (a.l2) {
                for (int c = lane; c < a.d4; c += 64) {
                    const f32x4m t = qr[c] - xr[c];
                    s = fmaf(t[0], t[0], s);
                    s = fmaf(t[1], t[1], s);
                    s = fmaf(t[2], t[2], s);
                    s = fmaf(t[3], t[3], s);
                }
            } else {
                for (int c = lane; c < a.d4; c += 64) {
                    const f32x4m x = xr[c], w = qr[c];
                    s = fmaf(x[0], w[0], s);
                    s = fmaf(x[1], w[1], s);
                    s = fmaf(x[2], w[2], s);
                    s = fmaf(x[3], w[3], s);
                }
            }
            for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
            ek = make_key(a.l2 ? -s : s, row);   // larger key = better: the smaller distance
        }
        if (lane == 0) exact[wave] = ek;
    }
    __syncthreads();
    if (wave == 0) {
        const uint64_t mine = lane < kSplitKeep ? exact[lane] : 0ull;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < kSplitKeep; ++j) rank += exact[j] > mine;
        if (mine && rank < a.k) {
            a.D[(int64_t)qi * a.k + rank] = a.l2 ? -key_score(mine) : key_score(mine);
            a.I[(int64_t)qi * a.k + rank] = a.label_offset + (int64_t)key_row(mine);
        }
        const int valid = __popcll(__ballot(mine != 0ull));
        if (lane >= valid && lane < a.k) {  // fewer rows than k: faiss' missing-result convention
            a.D[(int64_t)qi * a.k + lane] = a.l2 ? 3.402823466e+38f : -3.402823466e+38f;
            a.I[(int64_t)qi * a.k + lane] = -1;
        }
        // certification: only needed when rows were left out (all 16 nominee slots taken)
        const uint64_t last = nominee[kSplitKeep - 1];
        const uint64_t holder = __ballot(mine && rank == a.k - 1);
        if (last && lane == 0) {
            bool ok = false;
            if (holder) {
                const int hl = __ffsll((long long)holder) - 1;
                const float t = key_score(exact[hl]);
                ok = a.l2 ? l2_certified(-t, key_score(last), a.eps * a.qnorm[qi], a.qnorm[qi], a.n2lo)
                          : t > key_score(last) + a.eps * a.qnorm[qi];
            }
            if (!ok) {
                atomicAdd(a.uncertified, 1);
                a.failed[qi] = 1;
            }
        }
    }
}

}  // namespace mvdb
