// half_scan.hip — query batches per corpus pass: ONE fp16 product over an fp16 SHADOW of the rows nominates, fp32 decides, a
// worst-case bound certifies.
//
//      a(x) = fp16(s_q q) . fp16(s_x x) / (s_q s_x)        one v_mfma_f32_32x32x16_f16 per 16-k block, fp32 accumulate
//      |a(x) - q.x| <= half_eps(d) |q| max|x|              (1.04e-3 at d = 512; derivation at half_eps below)
//
// s_q, s_x are powers of two that put the largest element at 2^14..2^15 (fp16's range is 2^-14..65504, so elements
// down to 2^-29 of the largest keep their 11 significant bits; what falls below is bounded as if flushed to zero).
//
// Nomination and certificate (per query):
//   * every block keeps a sorted list of its 16 best rows (LDS, gated by a threshold register: the larger of the
//     list's 16th score and the admission floor of the phase); the corpus is scanned in phases of growing size and
//     between phases phase_fold_kernel (mvdb.hip) folds the lists into the running 16 best and raises the floors;
//   * U = max(floor of the last phase, 16th score of every block list that filled up) bounds a(x) of every row that
//     was ever dropped;
//   * half_certify_kernel takes the 64 best candidates by a() from (running nominees + last phase's lists) — the set
//     R —, raises U to a(64th) when more than 64 candidates sit above it, re-scores R in fp32 (score r), emits the
//     top-k by r and certifies it with
//         r(k-th) > U + eps |q| max|x|
//     Every row outside R has a <= U, hence a true score below the k-th result's.  Queries that fail (duplicate-heavy
//     neighbourhoods) are flagged one by one; the rescue pass below or the exact fp32 kernels (mvdb.hip) answer them.
//
// Kernels (the queries' fp16 fragments stay in registers for the whole launch, the corpus streams through LDS-DMA rings):
//   flat_scan_h16_kernel<KT, KS, WV, NST, DEPTH>   the pass itself, over the shadow: 128 / 256 queries per pass (32 per wave),
//       d = 128 .. 1024; DEPTH = 32: the rescue launch, which walks a LIST of tiles (rescue_tiles_kernel: the tiles the main launches
//       flagged for its refused queries — HalfScanArgs::tflags — or every tile).  Algorithmic bytes per launch = rows scanned x d x 2.
//   flat_scan_seed_kernel<KQ, SKB, NG, NST>   the first launch of every pass: one 32-row tile of the fp32 rows per block (K
//       split over four waves, partial score tiles exchanged through LDS), every score dumped — it only produces the floors.
// (Retired in round 6: the forms of the pass that converted the fp32 rows on the fly — flat_scan_hq_kernel and the main-launch
//  forms of the K-split kernel, ~700 lines — an index without a shadow answers its batches on the exact fp32 passes.)
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>

#include "common.hpp"
#include "half_scan.hpp"
#include "topk_device.hpp"

namespace mvdb {

typedef _Float16 hs_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 hs_h2 __attribute__((ext_vector_type(2)));
typedef float hs_f2 __attribute__((ext_vector_type(2)));
typedef float hs_f4 __attribute__((ext_vector_type(4)));
typedef float hs_f16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* hs_lds_ptr;
typedef const __attribute__((address_space(1))) void* hs_gbl_ptr;
typedef const __attribute__((address_space(4))) int* hs_cst_i32;

// ---- queries: fp16 image, |q|, scale -----------------------------------------------------------------------------
// One block per (padded) query.  s_q = 2^(15 - e) with max|q_i| = m 2^e, m in [0.5, 1): the largest element lands in
// [2^14, 2^15).  Queries whose largest element is outside 2^-40 .. 2^40 (or not finite) get |q| = +inf, which fails
// the certificate and sends them to the exact path; a zero query scores 0 everywhere and needs no scale.
__global__ __launch_bounds__(256) void half_queries_kernel(const float* __restrict__ q, int64_t ld, int d, int nq,
                                                           int xexp, _Float16* __restrict__ qf,
                                                           float* __restrict__ qnorm, float* __restrict__ qinv) {
    const int row = blockIdx.x;
    float ss = 0.f, mx = 0.f;
    if (row < nq) {
        for (int c = threadIdx.x; c < d; c += 256) {
            const float v = q[(int64_t)row * ld + c];
            ss += v * v;
            mx = fmaxf(mx, fabsf(v));
            if (!(fabsf(v) <= 3.0e38f)) mx = INFINITY;  // NaN / inf
        }
    }
    __shared__ float red[8];
    for (int off = 32; off; off >>= 1) {
        ss += __shfl_xor(ss, off);
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = ss;
        red[4 + (threadIdx.x >> 6)] = mx;
    }
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    int e = 0;
    bool ok = true;
    if (mx > 0.f) {
        if (mx <= 3.0e38f)
            (void)frexpf(mx, &e);
        else
            ok = false;
        if (e < -40 || e > 40) ok = false;
    }
    const int qexp = (mx > 0.f && ok) ? 15 - e : 0;
    const float sq = ldexpf(1.f, qexp);
    for (int c = threadIdx.x; c < d; c += 256) {
        const float v = (row < nq && ok) ? q[(int64_t)row * ld + c] * sq : 0.f;
        qf[(int64_t)row * d + c] = (_Float16)v;  // RNE
    }
    if (threadIdx.x == 0) {
        qinv[row] = ldexpf(1.f, -(qexp + xexp));
        if (row < nq) qnorm[row] = ok ? sqrtf(ss) : INFINITY;
    }
}

// ---- the scan ---------------------------------------------------------------------------------------------------
// KQ  = 16-k blocks per wave (d = 64 KQ: the four waves of a block split K evenly)
// SKB = 16-k blocks per ring stage: 4 -> 32 rows x 256 B (8 KiB, eight DMA instructions of 4 rows), 2 -> 32 rows x 128 B
//       (4 KiB, four DMA instructions of 8 rows); KQ / SKB stages per tile
// NG  = 32-query groups per pass (4 or 8), NST = ring depth in stages
template <int SKB>
struct HsStage {
    static constexpr int kBytes = 2048 * SKB;  // 32 rows x SKB x 64 B
    static constexpr int kDma = 2 * SKB;       // 1-KiB DMA instructions per stage
    static constexpr int kPitch = 64 * SKB;    // bytes per row in LDS
    static constexpr int kRowsPerDma = 1024 / kPitch;
    static constexpr int kSlots = 4 * SKB;     // 16-byte slots per row
    // bank swizzle: 16-byte slot p of row r holds the row's logical slot p ^ g(r)
    __device__ static __forceinline__ int g(int r) { return SKB == 4 ? (r & 15) : ((r >> 1) & 7); }
};

template <int KQ, int SKB, int NG, int NST>
__global__ __launch_bounds__(256) void flat_scan_seed_kernel(HalfScanArgs a) {
    constexpr int NW = 4;
    static_assert(NG % NW == 0 && (SKB == 2 || SKB == 4) && KQ % SKB == 0 && NST >= 2, "shape");
    using St = HsStage<SKB>;
    static_assert((NST - 1) * St::kDma <= 63, "vmcnt is a 6-bit counter");
    constexpr int NR = NG / NW;   // rounds per tile = query groups a wave owns
    constexpr int NS = KQ / SKB;  // stages per tile
    constexpr int kRing = NST * St::kBytes;
    constexpr int K = NW * KQ * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // NW rings | exchange window
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 31, fk = lane >> 5;
    unsigned char* wbuf = smem + (size_t)wave * kRing;
    unsigned char* exch = smem + (size_t)NW * kRing;

    // ---- query fragments of this wave's columns.  In round r the wave works on the NW physical groups
    // r NW .. r NW + NW - 1, rotated so that local group j = 0 is the one it OWNS (physical r NW + wave) and local
    // group j goes to wave (wave + j) % NW: no run-time register indexing anywhere.
    hs_h8 Q[KQ][NR][NW];
#pragma unroll
    for (int kb = 0; kb < KQ; ++kb)
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int query = (r * NW + (wave + j) % NW) * 32 + fr;
                Q[kb][r][j] = *reinterpret_cast<const hs_h8*>(a.qf + (int64_t)query * K + (wave * KQ + kb) * 16 + fk * 8);
            }
    float inv[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) inv[r] = a.qinv[(r * NW + wave) * 32 + fr];
    // consume every global load here: the hand-placed vmcnt waits below are invisible to hipcc, a first use inside
    // the loop would get a compiler-made s_waitcnt vmcnt(0) that also drains the DMA ring
#pragma unroll
    for (int kb = 0; kb < KQ; ++kb)
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int j = 0; j < NW; ++j) asm volatile("" : "+v"(Q[kb][r][j]));
#pragma unroll
    for (int r = 0; r < NR; ++r) asm volatile("" : "+v"(inv[r]));

    const int64_t ntiles = a.tile1 - a.tile0;
    const int64_t last = a.n - 1;
    // ---- DMA roles: instruction i of a stage moves kRowsPerDma rows; lane -> row, 16-byte LDS slot p of that row,
    // which receives the row's LOGICAL slot p ^ g(row) (bank swizzle on the source: the LDS image of a DMA
    // instruction is lane-linear).  Source address = wave-uniform base of (tile, stage) + a per-lane byte offset
    // that never changes.
    uint32_t voff[St::kDma];
#pragma unroll
    for (int i = 0; i < St::kDma; ++i) {
        const int r = St::kRowsPerDma * i + lane / St::kSlots;
        const int slot = (lane % St::kSlots) ^ St::g(r);
        voff[i] = (uint32_t)(((int64_t)r * a.ld + 4 * slot) * 4);
    }
    // Rows past the end of the corpus (last tile) are READ like any other — the index keeps 32 rows of slack behind
    // row n - 1 (mvdb.hip: grow) — and never nominated.
    auto issue_stage = [&](int64_t tile, int ks, int buf) {
        const int64_t row0 = (a.tile0 + tile) * 32;
        const char* sbase = reinterpret_cast<const char*>(a.X + row0 * a.ld + (wave * KQ + ks * SKB) * 16);
        unsigned char* dst = wbuf + buf * St::kBytes;
#pragma unroll
        for (int i = 0; i < St::kDma; ++i)
            __builtin_amdgcn_global_load_lds((hs_gbl_ptr)(sbase + voff[i]), (hs_lds_ptr)(dst + i * 1024), 16, 0, 2 /* nt */);
    };
    // fragment read: row fr, 16-k block b of the stage (k = 16 b + 8 fk .. + 7) -> logical slots 4 b + 2 fk (+1)
    int f_off[SKB][2];
#pragma unroll
    for (int b = 0; b < SKB; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) f_off[b][h] = fr * St::kPitch + (((4 * b + 2 * fk + h) ^ St::g(fr)) << 4);

    const int64_t step = gridDim.x;
    int64_t tile = blockIdx.x;
    const hs_f2 xs = {a.xscale, a.xscale};
    int rb = 0;  // ring buffer of the stage consumed next
    struct Raw {
        hs_f4 v[SKB][2];
    };
    // -- the four steps that bring stage ks of a tile from the ring into fp16 A fragments (row fr, k = 16 kb + 8 fk ..)
    auto wait_stage = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * St::kDma) : "memory");  // oldest stage landed, the others stay in flight
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_stage = [&](Raw& x) {
        const unsigned char* sb = wbuf + rb * St::kBytes;
#pragma unroll
        for (int b = 0; b < SKB; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) x.v[b][h] = *reinterpret_cast<const hs_f4*>(sb + f_off[b][h]);
    };
    // refills the buffer just drained with the stage NST ahead.  Stages past the block's last tile are issued too
    // (clamped to its current tile, landing in buffers nobody reads): the loop body is branch-free and every counted
    // wait sees a full ring.
    auto refill = [&](int64_t cur, int ks) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the fragments are in registers
        __builtin_amdgcn_sched_barrier(0);
        const int64_t t = cur + (int64_t)((ks + NST) / NS) * step;
        issue_stage(t < ntiles ? t : (cur < ntiles ? cur : tile), (ks + NST) % NS, rb);
        rb = rb + 1 == NST ? 0 : rb + 1;
    };
    auto convert = [&](const Raw& x, hs_h8* F, int ks) {
#pragma unroll
        for (int b = 0; b < SKB; ++b) {
            union {
                hs_h8 v;
                hs_h2 p[4];
            } u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const hs_f2 lo = {x.v[b][h][0], x.v[b][h][1]}, hi = {x.v[b][h][2], x.v[b][h][3]};
                u.p[2 * h] = __builtin_convertvector(lo * xs, hs_h2);  // RNE
                u.p[2 * h + 1] = __builtin_convertvector(hi * xs, hs_h2);
            }
            F[ks * SKB + b] = u.v;
        }
    };
    // -- exchange: local group j goes to wave (wave + j) % NW, which finds it in its source slot j - 1
    auto exch_write = [&](hs_f16 (&acc)[NW]) {
#pragma unroll
        for (int j = 1; j < NW; ++j) {
            unsigned char* dst = exch + (((wave + j) % NW) * (NW - 1) + (j - 1)) * 4096 + lane * 16;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<hs_f4*>(dst + r4 * 1024) =
                    hs_f4{acc[j][4 * r4], acc[j][4 * r4 + 1], acc[j][4 * r4 + 2], acc[j][4 * r4 + 3]};
        }
    };
    struct Parts {
        hs_f4 v[NW - 1][4];
    };
    auto exch_read = [&](Parts& p) {
        const unsigned char* src = exch + wave * (NW - 1) * 4096 + lane * 16;
#pragma unroll
        for (int s = 0; s < NW - 1; ++s)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) p.v[s][r4] = *reinterpret_cast<const hs_f4*>(src + s * 4096 + r4 * 1024);
    };
    // the NW partial tiles of the group this wave owns -> 16 scores per lane (query fr, rows rl + 4 fk)
    auto sum_parts = [&](auto rc, const hs_f16& own, const Parts& p, float (&sc)[16]) {
        constexpr int r = decltype(rc)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float s = own[e];
#pragma unroll
            for (int t = 0; t < NW - 1; ++t) s += p.v[t][e >> 2][e & 3];
            sc[e] = s * inv[r];  // exact: 1 / (s_q s_x) is a power of two
        }
    };
    // every score of the tile goes out: [query][block][32 rows] keys (rows past the corpus or outside the bitmap: 0)
    auto gate = [&](auto rc, const float (&sc)[16], int64_t m0) {
        constexpr int r = decltype(rc)::value;
        const int myq = (r * NW + wave) * 32 + fr;
        if (myq < a.nq) {
            uint64_t* out = a.cand + ((int64_t)myq * gridDim.x + blockIdx.x) * 32;
            const uint32_t mw = a.mask ? a.mask[m0 >> 5] : 0xffffffffu;  // the tile's 32 rows = one word of the bitmap
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = (e & 3) + 8 * (e >> 2) + 4 * fk;
                out[rl] = (m0 + rl <= last && ((mw >> rl) & 1u)) ? make_key(sc[e], (uint32_t)(m0 + rl)) : 0ull;
            }
        }
    };
    // whole exchange of one finished round
    auto finish_round = [&](auto rc, hs_f16 (&acc)[NW], int64_t m0) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // every wave has read what the previous round left in the window
        __builtin_amdgcn_sched_barrier(0);
        exch_write(acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // all partial tiles are in LDS
        __builtin_amdgcn_sched_barrier(0);
        Parts p;
        exch_read(p);
        float sc[16];
        sum_parts(rc, acc[0], p, sc);
        gate(rc, sc, m0);
    };

    hs_h8 F[KQ];  // the wave's slice of the tile as fp16 A fragments
    if (tile < ntiles) {  // ONE tile per block (the launch's grid = its tiles)
#pragma unroll
        for (int g = 0; g < NST; ++g) issue_stage(tile, g % NS, g);  // (stages past the tile's NS: re-reads nobody consumes)
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            Raw x;
            wait_stage();
            read_stage(x);
            refill(tile, ks);
            convert(x, F, ks);
        }
        const int64_t m0 = (a.tile0 + tile) * 32;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            hs_f16 acc[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
            for (int kb = 0; kb < KQ; ++kb)
#pragma unroll
                for (int j = 0; j < NW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[kb], Q[kb][r][j], acc[j], 0, 0, 0);
            if (r == 0)
                finish_round(std::integral_constant<int, 0>{}, acc, m0);
            else
                finish_round(std::integral_constant<int, (NR > 1 ? 1 : 0)>{}, acc, m0);
        }
    }
    static_assert(NR <= 2, "finish_round dispatch covers two rounds");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped look-ahead DMAs must land before the LDS is released
}

// ---- the same pass over the fp16 SHADOW of the corpus (round 4) -------------------------------------------------------
// (The forms of rounds 3 - 5 read fp32 rows and converted every 32-row tile to fp16 on its way to the matrix cores: for a
// nomination pass that is twice the bytes the arithmetic needs, plus a raw staging ring, the conversion's VALU work and a
// second LDS round trip per element; retired in round 6.)  An index that answers batches keeps — lazily, from its first batch search on — an
// fp16 copy of its rows, Xh[n][d] = fp16(s_x x) (mvdb.hip: shadow; exactly the values the conversion produces, so the
// certificate and its bound are unchanged; 2 bytes per element on top of the 4 of the fp32 matrix, which stays the home of
// the exact scans and of the re-scores).  This kernel streams THAT:
//   * a stage = KS 16-k blocks of a 32-row tile = 32 rows x 32 KS bytes (in use: KS = KT, the whole tile), DMA'd straight into
//     the image the MFMA fragments are read from: no conversion, no second buffer.  The LDS image of a DMA is lane-linear,
//     so the bank swizzle (16-byte slot p of row r holds the row's logical slot p ^ (r & 15), inside aligned groups of 16
//     slots: conflict-free b128 fragment reads for any multiple of 16 slots per row) is applied to the per-lane SOURCE address;
//   * NST stages in flight per workgroup (each wave issues its share and waits for its own with a counted vmcnt; one bare
//     s_barrier per stage), the queries' fragments in registers for the whole launch;
//   * gate: a threshold register per query (the larger of the list's 16th score and the admission floor), lists of 16 keys per
//     (wave, query) in LDS with wave-cooperative sorted inserts, the row bitmap looked at on the slow path only.
// ALGORITHMIC bytes per launch = rows scanned x d x 2 (the pass's operand is the shadow).  At 256 queries per pass and d = 512
// the matrix cores bound (2 x 32 MFMAs per SIMD and tile = 2,048 cycles against ~2,400 for the tile's 32 KiB at the HBM
// rate ... at a clock the chip lowers under this load); at 128 queries HBM bounds.
template <int KT, int KS, int WV, int NST, int DEPTH = kHalfKeep>
__global__ __launch_bounds__(WV * 64) void flat_scan_h16_kernel(HalfScanArgs a) {
    if (a.gate && *a.gate <= a.gate_lo) return;  // (rescue launches: nothing was refused / not this many)
    constexpr int K = KT * 16;
    constexpr int NSTG = KT / KS;            // stages per tile
    constexpr int HSL = KS * 2;              // 16-byte slots (8 fp16) per row and stage
    constexpr int kStage = 32 * HSL * 16;    // bytes
    constexpr int NP = kStage / 1024;        // DMA instructions per stage
    constexpr int DPW = NP / WV;             // ... per wave
    static_assert(KT % KS == 0 && HSL % 16 == 0 && NP % WV == 0 && DPW >= 1, "shape");
    static_assert((NST - 1) * DPW <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // NST stages
    __shared__ uint64_t lists[WV * 32 * DEPTH];                         // [wave][32 queries][16] keys
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 31, fk = lane >> 5;
    uint64_t* mylists = lists + (size_t)wave * 32 * DEPTH;
    for (int e = lane; e < 32 * DEPTH; e += 64) mylists[e] = 0ull;

    // ---- this wave's 32 queries over the whole K: B[k = 16 kb + 8 fk + j][query fr]
    hs_h8 Q[KT];
#pragma unroll
    for (int kb = 0; kb < KT; ++kb)
        Q[kb] = *reinterpret_cast<const hs_h8*>(a.qf + (int64_t)(wave * 32 + fr) * K + kb * 16 + fk * 8);
    const int myq = wave * 32 + fr;
    float floor0 = myq < a.nq ? (a.thr0 ? a.thr0[myq] - (a.thr_qn ? a.thr_eps * a.thr_qn[myq] : 0.f) : -INFINITY) : INFINITY;
    if (!(floor0 == floor0)) floor0 = -INFINITY;  // (inf - inf: a query outside the fp16 range keeps no floor)
    float thr = floor0;
    uint32_t thr_row = 0u;
    float thr_lo = thr;  // (eight-wave form) thr less the flag band
    float inv = a.qinv[myq];
    // LIST: the rescue launch walks the tiles rescue_tiles_kernel listed (tile_list[0 .. *tile_count)), in list order
    constexpr bool LIST = DEPTH != kHalfKeep;
    constexpr int LA = (NSTG - 1 + NST - 1) / NSTG;  // the look-ahead stage of a tile's last stage lies this many tiles ahead
    // main launches: the tile's bit of the query's flag row is raised when a score comes within fband of the running threshold
    float fband = 0.f;
    uint32_t* myflags = nullptr;
    if constexpr (!LIST) {
        if (a.tflags && myq < a.nq) {
            fband = a.flag_coef * a.flag_qn[myq];
            myflags = a.tflags + (int64_t)myq * a.twords;
        }
    }
    int listed = 0;
    if constexpr (LIST) listed = *a.tile_count;
    // consume every global load here: the hand-placed vmcnt waits below are invisible to hipcc (see flat_scan_seed_kernel)
#pragma unroll
    for (int kb = 0; kb < KT; ++kb) asm volatile("" : "+v"(Q[kb]));
    asm volatile("" : "+v"(floor0), "+v"(thr), "+v"(inv), "+v"(fband), "+v"(listed));

    thr_lo = thr - fband;
    const int64_t ntiles = LIST ? (int64_t)__builtin_amdgcn_readfirstlane(listed) : a.tile1 - a.tile0;
    const int64_t last = a.n - 1;
    // DMA roles: piece p = wave DPW + i of a stage fills LDS bytes [1024 p, 1024 p + 1024) = rows (64 p + lane) / HSL
    uint32_t voff[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int j = (wave * DPW + i) * 64 + lane;
        const int row = j / HSL, ps = j % HSL;
        voff[i] = (uint32_t)(row * (K * 2) + ((ps ^ (row & 15)) << 4));
    }
    const int64_t step = gridDim.x;
    int64_t tile = blockIdx.x;
    // (LIST) the tiles at list positions tile, tile + step, ..., tile + LA step: scalar loads through the constant address space —
    // the vector-memory counter stays the DMA ring's alone; positions past the end repeat the last entry (re-reads nobody consumes)
    hs_cst_i32 tl = (hs_cst_i32)a.tile_list;
    auto list_at = [&](int64_t t) { return tl[t < ntiles ? t : (ntiles > 0 ? ntiles - 1 : 0)]; };
    int rq[LA + 1];
#pragma unroll
    for (int i = 0; i <= LA; ++i) rq[i] = LIST ? list_at(tile + i * step) : 0;
    // stage c of the block's flat sequence = (tile + (c / NSTG) step, K part c % NSTG); tiles past the end are clamped
    auto issue_piece = [&](int64_t base, int c, int buf, int i) {
        int64_t t = base + (int64_t)(c / NSTG) * step;
        t = t < ntiles ? t : (base < ntiles ? base : 0);
        if constexpr (LIST) t = rq[c / NSTG];
        const char* sbase = reinterpret_cast<const char*>(a.Xh) + ((a.tile0 + t) * 32 * (int64_t)K + (c % NSTG) * KS * 16) * 2;
        unsigned char* dst = smem + buf * kStage + wave * DPW * 1024;
        __builtin_amdgcn_global_load_lds((hs_gbl_ptr)(sbase + voff[i]), (hs_lds_ptr)(dst + i * 1024), 16, 0, 2 /* nt */);
    };
    auto issue_stage = [&](int64_t base, int c, int buf) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) issue_piece(base, c, buf, i);
    };
    const int frow = fr * HSL * 16;
    const int fsw = fr & 15;
    unsigned n_ins = 0, n_slow = 0;
    hs_f16 acc;
    auto gate = [&](int64_t m0) {
        float sc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = acc[e] * inv;  // exact: 1 / (s_q s_x) is a power of two
        if (a.hn) {
            // L2 metric: nominate by q.x - |x|^2 / 2.  The tile's 32 offsets come through the SCALAR cache (lgkmcnt): the
            // vector-memory counter stays the DMA ring's alone.  acc[e] is row (e & 3) + 8 (e >> 2) + 4 fk of the tile.
            hs_f16 h0, h1;
            const float* hp = a.hn + m0;
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(h0), "=&s"(h1)
                         : "s"(hp)
                         : "memory");
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = (e & 3) + 8 * (e >> 2);
                const float lo = rl < 16 ? h0[rl & 15] : h1[rl & 15];
                const float hi = rl + 4 < 16 ? h0[(rl + 4) & 15] : h1[(rl + 4) & 15];
                sc[e] -= fk ? hi : lo;
            }
        }
        float mx = sc[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sc[e]);
        // A score within fband of the threshold this wave holds BEFORE the tile: the tile may hold a row the rescue pass has to see,
        // should the query be refused (mvdb.hip: the band's derivation).  NaN scores flag too.
        if constexpr (WV == 8) {
            // eight waves: ONE test on the fast path, thr_lo = thr - fband (= thr where no flags are kept), the flag write and the
            // inserts behind it — 256 queries per call, 10M x 512: 0.673 -> 0.663 ms per launch.  (The four-wave launches LOSE 10 - 15 %
            // with this form — same instruction counts, another placement of the ring's waits — and keep the second ballot below.)
            if (__ballot(!(mx < thr_lo)) == 0ull) return;
            if constexpr (!LIST) {
                if (myflags != nullptr && !(mx < thr_lo)) {
                    const int64_t gt = m0 >> 5;
                    atomicOr(myflags + (gt >> 5), 1u << (gt & 31));
                }
            }
        } else if constexpr (!LIST) {
            const bool hit = myflags != nullptr && !(mx < thr - fband);
            if (__ballot(hit) != 0ull) {
                const int64_t gt = m0 >> 5;
                if (hit) atomicOr(myflags + (gt >> 5), 1u << (gt & 31));
            }
        }
        if (__ballot(mx >= thr) != 0ull) {
            ++n_slow;
            const uint32_t mw = a.mask ? a.mask[m0 >> 5] : 0xffffffffu;  // row selection: looked at on the slow path only
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = (e & 3) + 8 * (e >> 2);
                const float s = sc[e];
                uint64_t mask = __ballot(m0 + rl + 4 * fk <= last && ((mw >> (rl + 4 * fk)) & 1u) && beats_key(s, (uint32_t)(m0 + rl + 4 * fk), thr, thr_row));
                while (mask) {
                    const int srcl = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    ++n_ins;
                    const int sq = srcl & 31;
                    const float sv = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(s), srcl));
                    const uint32_t rv = (uint32_t)(m0 + rl + 4 * (srcl >> 5));
                    const uint64_t kth = lds_list_insert(mylists + (size_t)sq * DEPTH, DEPTH, make_key(sv, rv), lane);
                    if (fr == sq) set_threshold(kth, floor0, thr, thr_row);  // both lane halves
                }
            }
            if constexpr (WV == 8) thr_lo = thr - fband;
        }
    };

    if (tile < ntiles) {
#pragma unroll
        for (int c = 0; c < NST - 1; ++c) issue_stage(tile, c, c);
    }
    int buf = 0;
    while (tile < ntiles) {
        const int64_t m0 = (LIST ? (int64_t)rq[0] : a.tile0 + tile) * 32;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int sg = 0; sg < NSTG; ++sg) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * DPW) : "memory");  // this wave's pieces of the stage have landed
            __builtin_amdgcn_s_barrier();  // everybody's have; and every wave is done reading the stage before
            __builtin_amdgcn_sched_barrier(0);
            const int abuf = buf == 0 ? NST - 1 : buf - 1;  // the look-ahead stage goes into the buffer of the stage before
            issue_stage(tile, sg + NST - 1, abuf);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* src = smem + buf * kStage + frow;
            auto frag = [&](int kb) { return *reinterpret_cast<const hs_h8*>(src + (((2 * kb + fk) ^ fsw) << 4)); };
            constexpr int AHEAD = KS < 6 ? KS : 6;
            hs_h8 f[AHEAD + 1];
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) f[u] = frag(u);
#pragma unroll
            for (int kb = 0; kb < KS; ++kb) {
                if (kb + AHEAD < KS) f[(kb + AHEAD) % (AHEAD + 1)] = frag(kb + AHEAD);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[kb % (AHEAD + 1)], Q[sg * KS + kb], acc, 0, 0, 0);
            }
            // the order the scheduler must keep: the first fragments up front, then one fragment read per MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
            for (int kb = 0; kb < KS; ++kb) {
                if (kb + AHEAD < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the fragments are in registers before the buffer may be refilled)
            buf = buf == NST - 1 ? 0 : buf + 1;
        }
        gate(m0);   // (timing ablation, 256 queries at 10M x 512: without the gate the launch takes 0.91 ms instead of 0.89 — not what bounds it)
        tile += step;
        if constexpr (LIST) {
#pragma unroll
            for (int i = 0; i < LA; ++i) rq[i] = rq[i + 1];
            rq[LA] = list_at(tile + LA * step);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped look-ahead DMAs must land before the LDS is released
    if (a.stats && lane == 0) {
        atomicAdd(a.stats, n_ins);
        atomicAdd(a.stats + 1, n_slow);
    }
#pragma unroll 1
    for (int q = 0; q < 32; ++q) {
        const int qq = wave * 32 + q;
        if (qq >= a.nq) break;
        if (lane < DEPTH)
            a.cand[((int64_t)qq * gridDim.x + blockIdx.x) * DEPTH + lane] = mylists[(size_t)q * DEPTH + lane];
    }
}

// fp32 rows -> the shadow: Xh[r][c] = fp16(s_x X[r][c]) (RNE; s_x a power of two: the product is exact) — the values
// the on-the-fly conversion of rounds 3 - 5 produced
__global__ __launch_bounds__(256) void half_shadow_kernel(const float* __restrict__ X, int64_t ld, int d, int64_t n, float xscale,
                                                          _Float16* __restrict__ Xh) {
    const int64_t total = n * (int64_t)(d / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (d / 4);
        const int c = (int)(i % (d / 4)) * 4;
        const hs_f4 v = *reinterpret_cast<const hs_f4*>(X + r * ld + c);
        union {
            hs_h2 p[2];
            hs_f2 f;
        } u;
        u.p[0] = __builtin_convertvector(hs_f2{v[0], v[1]} * hs_f2{xscale, xscale}, hs_h2);
        u.p[1] = __builtin_convertvector(hs_f2{v[2], v[3]} * hs_f2{xscale, xscale}, hs_h2);
        *reinterpret_cast<hs_f2*>(Xh + r * d + c) = u.f;
    }
}

int launch_half_shadow(const float* X, int64_t ld, int d, int64_t n, float xscale, _Float16* Xh, int device, hipStream_t stream) {
    if (n <= 0) return 0;
    const int64_t total = n * (int64_t)(d / 4);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, (int64_t)device_cus(device) * 16));
    hipLaunchKernelGGL(half_shadow_kernel, dim3(grid), dim3(256), 0, stream, X, ld, d, n, xscale, Xh);
    MVDB_HIP(hipGetLastError());
    return 0;
}

// Hn[r] = |x_r|^2 / 2: a wave per row, every lane sums its 16-byte pieces with fmas, butterfly over the lanes
__global__ __launch_bounds__(256) void half_norms_kernel(const float* __restrict__ X, int64_t ld, int d4, int64_t n,
                                                         float* __restrict__ Hn) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
        const hs_f4* xr = reinterpret_cast<const hs_f4*>(X + r * ld);
        float s = 0.f;
        for (int c = lane; c < d4; c += 64) {
            const hs_f4 v = xr[c];
            s = fmaf(v[0], v[0], s);
            s = fmaf(v[1], v[1], s);
            s = fmaf(v[2], v[2], s);
            s = fmaf(v[3], v[3], s);
        }
        for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) Hn[r] = 0.5f * s;
    }
}

int launch_half_norms(const float* X, int64_t ld, int d, int64_t n, float* Hn, int device, hipStream_t stream) {
    if (n <= 0) return 0;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)device_cus(device) * 16));
    hipLaunchKernelGGL(half_norms_kernel, dim3(grid), dim3(256), 0, stream, X, ld, (d + 3) / 4, n, Hn);
    MVDB_HIP(hipGetLastError());
    return 0;
}

// d <= 512: 128 and 256 queries per pass; d = 640 .. 1024 (e5-large / bge-m3 widths): 128 queries per pass, one wave per SIMD
// (160 - 256 registers of query fragments)
bool half_shadow_dim(int d) { return d == 128 || d == 256 || d == 384 || d == 512 || d == 640 || d == 768 || d == 896 || d == 1024; }

template <int KT, int KS, int WV, int NST, int BPC = 1>
static int launch_h16_inst(const HalfScanArgs& a, int device, hipStream_t stream, int* nblocks_out) {
    auto kern = flat_scan_h16_kernel<KT, KS, WV, NST>;
    constexpr size_t lds = (size_t)NST * 32 * KS * 2 * 16;
    static_assert(BPC * (lds + WV * 32 * kHalfKeep * 8) <= 160 * 1024, "LDS budget of a CU");
    {
        static std::mutex mu;
        static std::map<int, bool> done;
        std::lock_guard<std::mutex> lk(mu);
        if (!done[device]) {
            MVDB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            done[device] = true;
        }
    }
    const int64_t ntiles = a.tile1 - a.tile0;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, (int64_t)device_cus(device) * BPC));
    *nblocks_out = nblocks;
    // (a launch without admission floors is the seed of an L2 pass over the shadow, mvdb.hip: launch_half_pass)
    const char* pname = a.thr0 ? "ip_scan_half" : "ip_scan_half_seed";
    prof_symbol(pname, "flat_scan_h16_kernel<%d, %d, %d, %d, %d>", KT, KS, WV, NST, kHalfKeep);
    int slot = prof_begin(pname, stream);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WV * 64), lds, stream, a);
    prof_end(slot, stream);
    MVDB_HIP(hipGetLastError());
    return 0;
}

// main launches over the shadow.  A stage = a whole 32-row tile (KS = KT: 16 / 24 / 32 KiB at d = 256 / 384 / 512: one barrier
// per tile).  256 queries per pass: eight waves (two per SIMD), a ring of 3 - 4 tiles.  128 queries per pass: four waves and
// TWO workgroups per CU — the second workgroup's waves issue MFMAs while the first's sit at their barrier or in the gate —,
// 2 - 4 tiles per ring.  Sweep at 10M x 512 (profiles/r04_h16_variants.txt): 128 queries 66.5k q/s with one workgroup per CU
// (KS 16, ring of 6), 72.3k with two (KS 32, ring of 2); 256 queries 89.2k (KS 16, ring of 6) -> 92.1k (KS 32, ring of 3);
// DMA pieces spread between the MFMAs: +-1 %; 8-KiB stages: -5 %.
static int launch_h16(int d, int nqpad, const HalfScanArgs& a, int device, hipStream_t stream, int* nb) {
    const bool wide = nqpad == 256;
    if (nqpad != 128 && nqpad != 256) return fail(MVDB_ERR_ARG, "no fp16-shadow kernel for %d queries per pass", nqpad);
    switch (d) {
        case 128: return wide ? launch_h16_inst<8, 8, 8, 4>(a, device, stream, nb) : launch_h16_inst<8, 8, 4, 4, 2>(a, device, stream, nb);
        case 256: return wide ? launch_h16_inst<16, 16, 8, 4>(a, device, stream, nb) : launch_h16_inst<16, 16, 4, 4, 2>(a, device, stream, nb);
        case 384: return wide ? launch_h16_inst<24, 24, 8, 4>(a, device, stream, nb) : launch_h16_inst<24, 24, 4, 2, 2>(a, device, stream, nb);
        case 512: return wide ? launch_h16_inst<32, 32, 8, 3>(a, device, stream, nb) : launch_h16_inst<32, 32, 4, 2, 2>(a, device, stream, nb);
        case 640: if (!wide) return launch_h16_inst<40, 40, 4, 3>(a, device, stream, nb); break;
        case 768: if (!wide) return launch_h16_inst<48, 48, 4, 3>(a, device, stream, nb); break;
        case 896: if (!wide) return launch_h16_inst<56, 56, 4, 2>(a, device, stream, nb); break;
        case 1024: if (!wide) return launch_h16_inst<64, 64, 4, 2>(a, device, stream, nb); break;
        default: break;
    }
    return fail(MVDB_ERR_ARG, "no fp16-shadow kernel for %d queries per pass at d = %d", nqpad, d);
}

// ---- the rescue pass (round 5) -----------------------------------------------------------------------------------------
// A refused query is not an unknown one: half_certify_kernel has re-scored its 64 nominees exactly, and the k-th of those scores,
// t, bounds the k-th result from below.  Every row of the top k therefore has an APPROXIMATE score >= t - margin - eps |q| =: f.
// The rescue launch streams the shadow ONCE more — since round 6 only the 32-row tiles some refused query of the launch was flagged for
// by the certified pass's main launches (mvdb.hip: launch_half_pass has the bound; L2, k > 16, small corpora: every tile) — for up to
// 128 refused queries with f as the admission floor and 32-deep lists per
// (block, query): unless a list fills, the lists hold EVERY row that can be in the top k; half_rescue_certify_kernel re-scores
// all of them in fp32 and takes the top k — exact, no certificate needed.  A full list (a neighbourhood of more than ~30 rows per
// block inside the band) raises the query's `need` word and its 32-query exact pass runs as before.
template <int KT, int KS, int NST, int BPC>
static int launch_h16_rescue_inst(const HalfScanArgs& a, int device, hipStream_t stream, int* nblocks_out) {
    auto kern = flat_scan_h16_kernel<KT, KS, 4, NST, kRescueKeep>;
    constexpr size_t lds = (size_t)NST * 32 * KS * 2 * 16;
    static_assert(BPC * (lds + 4 * 32 * kRescueKeep * 8) <= 160 * 1024, "LDS budget of a CU");
    static_assert(BPC <= kRescueBlocksPerCu, "mvdb.hip sizes the lists by kRescueBlocksPerCu");
    {
        static std::mutex mu;
        static std::map<int, bool> done;
        std::lock_guard<std::mutex> lk(mu);
        if (!done[device]) {
            MVDB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            done[device] = true;
        }
    }
    const int64_t ntiles = a.tile1 - a.tile0;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, (int64_t)device_cus(device) * BPC));
    *nblocks_out = nblocks;
    prof_symbol("ip_scan_rescue", "flat_scan_h16_kernel<%d, %d, 4, %d, %d>", KT, KS, NST, kRescueKeep);
    int slot = prof_begin("ip_scan_rescue", stream);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, stream, a);
    prof_end(slot, stream);
    MVDB_HIP(hipGetLastError());
    return 0;
}
// The rescue launches' tile lists: thread w of slot s = blockIdx.y ORs word w of the flag rows of the slot's refused queries, adds
// the seed's tiles, drops the bits past the last tile and appends the set bits to the slot's list (order: as the atomics fall).
__global__ __launch_bounds__(256) void rescue_tiles_kernel(RescueTilesArgs a) {
    const int nb = *a.nfail;
    const int r0 = blockIdx.y * kRescueQueries;
    if (nb <= r0) return;
    const int r1 = nb < r0 + kRescueQueries ? nb : r0 + kRescueQueries;
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t t0 = w * 32;
    if (a.stats && w == 0) atomicAdd(a.stats + 1, (unsigned long long)a.ntiles);
    if (t0 >= a.ntiles) return;
    uint32_t word = a.tflags ? 0u : 0xffffffffu;
    if (a.tflags)
        for (int r = r0; r < r1; ++r) word |= a.tflags[a.map[r] * a.twords + w];
    if (t0 < a.seed_tiles) word |= a.seed_tiles - t0 >= 32 ? 0xffffffffu : (1u << (a.seed_tiles - t0)) - 1u;
    if (t0 + 32 > a.ntiles) word &= (1u << (a.ntiles - t0)) - 1u;
    const int c = __popc(word);
    if (c == 0) return;
    int at = atomicAdd(a.counts + blockIdx.y, c);
    if (a.stats) atomicAdd(a.stats, (unsigned long long)c);
    int* list = a.lists + (int64_t)blockIdx.y * a.ntiles;
    while (word) {
        const int b = __ffs(word) - 1;
        word &= word - 1;
        list[at++] = (int)(t0 + b);
    }
}
int launch_rescue_tiles(const RescueTilesArgs& a, int slots, hipStream_t stream) {
    const int64_t words = (a.ntiles + 31) / 32;
    hipLaunchKernelGGL(rescue_tiles_kernel, dim3((unsigned)((words + 255) / 256), (unsigned)slots), dim3(256), 0, stream, a);
    MVDB_HIP(hipGetLastError());
    return 0;
}

bool half_rescue_dim(int d) { return half_shadow_dim(d); }
int launch_half_rescue_scan(int d, const HalfScanArgs& a, int device, hipStream_t stream, int* nb) {
    // d <= 512: TWO workgroups per CU (80 KiB each: the 32 KiB of lists + a ring of 48 KiB — three half-tile stages at d = 512, two
    // whole tiles at 384, three at 256) — as in the 128-query main launch, one workgroup's waves issue while the other's sit at
    // their barrier or in the gate.  Against one workgroup per CU on a ring of three whole tiles (round 5), clustered corpus, 256
    // queries per call: live rescue launch 1.86 -> 1.61 ms at 10M x 512 (10.24 GB: 0.69 -> 0.80 of HBM), 1.39 -> 1.15 at 10M x 384,
    // 0.59 -> 0.41 at 1M x 512, 1.49 -> 0.92 at 4M x 256, 0.97 -> 0.57 at 4M x 128 (both launches live there); the call 4.82 -> 4.65 / 3.81 -> 3.63 /
    // 1.16 -> 1.00 / 2.46 -> 1.90 / 1.72 -> 1.34 ms (profiles/r06_rescue_form_ab.jsonl).  The
    // wide shapes keep one workgroup per CU: their query fragments alone are 160 - 256 registers per lane.
    switch (d) {
        case 128: return launch_h16_rescue_inst<8, 8, 4, 2>(a, device, stream, nb);
        case 256: return launch_h16_rescue_inst<16, 16, 3, 2>(a, device, stream, nb);
        case 384: return launch_h16_rescue_inst<24, 24, 2, 2>(a, device, stream, nb);
        case 512: return launch_h16_rescue_inst<32, 16, 3, 2>(a, device, stream, nb);
        case 640: return launch_h16_rescue_inst<40, 40, 3, 1>(a, device, stream, nb);   // (e5-large / bge-m3 widths: two-stage rings from 768 on)
        case 768: return launch_h16_rescue_inst<48, 48, 2, 1>(a, device, stream, nb);
        case 896: return launch_h16_rescue_inst<56, 56, 2, 1>(a, device, stream, nb);
        case 1024: return launch_h16_rescue_inst<64, 64, 2, 1>(a, device, stream, nb);
        default: return fail(MVDB_ERR_ARG, "no rescue kernel for d = %d", d);
    }
}

// One block per compact query.  Collects the rescue lists' keys; a full list or more than kRescueCap keys: the query stays with the
// exact pass (need word raised).  Else every candidate is re-scored in fp32 (the arithmetic of half_certify_kernel) and the k
// best (score, row) keys are the query's results.
__global__ __launch_bounds__(1024) void half_rescue_certify_kernel(HalfRescueArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x;                    // compact query gate_lo + slot
    if (*a.gate <= a.gate_lo + slot) return;
    __shared__ uint64_t cand[kRescueCap];           // approximate keys, then exact keys
    __shared__ int s_cnt, s_full;
    __shared__ uint64_t wmax[16];
    if (threadIdx.x == 0) {
        s_cnt = 0;
        s_full = 0;
    }
    __syncthreads();
    const int64_t total = (int64_t)a.nlists * kRescueKeep;
    const uint64_t* src = a.keys + (int64_t)slot * total;
    for (int64_t i = threadIdx.x; i < total; i += 1024) {
        const uint64_t key = src[i];
        if (!key) continue;
        if ((i % kRescueKeep) == kRescueKeep - 1) s_full = 1;   // the list's last slot is taken: rows may have been dropped
        const int at = atomicAdd(&s_cnt, 1);
        if (at < kRescueCap) cand[at] = key;
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (s_full || cnt > kRescueCap || cnt < a.k) {
        if (threadIdx.x == 0) atomicOr(a.need + slot / a.per_pass, 1);
        return;
    }
    const hs_f4* qr = reinterpret_cast<const hs_f4*>(a.q + (int64_t)slot * a.ld);
    // a wave re-scores FOUR candidates per round (i, i + 16, i + 32, i + 48: four rows' loads in flight instead of one — a refused
    // query of the clustered corpus brings ~2,400 candidates to ONE block: 0.21 -> 0.07 ms); per candidate the arithmetic is what
    // it was: lane-strided fmas, butterfly over the lanes
    for (int i0 = wave; i0 < cnt; i0 += 64) {
        const hs_f4* xr[4];
        uint32_t row[4];
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 16 * u;
            row[u] = key_row(cand[i < cnt ? i : i0]);
            xr[u] = reinterpret_cast<const hs_f4*>(a.X + (int64_t)row[u] * a.ld);
        }
        if (a.l2) {
            for (int c = lane; c < a.d4; c += 64) {
                const hs_f4 w = qr[c];
                hs_f4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = xr[u][c];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const hs_f4 t = w - x[u];
                    s[u] = fmaf(t[0], t[0], s[u]);
                    s[u] = fmaf(t[1], t[1], s[u]);
                    s[u] = fmaf(t[2], t[2], s[u]);
                    s[u] = fmaf(t[3], t[3], s[u]);
                }
            }
        } else {
            for (int c = lane; c < a.d4; c += 64) {
                const hs_f4 w = qr[c];
                hs_f4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = xr[u][c];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s[u] = fmaf(x[u][0], w[0], s[u]);
                    s[u] = fmaf(x[u][1], w[1], s[u]);
                    s[u] = fmaf(x[u][2], w[2], s[u]);
                    s[u] = fmaf(x[u][3], w[3], s[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = s[u];
            for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
            const int i = i0 + 16 * u;
            if (lane == 0 && i < cnt) cand[i] = make_key(a.l2 ? -v : v, row[u]);   // larger key = better: the smaller distance
        }
    }
    __syncthreads();
    // k rounds of a block-wide maximum over the exact keys (unique: the row is part of the key)
    for (int r = 0; r < a.k; ++r) {
        uint64_t best = 0ull;
        for (int i = threadIdx.x; i < cnt; i += 1024) best = cand[i] > best ? cand[i] : best;
        for (int off = 32; off; off >>= 1) {
            const uint64_t o = __shfl_xor(best, off);
            best = o > best ? o : best;
        }
        if (lane == 0) wmax[wave] = best;
        __syncthreads();
        best = wmax[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) best = wmax[w] > best ? wmax[w] : best;
        if (threadIdx.x == 0) {
            a.D[(int64_t)slot * a.k + r] = a.l2 ? -key_score(best) : key_score(best);
            a.I[(int64_t)slot * a.k + r] = a.label_offset + (int64_t)key_row(best);
        }
        for (int i = threadIdx.x; i < cnt; i += 1024)
            if (cand[i] == best) cand[i] = 0ull;
        __syncthreads();
    }
}

int launch_half_rescue_certify(const HalfRescueArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(half_rescue_certify_kernel, dim3(kRescueQueries), dim3(1024), 0, stream, a);
    MVDB_HIP(hipGetLastError());
    return 0;
}

// ---- certification ------------------------------------------------------------------------------------------------
// One block per query.  U = max(last floor, 16th score of every full block list); R = the 64 best candidates by
// approximate score; U is raised to a(64th) when R is full; fp32 re-score of R (one wave per nominee, the arithmetic
// of split_certify_kernel); top-k by exact score; certificate r(k-th) > U + eps |q|.
__global__ __launch_bounds__(1024) void half_certify_kernel(HalfCertifyArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = blockIdx.x;
    const int64_t total = (int64_t)a.nlists * kHalfKeep;
    const uint64_t* src = a.keys + (int64_t)qi * total;
    __shared__ uint64_t sh[15 * 64];
    __shared__ uint64_t nominee[kHalfRescore];
    __shared__ uint64_t exact[kHalfRescore];
    __shared__ float ured[16];
    float u = a.thr0 ? a.thr0[qi] : -INFINITY;
    for (int b = threadIdx.x; b < a.nlists; b += 1024) {
        const uint64_t l = src[(int64_t)b * kHalfKeep + kHalfKeep - 1];
        if (l) u = fmaxf(u, key_score(l));
    }
    for (int off = 32; off; off >>= 1) u = fmaxf(u, __shfl_xor(u, off));
    if (lane == 0) ured[wave] = u;
    WaveTopK tk;
    tk.init(kHalfRescore);
    for (int64_t base = (int64_t)wave * 64; base < total; base += 1024) {
        const int64_t i = base + lane;
        tk.offer(i < total ? src[i] : 0ull);
    }
    if (a.base && wave == 0) tk.offer(lane < kHalfKeep ? a.base[(int64_t)qi * kHalfKeep + lane] : 0ull);
    block_merge_topk(tk, sh, 16);  // holds a __syncthreads(): ured is visible after it
    if (wave == 0) nominee[lane] = tk.key;
    __syncthreads();
    // wave w re-scores nominees w, w + 16, w + 32, w + 48: lane-strided 16-byte loads, butterfly sum (deterministic)
    for (int i = wave; i < kHalfRescore; i += 16) {
        const uint64_t key = nominee[i];
        uint64_t ek = 0ull;
        if (key) {
            const uint32_t row = key_row(key);
            const hs_f4* xr = reinterpret_cast<const hs_f4*>(a.X + (int64_t)row * a.ld);
            const hs_f4* qr = reinterpret_cast<const hs_f4*>(a.q + (int64_t)qi * a.ld);
            float s = 0.f;
            if (a.l2) {
                for (int c = lane; c < a.d4; c += 64) {
                    const hs_f4 t = qr[c] - xr[c];
                    s = fmaf(t[0], t[0], s);
                    s = fmaf(t[1], t[1], s);
                    s = fmaf(t[2], t[2], s);
                    s = fmaf(t[3], t[3], s);
                }
            } else {
                for (int c = lane; c < a.d4; c += 64) {
                    const hs_f4 x = xr[c], w = qr[c];
                    s = fmaf(x[0], w[0], s);
                    s = fmaf(x[1], w[1], s);
                    s = fmaf(x[2], w[2], s);
                    s = fmaf(x[3], w[3], s);
                }
            }
            for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
            ek = make_key(a.l2 ? -s : s, row);   // larger key = better: the smaller distance
        }
        if (lane == 0) exact[i] = ek;
    }
    __syncthreads();
    if (wave == 0) {
        u = ured[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) u = fmaxf(u, ured[w]);
        const uint64_t worst = nominee[kHalfRescore - 1];
        if (worst) u = fmaxf(u, key_score(worst));  // R is full: candidates left out of it score at most a(64th)
        const uint64_t mine = exact[lane];
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < kHalfRescore; ++j) rank += exact[j] > mine;
        if (mine && rank < a.k) {
            a.D[(int64_t)qi * a.k + rank] = a.l2 ? -key_score(mine) : key_score(mine);
            a.I[(int64_t)qi * a.k + rank] = a.label_offset + (int64_t)key_row(mine);
        }
        const int valid = __popcll(__ballot(mine != 0ull));
        if (lane >= valid && lane < a.k) {  // fewer rows than k: faiss' missing-result convention
            a.D[(int64_t)qi * a.k + lane] = a.l2 ? 3.402823466e+38f : -3.402823466e+38f;
            a.I[(int64_t)qi * a.k + lane] = -1;
        }
        // certification: needed as soon as any row was left out (u > -inf)
        const uint64_t holder = __ballot(mine && rank == a.k - 1);
        if (lane == 0 && a.floor_out) {
            float fl = -INFINITY;
            // (one ulp further down: a floor admits nothing that merely TIES with it, and with |q| = 0 the margin is 0 and every
            //  row ties with the k-th score)
            if (holder) {
                const float t = key_score(exact[__ffsll((long long)holder) - 1]);   // IP: the k-th exact score; L2: minus the k-th distance
                const float qn = a.qnorm[qi];
                if (!a.l2) {
                    fl = t - a.floor_margin * qn;
                } else {
                    // every row of the top k is at most r = -t away (up to the re-score's rounding): in the units of the keys
                    // that is a score of at least (|q|^2 + n2lo - r) / 2 (keys = q.x, rows of norm^2 >= n2lo) resp.
                    // (|q|^2 - r) / 2 - eps_h (keys = q.x - |x|^2 / 2); the factors are l2_certified's
                    const float n2 = a.l2 == 2 ? 0.f : a.n2lo;
                    fl = 0.5f * (fmaf(qn * 0.99999f, qn, n2) - (-t) * 1.00002f) - (a.l2 == 2 ? a.eps_h : 0.f) -
                         1e-6f * (qn * qn + fabsf(n2) + fabsf(t));
                }
                fl = nextafterf(fl, -INFINITY);
            }
            a.floor_out[qi] = fl == fl ? fl : -INFINITY;
        }
        if (lane == 0 && u > -INFINITY) {
            bool ok = false;
            if (holder) {
                const int hl = __ffsll((long long)holder) - 1;
                const float t = key_score(exact[hl]);
                ok = a.l2 == 2 ? l2_certified(-t, u, a.eps * a.qnorm[qi] + a.eps_h, a.qnorm[qi], 0.f)
                     : a.l2    ? l2_certified(-t, u, a.eps * a.qnorm[qi], a.qnorm[qi], a.n2lo)
                               : t > u + a.eps * a.qnorm[qi];
            }
            if (!ok) {
                atomicAdd(a.uncertified, 1);
                a.failed[qi] = 1;
            }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
// Worst-case bound, per unit |q| * max|x|, on |a(x) - q.x| plus the error of the fp32 re-score and of the
// certificate's own arithmetic (tests/test_split_bound.py restates and checks it):
//   (1) both operands rounded to fp16 (11 significant bits, RNE):  (1 + 2^-11)^2 - 1 = 2^-10 + 2^-22, summed with
//       Cauchy-Schwarz over the d products;
//   (2) elements of the scaled images below fp16's normal range (2^-14): absolute error <= 2^-14 each whether the
//       matrix cores keep subnormals or flush them; with the largest element of each image >= 2^14 that is
//       sqrt(d) 2^-27 (1 + 2^-11) + d 2^-56;
//   (3) fp32 accumulation of the d exact fp16 x fp16 products (22 significant bits each) and the NW - 1 <= 3
//       additions of the exchange, in ANY order, every addition rounded or truncated at fp32 width (unit 2^-23):
//       gamma(d + 4) (1 + 2^-11)^2;
//   (4) the fp32 re-score of the nominees and (5) |q| and the comparison, as in split_eps (mvdb.hip).
double half_eps(int d) {
    const double u11 = std::ldexp(1.0, -11), u23 = std::ldexp(1.0, -23), u24 = std::ldexp(1.0, -24);
    const double e_op = 2.0 * u11 + u11 * u11;
    const double e_uf = std::sqrt((double)d) * std::ldexp(1.0, -27) * (1.0 + u11) + d * std::ldexp(1.0, -56);
    const double n = d + 4.0;
    const double e_acc = n * u23 / (1.0 - n * u23) * (1.0 + u11) * (1.0 + u11);
    const double depth = ((d + 3) / 4 + 63) / 64 * 4 + 6;
    const double e_re = depth * u24 / (1.0 - depth * u24);
    return (e_op + e_uf + e_acc + e_re) * (1.0 + 4e-6) + 4.0 * u24;
}

// s_x = 2^(15 - e) with bound = m 2^e, m in [0.5, 1): every corpus element times s_x is below 2^15.
// 0 when the bound is outside 2^-40 .. 2^40 (the pass is not used then).
float half_xscale(float row_norm_bound) {
    if (!(row_norm_bound > 0.f) || !(row_norm_bound < 1.0e30f)) return 0.f;
    int e = 0;
    (void)std::frexp(row_norm_bound, &e);
    if (e < -40 || e > 40) return 0.f;
    return std::ldexp(1.f, 15 - e);
}

// widths the pass serves (d = 64 KQ for the seed launch, whose four waves split K evenly): those with a shadow kernel
static int half_kq(int d) { return half_shadow_dim(d) ? d / 64 : 0; }

int half_max_queries(int d) {
    if (!half_kq(d)) return 0;
    return d <= 512 ? 256 : 128;
}

int half_chunk_queries(int d, int nq) {
    const int mx = half_max_queries(d);
    if (!mx) return 0;
    return nq > 128 && mx >= 256 ? 256 : 128;
}

int launch_half_queries(const float* q, int64_t ld, int d, int nq, int nqpad, float xscale, _Float16* qf, float* qnorm,
                        float* qinv, hipStream_t stream) {
    int xexp = 0;
    (void)std::frexp(xscale, &xexp);  // xscale = 0.5 * 2^xexp
    hipLaunchKernelGGL(half_queries_kernel, dim3(nqpad), dim3(256), 0, stream, q, ld, d, nq, xexp - 1, qf, qnorm, qinv);
    MVDB_HIP(hipGetLastError());
    return 0;
}

// SEED launch of a pass: one 32-row tile per block over the fp32 rows (K split over the four waves, partial score tiles exchanged
// through LDS), every score dumped: [nq][blocks][32] keys.  It only has to produce the first admission floors.
template <int KQ, int NG, int NST>
static int launch_seed_inst(const HalfScanArgs& a, int device, hipStream_t stream, int* nblocks_out) {
    auto kern = flat_scan_seed_kernel<KQ, 2, NG, NST>;
    constexpr size_t lds = (size_t)4 * NST * HsStage<2>::kBytes + (size_t)4 * 3 * 4096;
    static_assert(lds + 128 <= 160 * 1024, "LDS budget of a CU");
    {
        static std::mutex mu;
        static std::map<int, bool> done;
        std::lock_guard<std::mutex> lk(mu);
        if (!done[device]) {
            MVDB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            done[device] = true;
        }
    }
    const int64_t ntiles = a.tile1 - a.tile0;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, (int64_t)device_cus(device)));
    *nblocks_out = nblocks;
    prof_symbol("ip_scan_half_seed", "flat_scan_seed_kernel<%d, 2, %d, %d>", KQ, NG, NST);
    int slot = prof_begin("ip_scan_half_seed", stream);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, stream, a);
    prof_end(slot, stream);
    MVDB_HIP(hipGetLastError());
    return 0;
}

template <int KQ>
static int launch_seed_kq(int nqpad, const HalfScanArgs& a, int device, hipStream_t stream, int* nb) {
    if (nqpad == 128) return launch_seed_inst<KQ, 4, 6>(a, device, stream, nb);
    if constexpr (KQ <= 8)
        if (nqpad == 256) return launch_seed_inst<KQ, 8, 5>(a, device, stream, nb);
    return fail(MVDB_ERR_ARG, "no seed kernel for %d queries per pass at d = %d", nqpad, KQ * 64);
}

// seed = true: the seed launch (fp32 rows); else a main launch over the fp16 shadow a.Xh (which the caller has ensured)
int launch_half_scan(int d, int nqpad, bool seed, const HalfScanArgs& a, const Knobs& kn, int device, hipStream_t stream, int* nblocks_out) {
    (void)kn;
    if (!seed) {
        if (!a.Xh || !half_shadow_dim(d)) return fail(MVDB_ERR_ARG, "internal: the certified pass needs the fp16 shadow (d = %d)", d);
        return launch_h16(d, nqpad, a, device, stream, nblocks_out);
    }
    switch (half_kq(d)) {
        case 2: return launch_seed_kq<2>(nqpad, a, device, stream, nblocks_out);
        case 4: return launch_seed_kq<4>(nqpad, a, device, stream, nblocks_out);
        case 6: return launch_seed_kq<6>(nqpad, a, device, stream, nblocks_out);
        case 8: return launch_seed_kq<8>(nqpad, a, device, stream, nblocks_out);
        case 10: return launch_seed_kq<10>(nqpad, a, device, stream, nblocks_out);
        case 12: return launch_seed_kq<12>(nqpad, a, device, stream, nblocks_out);
        case 14: return launch_seed_kq<14>(nqpad, a, device, stream, nblocks_out);
        case 16: return launch_seed_kq<16>(nqpad, a, device, stream, nblocks_out);
        default: return fail(MVDB_ERR_ARG, "no fp16 nomination kernel for d = %d", d);
    }
}

int launch_half_certify(const HalfCertifyArgs& a, int nq, hipStream_t stream) {
    hipLaunchKernelGGL(half_certify_kernel, dim3(nq), dim3(1024), 0, stream, a);
    MVDB_HIP(hipGetLastError());
    return 0;
}

}  // namespace mvdb
