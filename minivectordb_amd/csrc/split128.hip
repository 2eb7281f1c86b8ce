// split128.hip — main launch of the split-precision batch pass for 33..128 queries per corpus pass.
//
// Same nominate-and-certify scheme as scan_split_kernels.hpp (q.x ~ qh.xh + qh.xl + ql.xh on
// v_mfma_f32_32x32x16_bf16, 16 nominees per query, exact fp32 re-score + certificate in split_certify_kernel);
// what changes is WHERE the operands live.  The first 128-query kernel (flat_scan_split_kernel) stages corpus AND
// query fragments through LDS for every K-step and moves 8 bytes of LDS traffic per corpus byte — its DMA + LDS
// skeleton alone runs at the speed of the LDS port (3.3 ms at 10M x 512), before any MFMA.  Here the QUERIES stay
// in registers for the whole launch, as in the 32-query kernel, by splitting K across the four waves of a block:
//
//   wave w owns columns [w K/4, (w+1) K/4) of every row and ALL 128 queries: 128 x K/4 (hi, lo) bf16 fragments =
//   256 VGPRs at d = 512.  It streams its K-quarter of a 32-row tile through a private two-stage LDS-DMA ring
//   (8 KiB stages, one DMA write + one fragment read per corpus byte: 2 B of LDS traffic per corpus byte) and
//   accumulates a 32 x 128 PARTIAL score tile (4 accumulators of 32 x 32).  At the end of the tile the four
//   partial tiles are summed through LDS: wave w keeps query group w, hands the other three groups to their
//   owners (12 KiB out, 12 KiB in per wave: + 1.5 B per corpus byte), and gates the 32 finished scores per
//   lane (query on the lane, one threshold register) into its 32 nominee lists.
//
// No barrier inside a tile (the rings are wave-private); two bare s_barriers per 96 MFMAs around the exchange.
// Accumulation order differs from the other kernels (four K-quarter chains, then three fp32 adds) — the
// certificate's bound (mvdb.hip: split_eps) holds for ANY order of the 3 d products.
//
// Registers (d = 512): 256 query + 64 accumulator + 64 staged corpus fragments (two stages) + addresses: one
// wave per SIMD, 4 waves per CU, like the 32-query kernel; latency is hidden inside the wave (fragments of stage
// s + 1 are read under the MFMAs of stage s, two stages stay in flight in the ring).
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>

#include "common.hpp"
#include "split128.hpp"
#include "topk_device.hpp"

namespace mvdb {

typedef __bf16 s128_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s128_bf16x2 __attribute__((ext_vector_type(2)));
typedef float s128_f32x2 __attribute__((ext_vector_type(2)));
typedef float s128_f32x4 __attribute__((ext_vector_type(4)));
typedef float s128_f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* s128_lds_ptr;
typedef const __attribute__((address_space(1))) void* s128_gbl_ptr;

constexpr int kS128Keep = 16;                 // nominees per query (= kSplitKeep)
constexpr int kS128Stage = 8192;              // 32 rows x 256 B
constexpr int kS128Ring = 3 * kS128Stage;     // per wave: three stages in flight
constexpr int kS128Exch = 4 * 3 * 4 * 1024;   // [dest wave][source slot][register quad][lane x 16 B]
constexpr size_t kS128Lds = 4 * kS128Ring + kS128Exch;

__device__ __forceinline__ uint32_t s128_pack(float a, float b) {  // (lo: bf16(a), hi: bf16(b)), RNE
    union { s128_bf16x2 v; uint32_t u; } c;
    c.v = __builtin_convertvector(s128_f32x2{a, b}, s128_bf16x2);
    return c.u;
}

// KQ = 16-element K blocks per wave (K / 64), SKB = K blocks per ring stage (stage = 32 rows x SKB x 64 B, laid out
// with a 256-byte row pitch whatever SKB is).
// DBG != 0: timing ablations (MVDB_SPLIT128_DBG; results invalid): 1 no exchange, 2 no MFMA, 4 no corpus DMA,
// 8 no s_barrier (exchange traffic kept), 16 default cache policy instead of nt, 32 no nomination (scores kept alive),
// 64 nomination code present but never entered (floors = +inf), 128 every DMA re-reads the block's first tile (L2 hits),
// 256 no vmcnt waits inside the loop (races), 1024 DMA instructions issued back to back (not spread between MFMAs), 2048 s_memtime instrumentation (MVDB_SPLIT_STATS prints it),
// 512 exchange replaced by a register sum (no LDS traffic, no barriers)
template <int KQ, int SKB, int DBG = 0>
__global__ __launch_bounds__(256) void flat_scan_split128_kernel(Split128Args a) {
    static_assert(KQ % SKB == 0 && SKB >= 1 && SKB <= 4, "stage shape");
    constexpr int NS = KQ / SKB;  // stages per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // rings | exchange
    __shared__ uint64_t lists[128 * kS128Keep];                            // [query][16] keys, wave w owns 32 w ..
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 31, fk = lane >> 5;
    unsigned char* wbuf = smem + (size_t)wave * kS128Ring;
    unsigned char* exch = smem + 4 * kS128Ring;
    uint64_t* mylists = lists + (size_t)wave * 32 * kS128Keep;
    for (int e = lane; e < 32 * kS128Keep; e += 64) mylists[e] = 0ull;

    // ---- query fragments of this wave's K-quarter.  Local group jj is physical query group (wave + jj) & 3, so
    // that the group this wave OWNS is always accumulator 0 (no run-time register indexing).
    s128_bf16x8 qh[KQ][4], ql[KQ][4];
#pragma unroll
    for (int kb = 0; kb < KQ; ++kb) {
        const int blk = wave * KQ + kb;  // 16-element block of the row
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int query = ((wave + jj) & 3) * 32 + fr;
            const int64_t o = ((int64_t)(blk >> 1) * 128 + query) * 32 + (blk & 1) * 16 + fk * 8;
            qh[kb][jj] = *reinterpret_cast<const s128_bf16x8*>(a.qh + o);
            ql[kb][jj] = *reinterpret_cast<const s128_bf16x8*>(a.ql + o);
        }
    }
    const int myq = wave * 32 + fr;
    float floor0 = myq < a.nq ? (a.thr0 ? a.thr0[myq] : -INFINITY) : INFINITY;
    if (DBG & 64) floor0 = INFINITY;
    float thr = floor0;
    // consume every global load here: the hand-placed vmcnt waits below are invisible to hipcc, a first use inside
    // the loop would get a compiler-made s_waitcnt vmcnt(0) that also drains the DMA ring
#pragma unroll
    for (int kb = 0; kb < KQ; ++kb)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) asm volatile("" : "+v"(qh[kb][jj]), "+v"(ql[kb][jj]));
    asm volatile("" : "+v"(floor0), "+v"(thr));

    const int64_t ntiles = a.tile1 - a.tile0;
    const int64_t last = a.n - 1;
    // ---- DMA roles: instruction i (0..7) of a stage moves rows 4i .. 4i+3; lane -> row 4i + (lane >> 4), 16-byte slot
    // lane & 15 of the 256-byte LDS row, which receives the row's LOGICAL slot p ^ (r & 15) (bank swizzle on the
    // source).  Source address = wave-uniform base of (tile, stage) + a per-lane byte offset that never changes:
    // the offsets live in 8 registers and an issue costs scalar arithmetic only.
    const int dma_r = lane >> 4, dma_p = lane & 15;
    const int col0 = wave * KQ * 16;
    uint32_t voff[8];
    bool vact[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = 4 * i + dma_r;
        const int slot = dma_p ^ (r & 15);
        voff[i] = (uint32_t)(((int64_t)r * a.ld + 4 * slot) * 4);
        vact[i] = SKB == 4 || slot < 4 * SKB;
    }
    // Rows past the end of the corpus (last tile) are READ like any other — the index keeps 32 rows of slack behind
    // row n - 1 (mvdb.hip: grow) — and never nominated; that keeps the issue straight-line code the scheduler can
    // spread between MFMAs.
    auto issue_stage = [&](int64_t tile, int ks, int buf) {
        if (DBG & 4) return;
        const int64_t row0 = (a.tile0 + ((DBG & 128) ? (int64_t)blockIdx.x : tile)) * 32;
        const char* sbase = reinterpret_cast<const char*>(a.X + row0 * a.ld + col0 + ks * SKB * 16);
        unsigned char* dst = wbuf + buf * kS128Stage;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (vact[i])
                __builtin_amdgcn_global_load_lds((s128_gbl_ptr)(sbase + voff[i]), (s128_lds_ptr)(dst + i * 1024), 16, 0, (DBG & 16) ? 0 : 2 /* nt */);
    };
    // fragment read: row fr, 16-k block b of the stage (k = 16 b + 8 fk .. + 7) -> logical slots 4 b + 2 fk (+1)
    int f_off[SKB][2];
#pragma unroll
    for (int b = 0; b < SKB; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) f_off[b][h] = fr * 256 + (((4 * b + 2 * fk + h) ^ (fr & 15)) << 4);

    s128_f32x16 acc[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;

    const int64_t step = gridDim.x;
    int64_t tile = blockIdx.x;
    // stage g of this block's flat sequence: tile + (g / NS) * step, K stage g % NS
    auto st_tile = [&](int64_t t, int ks_abs) { return t + (int64_t)(ks_abs / NS) * step; };
    struct Frag {
        s128_f32x4 v[SKB][2];
    };
    union Op {
        s128_bf16x8 v;
        uint32_t w[4];
    };
    Frag fx[2];        // staged corpus fragments of two stages (ping-pong)
    Op ah[2], al[2];   // (hi, lo) bf16 operands of two K blocks (ping-pong): block b+1 is split under the MFMAs of block b
    auto read_frags = [&](Frag& dst, int buf) {
        const unsigned char* sb = wbuf + buf * kS128Stage;
#pragma unroll
        for (int b = 0; b < SKB; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) dst.v[b][h] = *reinterpret_cast<const s128_f32x4*>(sb + f_off[b][h]);
    };
    auto split_block = [&](const s128_f32x4 (&x)[2], Op& hi, Op& lo) {  // 6 VALU per element pair
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = x[e >> 1][(e & 1) * 2], x1 = x[e >> 1][(e & 1) * 2 + 1];
            const uint32_t hp = s128_pack(x0, x1);
            hi.w[e] = hp;
            lo.w[e] = s128_pack(x0 - __uint_as_float(hp << 16), x1 - __uint_as_float(hp & 0xFFFF0000u));
        }
    };
    auto mfma_block = [&](int kb, const Op& hi, const Op& lo) {
        if (DBG & 2) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                acc[jj][0] += (float)lo.v[0] * (float)qh[kb][jj][1] + (float)hi.v[2] * (float)ql[kb][jj][3];
            return;
        }
        // small cross terms first, the leading product last
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo.v, qh[kb][jj], acc[jj], 0, 0, 0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi.v, ql[kb][jj], acc[jj], 0, 0, 0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi.v, qh[kb][jj], acc[jj], 0, 0, 0);
    };

    unsigned n_ins = 0, n_slow = 0;
    long long tm_vm = 0, tm_issue = 0, tm_stage = 0, tm_last = 0;  // DBG 2048: cycle counters (s_memtime)
    // ring: stage g of the flat sequence lives in buffer g % 3; `rb` = buffer of the stage after the one in registers
    int rb = 1;
    // Stages past the block's last tile are issued too (clamped to its current tile, landing in buffers nobody reads):
    // the loop body is branch-free, every counted wait sees a full ring.
    auto clamp_tile = [&](int64_t t, int64_t fallback) { return t < ntiles ? t : fallback; };
    if (tile < ntiles) {
        issue_stage(tile, 0, 0);
        issue_stage(clamp_tile(st_tile(tile, 1), tile), 1 % NS, 1);
        issue_stage(clamp_tile(st_tile(tile, 2), tile), 2 % NS, 2);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        read_frags(fx[0], 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        issue_stage(clamp_tile(st_tile(tile, 3), tile), 3 % NS, 0);
        __builtin_amdgcn_sched_barrier(0);
        split_block(fx[0].v[0], ah[0], al[0]);
    }

    // One tile: NS stages of SKB blocks.  R0 / P0 (compile time) say which fragment set / operand set the tile starts in.
    auto tile_body = [&](auto R0c, auto P0c) {
        constexpr int R0 = decltype(R0c)::value, P0 = decltype(P0c)::value;
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            constexpr int dummy = 0;
            (void)dummy;
            const int R = (R0 + ks) & 1;
            constexpr int unused_ = 0;
            (void)unused_;
            // here: fx[R] = stage (tile, ks); ring buffers rb, rb+1, rb+2 (mod 3) = the next three stages
#pragma unroll
            for (int b = 0; b < SKB; ++b) {
                const int P = (P0 + ks * SKB + b) & 1;
                const int kb = ks * SKB + b;
                if (b == (SKB >= 2 ? SKB - 2 : 0)) {
                    // the next stage's fragments are read under the MFMAs of this block: its DMAs were issued three
                    // stages ago, the two stages after it stay in flight
                    __builtin_amdgcn_sched_barrier(0);
                    long long t2 = 0;
                    if (DBG & 2048) t2 = __builtin_readcyclecounter();
                    if (!(DBG & 256)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    if (DBG & 2048) tm_vm += __builtin_readcyclecounter() - t2;
                    __builtin_amdgcn_sched_barrier(0);
                    read_frags(fx[R ^ 1], rb);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (b == SKB - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments of the next stage are in registers
                    __builtin_amdgcn_sched_barrier(0);
                    long long t0 = 0;
                    if (DBG & 2048) {
                        t0 = __builtin_readcyclecounter();
                        if (tm_last) tm_stage += t0 - tm_last;
                        tm_last = t0;
                    }
                    // refill the buffer just drained
                    issue_stage(clamp_tile(st_tile(tile, ks + 4), tile), (ks + 4) % NS, rb);
                    if (DBG & 2048) tm_issue += __builtin_readcyclecounter() - t0;
                    if (DBG & (2048 | 1024)) __builtin_amdgcn_sched_barrier(0);  // else: spread between this block's MFMAs
                }
                // this block's 12 MFMAs; the split of the FOLLOWING block rides in their shadow
                if (b < SKB - 1)
                    split_block(fx[R].v[b + 1], ah[P ^ 1], al[P ^ 1]);
                else
                    split_block(fx[R ^ 1].v[0], ah[P ^ 1], al[P ^ 1]);
                mfma_block(kb, ah[P], al[P]);
                if (!(DBG & 2)) {
#pragma unroll
                    for (int t = 0; t < 12; ++t) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                        if (b == SKB - 1 && t >= 2 && t <= 9 && !(DBG & (2048 | 1024 | 4))) {
                            __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);  // M0 / base arithmetic
                            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // 1 DMA
                        }
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // up to 4 VALU in its shadow
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            rb = rb == 2 ? 0 : rb + 1;
        }
        // ---- tile end: sum the four K-quarter partial tiles.  Wave w keeps group w (accumulator 0) and hands local
        // group jj to wave (w + jj) & 3, which finds it in its source slot jj - 1.
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & (9 | 512))) __builtin_amdgcn_s_barrier();  // every wave has read what the previous tile left in the exchange area
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & (1 | 512))) {
#pragma unroll
            for (int jj = 1; jj < 4; ++jj) {
                unsigned char* dst = exch + (((wave + jj) & 3) * 3 + (jj - 1)) * 4096 + lane * 16;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    *reinterpret_cast<s128_f32x4*>(dst + r4 * 1024) =
                        s128_f32x4{acc[jj][4 * r4], acc[jj][4 * r4 + 1], acc[jj][4 * r4 + 2], acc[jj][4 * r4 + 3]};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & (9 | 512))) __builtin_amdgcn_s_barrier();  // all partial tiles are in LDS
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned char* src = exch + wave * 3 * 4096 + lane * 16;
            s128_f32x4 part[3][4];
            float sc[16];
            if (DBG & 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = acc[0][r];
            } else if (DBG & 512) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
            } else {
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) part[s][r4] = *reinterpret_cast<const s128_f32x4*>(src + s * 4096 + r4 * 1024);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sc[r] = ((acc[0][r] + part[0][r >> 2][r & 3]) + part[1][r >> 2][r & 3]) + part[2][r >> 2][r & 3];
            }
            // ---- nomination: D[row][query], query on the lane (fr), rows in the 16 registers
            const int64_t m0 = (a.tile0 + tile) * 32;
            float mx = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
            if (DBG & 32) {
                if (mx == 1.2345f) thr = 0.f;
            } else if (__ballot(mx >= thr) != 0ull) {
                ++n_slow;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = (r & 3) + 8 * (r >> 2);
                    const float s = sc[r];
                    uint64_t mask = __ballot(m0 + rl + 4 * fk <= last && s >= thr);
                    while (mask) {
                        const int srcl = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        ++n_ins;
                        const int sq = srcl & 31;
                        const float sv = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(s), srcl));
                        const uint32_t rv = (uint32_t)(m0 + rl + 4 * (srcl >> 5));
                        const uint64_t kth = lds_list_insert(mylists + (size_t)sq * kS128Keep, kS128Keep, make_key(sv, rv), lane);
                        if (fr == sq) thr = kth ? fmaxf(key_score(kth), floor0) : floor0;  // both lane halves
                    }
                }
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;
        tile += step;
    };
    // two tiles per trip: an odd number of stages (or of blocks) per tile flips the ping-pong roles every tile
    while (tile < ntiles) {
        tile_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if (tile >= ntiles) break;
        tile_body(std::integral_constant<int, NS & 1>{}, std::integral_constant<int, KQ & 1>{});
    }
    // every query's list lives in exactly one wave: write the block's nominee lists
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.stats && lane == 0) {
        atomicAdd(a.stats, n_ins);
        atomicAdd(a.stats + 1, n_slow);
        if ((DBG & 2048) && blockIdx.x == 7) {
            unsigned long long* t = reinterpret_cast<unsigned long long*>(a.stats + 2) + wave * 3;
            atomicAdd(t, (unsigned long long)tm_vm);
            atomicAdd(t + 1, (unsigned long long)tm_issue);
            atomicAdd(t + 2, (unsigned long long)tm_stage);
        }
    }
    for (int q = 0; q < 32; ++q) {
        const int qq = wave * 32 + q;
        if (qq >= a.nq) break;
        if (lane < kS128Keep)
            a.cand[((int64_t)qq * gridDim.x + blockIdx.x) * kS128Keep + lane] = mylists[(size_t)q * kS128Keep + lane];
    }
}

// d = 512 only: at 384 / 256 the first 128-query kernel (scan_split_kernels.hpp) is faster (3.42 vs 3.85 ms and 2.30 vs
// 2.51 ms per 128 queries over 10M rows: masked DMA lanes and single-stage tiles), see DESIGN.md section 4.3b''.
bool split128_supported(int d) { return d == 512; }

template <int KQ, int SKB, int DBG = 0>
static int launch_inst(const Split128Args& a, int device, hipStream_t stream, int* nblocks_out) {
    auto kern = flat_scan_split128_kernel<KQ, SKB, DBG>;
    {
        static std::mutex mu;
        static std::map<int, bool> done;
        std::lock_guard<std::mutex> lk(mu);
        if (!done[device]) {
            MVDB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kS128Lds));
            done[device] = true;
        }
    }
    const int64_t ntiles = a.tile1 - a.tile0;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, (int64_t)device_cus(device)));
    *nblocks_out = nblocks;
    int slot = prof_begin("ip_scan_split", stream);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), kS128Lds, stream, a);
    prof_end(slot, stream);
    MVDB_HIP(hipGetLastError());
    return 0;
}

int launch_split128(int d, const Split128Args& a, int device, hipStream_t stream, int* nblocks_out) {
    if (d != 512) return fail(MVDB_ERR_ARG, "no K-split 128-query kernel for d = %d", d);
    const char* e = getenv("MVDB_SPLIT128_DBG");  // timing ablations, benchmarks/split128_probe.py
    switch (e ? atoi(e) : 0) {
        case 2: return launch_inst<8, 4, 2>(a, device, stream, nblocks_out);
        case 4: return launch_inst<8, 4, 4>(a, device, stream, nblocks_out);
        case 64: return launch_inst<8, 4, 64>(a, device, stream, nblocks_out);
        case 68: return launch_inst<8, 4, 68>(a, device, stream, nblocks_out);
        case 576: return launch_inst<8, 4, 576>(a, device, stream, nblocks_out);
        case 580: return launch_inst<8, 4, 580>(a, device, stream, nblocks_out);
        case 1024: return launch_inst<8, 4, 1024>(a, device, stream, nblocks_out);
        case 2112: return launch_inst<8, 4, 2112>(a, device, stream, nblocks_out);
        case 2116: return launch_inst<8, 4, 2116>(a, device, stream, nblocks_out);
        default: break;
    }
    return launch_inst<8, 4>(a, device, stream, nblocks_out);
}

}  // namespace mvdb
