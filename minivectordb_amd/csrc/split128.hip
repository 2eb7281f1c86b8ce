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
#include <map>
#include <mutex>

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
constexpr int kS128Ring = 2 * kS128Stage;     // per wave
constexpr int kS128Exch = 4 * 3 * 4 * 1024;   // [dest wave][source slot][register quad][lane x 16 B]
constexpr size_t kS128Lds = 4 * kS128Ring + kS128Exch;

__device__ __forceinline__ uint32_t s128_pack(float a, float b) {  // (lo: bf16(a), hi: bf16(b)), RNE
    union { s128_bf16x2 v; uint32_t u; } c;
    c.v = __builtin_convertvector(s128_f32x2{a, b}, s128_bf16x2);
    return c.u;
}

// KQ = 16-element K blocks per wave (K / 64), SKB = K blocks per ring stage (stage = 32 rows x SKB x 64 B, laid out
// with a 256-byte row pitch whatever SKB is)
template <int KQ, int SKB>
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
    // DMA roles: instruction i (0..7) of a stage moves rows 4i .. 4i+3; lane -> row 4i + (lane >> 4), 16-byte slot
    // lane & 15 of the 256-byte LDS row, which receives the row's LOGICAL slot p ^ (r & 15) (bank swizzle on the source)
    const int dma_r = lane >> 4, dma_p = lane & 15;
    const int col0 = wave * KQ * 16;
    auto issue_stage = [&](int64_t tile, int ks, int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 4 * i + dma_r;
            const int slot = dma_p ^ (r & 15);
            int64_t row = (a.tile0 + tile) * 32 + r;
            row = row <= last ? row : last;
            const float* src = a.X + row * a.ld + col0 + ks * SKB * 16 + 4 * slot;
            if (SKB == 4 || slot < 4 * SKB)
                __builtin_amdgcn_global_load_lds((s128_gbl_ptr)src, (s128_lds_ptr)(wbuf + buf * kS128Stage + i * 1024), 16, 0,
                                                 2 /* nt */);
        }
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
    unsigned cnt = 0;
    s128_f32x4 xa[SKB][2], xn[SKB][2];
    auto read_frags = [&](s128_f32x4 (&dst)[SKB][2], int buf) {
        const unsigned char* sb = wbuf + buf * kS128Stage;
#pragma unroll
        for (int b = 0; b < SKB; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) dst[b][h] = *reinterpret_cast<const s128_f32x4*>(sb + f_off[b][h]);
    };
    if (tile < ntiles) {
        issue_stage(tile, 0, 0);
        const int64_t t1 = st_tile(tile, 1);
        if (t1 < ntiles) {
            issue_stage(t1, 1 % NS, 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        read_frags(xa, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int64_t t2 = st_tile(tile, 2);
        if (t2 < ntiles) issue_stage(t2, 2 % NS, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    while (tile < ntiles) {
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            // here: xa = stage (tile, ks); ring buffer (cnt + 1) & 1 = next stage, buffer cnt & 1 = the one after
            const int nbuf = (cnt + 1) & 1;
            const int64_t t1 = st_tile(tile, ks + 1), t2 = st_tile(tile, ks + 2), t3 = st_tile(tile, ks + 3);
#pragma unroll
            for (int b = 0; b < SKB; ++b) {
                const int kb = ks * SKB + b;
                union { s128_bf16x8 v; uint32_t w[4]; } ahu, alu;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = xa[b][e >> 1][(e & 1) * 2], x1 = xa[b][e >> 1][(e & 1) * 2 + 1];
                    const uint32_t hp = s128_pack(x0, x1);
                    ahu.w[e] = hp;
                    alu.w[e] = s128_pack(x0 - __uint_as_float(hp << 16), x1 - __uint_as_float(hp & 0xFFFF0000u));
                }
                if (b == SKB - 1) {
                    // the next stage's fragments are read under this block's 12 MFMAs: its DMAs were issued a
                    // whole stage ago; the stage after it stays in flight
                    __builtin_amdgcn_sched_barrier(0);
                    if (t1 < ntiles) {
                        if (t2 < ntiles)
                            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        read_frags(xn, nbuf);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // small cross terms first, the leading product last
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alu.v, qh[kb][jj], acc[jj], 0, 0, 0);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahu.v, ql[kb][jj], acc[jj], 0, 0, 0);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahu.v, qh[kb][jj], acc[jj], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t1 < ntiles) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments of the next stage are in registers
                __builtin_amdgcn_sched_barrier(0);
                if (t3 < ntiles) issue_stage(t3, (ks + 3) % NS, nbuf);  // refill the buffer just drained
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < SKB; ++b)
#pragma unroll
                    for (int h = 0; h < 2; ++h) xa[b][h] = xn[b][h];
            }
            ++cnt;
        }
        // ---- tile end: sum the four K-quarter partial tiles.  Wave w keeps group w (accumulator 0) and hands local
        // group jj to wave (w + jj) & 3, which finds it in its source slot jj - 1.
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // every wave has read what the previous tile left in the exchange area
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 1; jj < 4; ++jj) {
            unsigned char* dst = exch + (((wave + jj) & 3) * 3 + (jj - 1)) * 4096 + lane * 16;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<s128_f32x4*>(dst + r4 * 1024) =
                    s128_f32x4{acc[jj][4 * r4], acc[jj][4 * r4 + 1], acc[jj][4 * r4 + 2], acc[jj][4 * r4 + 3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // all partial tiles are in LDS
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned char* src = exch + wave * 3 * 4096 + lane * 16;
            s128_f32x4 part[3][4];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) part[s][r4] = *reinterpret_cast<const s128_f32x4*>(src + s * 4096 + r4 * 1024);
            float sc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sc[r] = ((acc[0][r] + part[0][r >> 2][r & 3]) + part[1][r >> 2][r & 3]) + part[2][r >> 2][r & 3];
            // ---- nomination: D[row][query], query on the lane (fr), rows in the 16 registers
            const int64_t m0 = (a.tile0 + tile) * 32;
            float mx = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
            if (__ballot(mx >= thr) != 0ull) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = (r & 3) + 8 * (r >> 2);
                    const float s = sc[r];
                    uint64_t mask = __ballot(m0 + rl + 4 * fk <= last && s >= thr);
                    while (mask) {
                        const int srcl = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
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
    }
    // every query's list lives in exactly one wave: write the block's nominee lists
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int q = 0; q < 32; ++q) {
        const int qq = wave * 32 + q;
        if (qq >= a.nq) break;
        if (lane < kS128Keep)
            a.cand[((int64_t)qq * gridDim.x + blockIdx.x) * kS128Keep + lane] = mylists[(size_t)q * kS128Keep + lane];
    }
}

bool split128_supported(int d) { return d == 512 || d == 384 || d == 256 || d == 128 || d == 64; }

template <int KQ, int SKB>
static int launch_inst(const Split128Args& a, int device, hipStream_t stream, int* nblocks_out) {
    auto kern = flat_scan_split128_kernel<KQ, SKB>;
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
    switch (d) {
        case 512: return launch_inst<8, 4>(a, device, stream, nblocks_out);
        case 384: return launch_inst<6, 3>(a, device, stream, nblocks_out);
        case 256: return launch_inst<4, 4>(a, device, stream, nblocks_out);
        case 128: return launch_inst<2, 2>(a, device, stream, nblocks_out);
        case 64: return launch_inst<1, 1>(a, device, stream, nblocks_out);
        default: return fail(MVDB_ERR_ARG, "no 128-query split kernel for d = %d", d);
    }
}

}  // namespace mvdb
