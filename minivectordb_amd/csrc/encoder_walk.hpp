// encoder_walk.hpp — the whole encoder forward of a SMALL batch as ONE launch that walks the layers itself.
//
// Reference shape: extract_embeddings(text) tokenises ONE sentence per call (minivectordb/embedding_model.py:62-71), i.e.
// 4 .. 60 tokens.  As a chain of per-op kernels that forward is ~76 launches of 4.5-9 us each (0.60 ms, round 4): the
// arithmetic (2.7 GFLOP at 64 tokens) and the weight bytes (85 MB, Infinity-Cache resident) are a few tens of us.  Here
// one persistent launch of G <= #CU workgroups (one per CU, all resident) runs
//     embeddings + LN | per layer: [QKV |] attention + out-proj partials | sum + LN | FFN partials | sum + LN | pooling
// as a chain of PHASES.  Each phase has its own monotonic arrival counter: the workgroups that produce in a phase add to it
// once their (write-through, `sc1`) stores have drained, and a workgroup polls a counter only in front of a phase it has work
// in — then one agent-scope acquire, and plain loads (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement &
// inter-workgroup visibility": "ONE relaxed poll -> ONE agent acquire -> s_waitcnt vmcnt(0) -> __syncthreads() -> plain
// loads").  A workgroup with nothing to do in a phase neither arrives nor polls: it goes straight to the wait in front of
// its next phase, with that phase's weights already on their way (weights do not depend on other workgroups, so every
// phase issues its weight / bias loads BEFORE it polls).  Buffers are reused safely because the phases form a chain: a
// phase's producers start only when ALL producers of the phase before have arrived, so whoever overwrites a buffer has,
// transitively, waited for every reader of its previous contents.
//
// Arithmetic: exact fp32 on v_mfma_f32_16x16x4_f32 (the parity mode's arithmetic; at <= 64 tokens no phase is bound by the
// matrix pipe, so the split-precision mode would buy nothing).  Op order as encoder.hip's header (modeling_bert.py).
//
// Work decomposition (T <= 64 packed tokens = MT <= 4 row tiles of 16; H hidden, F intermediate, hd head width):
//   * a "column unit" = 16 output columns of a [T,K]x[N,K]^T product, K split over the 8 waves of a workgroup in chunks of
//     16 (lane l holds W[n0 + (l&15)][16c + 4(l>>4) .. +3] and X[row][same k]: one float4 each, the 4 components feed 4 MFMAs
//     that contract k in {16c + 4g + m}); the wave partials meet in LDS and are added in wave order (deterministic).
//   * weights are the A operand (i = output column), activations the B operand (j = token): the accumulator of lane l then
//     holds 4 CONSECUTIVE output columns of one token -> float4 epilogues and stores.
//   * attention: one workgroup per (sentence, head[, column split of the out-projection]); S^T = K Q^T lands with the key
//     index on the accumulator registers, exactly where the next product (ctx^T = V^T P^T, summing over keys) wants its B
//     operand: no lane movement between the two products.  The head's context never leaves LDS: the workgroup multiplies
//     it by its hd columns of W_o and writes a [T,H] PARTIAL plane; planes are summed (fixed order) with bias and residual
//     by the row-owning workgroups of the next phase, which also apply the LayerNorm.  FFN the same way: a workgroup owns
//     16-wide slices of F, computes GELU(x W1_slice^T + b1) into LDS and multiplies by W2[:, slice] into its plane.
//   * a sentence's result does not depend on what else is in the batch or where its rows sit (every product of a row is the
//     same k-ordered chain; attention indexes keys from the sentence's own first token).
#pragma once

namespace walk {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kTmax = 128;    // packed tokens (and token slots B * S) per launch
constexpr int kSc1 = 16;      // cache-policy bit of the buffer builtins on gfx950: sc1
constexpr int kMaxPlanes = 128;
constexpr int kLdsHead = 4 * kTmax + 16;  // floats in front of the phase scratch: packing tables [3][kTmax], sentence starts [kTmax + 1 (+ pad)], 8 reduction slots

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct LayerPtrs {
    const float *wqkv, *bqkv, *wo, *bo, *ln1g, *ln1b, *w1, *b1, *w2, *b2, *ln2g, *ln2b;
};

struct Args {
    const int32_t *ids, *mask;  // [B, S]
    int B, S;
    int H, F, heads, hd, nlayers, position_offset, vocab, pooling;
    float eps;
    const float *word, *pos, *type, *embg, *embb;
    const LayerPtrs* layers;  // device array [nlayers]
    float *X, *X1;            // [kTmax, H]   layer input / post-attention state
    float* QKV;               // [kTmax, 3H]
    float* PL;                // [planes, kTmax, H] partial planes
    float* Hb;                // [kTmax, F] GELU(x1 W1^T + b1) (wide shapes only: FFN as two phases)
    unsigned int* bar;        // [kCtrCount][8 replicas][32 words] arrival counters of the phases (monotonic within a launch) + exits
    unsigned int* flag;       // the encoder's overflow word: cleared by the launch (exact fp32: nothing overflows into it)
    float* out;               // [B, H]
    float* hidden;            // NULL or [B, S, H]
    int np3;                  // workgroups (= planes) of the FFN phase
    int nsplit;               // column splits of the out-projection per (sentence, head)
    int off_rows, off_qkv, off_attn, off_ffn;  // first workgroup of the row phases (embeddings, sum + LN, pooling) / QKV / attention / FFN
    unsigned long long* trace;  // NULL, or [G][kTraceSlots] s_memrealtime stamps (ablation build: mvdb_debug_walk_trace)
    // BOUNDED WAITS.  No wait of the launch outlasts `deadline` ticks of s_memrealtime (100 MHz) counted from the waiting
    // workgroup's own start: the poller that sees it expire raises kAbortBit in every phase counter, which releases every
    // other poller (a counter with the bit set is >= any target), every workgroup leaves the phase chain at its next wait,
    // and the last one out counts the aborted launch in `aborts` (device word), mirrors the count into `aborts_host`
    // (host-mapped) and fills `out` with NaN before it re-arms the counters.  The host entries then run the per-op kernels.
    unsigned int deadline;      // ticks
    unsigned int* aborts;       // device word: launches of this encoder that were aborted
    unsigned int* aborts_host;  // host-mapped mirror of the count (read by the host entry behind its stream wait)
};
constexpr int kTraceSlots = 512;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// Loads of handed-off bytes: sc1 (L1-bypassing), so that no acquire is needed in front of them — the hand-off is the first
// row of MI355X_MICROARCH.md's table of hand-offs measured with sc1 loads in place of the acquire: one lane per storing
// workgroup adds to the counter behind the workgroup's drained sc1 stores, an sc1 poll, the other waves load behind a
// workgroup barrier, hipMalloc memory, one workgroup per CU, 16-byte sc1 stores and loads.  A/B (MVDB_WALK_ACQUIRE_LOADS
// build: one agent-scope acquire per wait, then plain loads): 13 / 16 / 62 us slower per forward at 8 / 32 / 64 tokens
// (profiles/r05_walk_ab.txt) — the acquire costs ~0.7 us per wait and the activations are read once per workgroup anyway.
__device__ __forceinline__ f32x4 ld4(rsrc_t r, int byte_off) {
#ifdef MVDB_WALK_ACQUIRE_LOADS
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
#else
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, kSc1));
#endif
}
__device__ __forceinline__ void st4(rsrc_t r, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, kSc1);
}

// ---- phase counters ------------------------------------------------------------------------------------------------------
enum { kCtrEmbed = 0, kCtrQkv, kCtrAttn, kCtrLn1, kCtrFfn1, kCtrFfn, kCtrLn2, kCtrExit, kCtrCount };

// stamps of (layer, phase): wait begins, released, arrived — ablation build only (a.trace != NULL)
__device__ __forceinline__ void stamp(unsigned long long* trace, int layer, int phase, int what) {
    if (trace && threadIdx.x == 0) {
        const int slot = (layer * 6 + phase) * 3 + what;
        if (slot < kTraceSlots) trace[slot] = __builtin_amdgcn_s_memrealtime();
    }
}

// Every counter is kept in kReplicas copies, each on a 128-byte line of its own: an arriving workgroup adds to ALL of them
// with ONE wave instruction (lane i -> replica i), a waiting workgroup polls replica (workgroup & 7).  (One word: 96 pollers
// were released over 1.8 us, first to last, and 96 arrivals queued on it — replicas cut the hand-offs from ~3 to 1.1 - 3.7 us.
// Tried instead, and slower by 8 - 16 us per forward: eight SHARDS — one atomic per arrival on shard (workgroup & 7), every
// waiter polling all eight with one 8-lane load — an eighth of the atomics but eight lines per poll.)
constexpr int kReplicas = 8, kCtrStride = 32;  // words
__device__ __forceinline__ unsigned int* ctr_word(unsigned int* bar, int ctr, int replica) {
    return bar + (ctr * kReplicas + replica) * kCtrStride;
}

// This workgroup has produced its share of a phase: every wave drains its own stores, the workgroup meets, eight lanes add.
__device__ __forceinline__ void phase_arrive(unsigned int* bar, int ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < kReplicas) __hip_atomic_fetch_add(ctr_word(bar, ctr, threadIdx.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A counter word with this bit set releases every waiter: the launch is being abandoned (Args::deadline).
constexpr unsigned int kAbortBit = 0x80000000u;
struct WaitCtx {
    unsigned long long t0;    // s_memrealtime when this workgroup started
    unsigned int deadline;    // ticks (100 MHz)
    unsigned int check_mask;  // the clock is read on every failed poll whose count & check_mask == 0: 1023 in production (a wait
                              // of a healthy launch fails a handful of polls and never reads it: reading it on every failed
                              // poll cost 2 - 4 % of a forward), 0 for deadlines under 1 ms (tests)
    int* lds_abort;           // one word of LDS, 0 until a wait of this workgroup was released by kAbortBit
};
__device__ __forceinline__ void raise_abort(unsigned int* bar) {
    for (int c = 0; c < kCtrExit; ++c)
        for (int rep = 0; rep < kReplicas; ++rep)
            __hip_atomic_fetch_or(ctr_word(bar, c, rep), kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait until the `producers` workgroups of a phase have each arrived `epochs` times at counter `ctr`: one lane polls its
// replica with L1-bypassing loads, the workgroup meets.  Returns true (to every thread) when the launch is being abandoned:
// the caller leaves the phase chain.  The clock is read on every 1024th poll that found the phase incomplete (WaitCtx::check_mask).
__device__ __forceinline__ bool phase_wait(unsigned int* bar, int ctr, unsigned int epochs, unsigned int producers, const WaitCtx& wc) {
    if (threadIdx.x == 0) {
        const unsigned int target = epochs * producers;
        const unsigned int* w = ctr_word(bar, ctr, blockIdx.x & (kReplicas - 1));
        unsigned int v, spins = 0;
        while ((v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            if ((++spins & wc.check_mask) == 0 && __builtin_amdgcn_s_memrealtime() - wc.t0 > (unsigned long long)wc.deadline) {
                raise_abort(bar);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the bits have landed before this workgroup can be counted out
                v = kAbortBit;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (v & kAbortBit) *wc.lds_abort = 1;
#ifdef MVDB_WALK_ACQUIRE_LOADS
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: this CU's L1 forgets what other CUs have rewritten
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the workgroup barrier below holds until the invalidate has completed
#endif
    }
    __syncthreads();
    return *wc.lds_abort != 0;
}

// sum over the workgroup (all kThreads threads call it); red8: 8 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red8, int lane, int wave) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();  // red8 free again
    if (lane == 0) red8[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s += red8[w];
    return s;
}

// LayerNorm of one row held 4 columns per thread (thread t < H / 4 holds columns 4t .. 4t + 3; the others pass zeros and
// active = false), biased variance, eps inside the sqrt; the normalised row goes to `out` (sc1) at byte offset row_off.
__device__ __forceinline__ void row_layernorm(f32x4 v, bool active, int t, int H, float eps, f32x4 g4, f32x4 b4, rsrc_t out,
                                              int row_off, float* red8, int lane, int wave) {
    const float s = block_sum(active ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f, red8, lane, wave);
    const float mean = s / (float)H;
    f32x4 d = v - mean;
    const float q = block_sum(active ? (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]) : 0.f, red8, lane, wave);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);
    if (active) st4(out, row_off + 16 * t, d * rstd * g4 + b4);
}

// ---- column units ------------------------------------------------------------------------------------------------------------
// acc[mt] += W[n0 .. n0 + 15][this wave's k chunks] . X[rows of tile mt][same k]^T, K split over the waves in chunks of 16:
// chunk c = wave + 8 i, i < HC.  Everything is branch-free so that every load of a batch is issued before the first MFMA (with
// `if (chunk exists)` / `if (tile exists)` around the loads hipcc emitted load, wait, MFMA group by group: one memory round
// trip per group, 10 us per QKV phase at 64 tokens): a chunk beyond K re-reads the last one against a zeroed weight
// fragment, row tiles beyond T read row T - 1 and their sums are never stored.
template <int HC>
__device__ __forceinline__ void colunit_load_w(const float* __restrict__ Wrow0, int K, f32x4 (&a)[HC], int lane, int wave) {
    const int r = lane & 15, g = lane >> 4, nch = K >> 4;
#pragma unroll
    for (int i = 0; i < HC; ++i) {
        const int c = wave + i * kWaves;
        const f32x4 v = *reinterpret_cast<const f32x4*>(Wrow0 + (int64_t)r * K + 16 * min(c, nch - 1) + 4 * g);
        a[i] = c < nch ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
template <int MT, int HC>
__device__ __forceinline__ void colunit_load_x(rsrc_t Ar, int K, int T, f32x4 (&b)[HC][MT], int lane, int wave, int row0 = 0) {
    const int r = lane & 15, g = lane >> 4, nch = K >> 4;
#pragma unroll
    for (int i = 0; i < HC; ++i) {
        const int cc = min(wave + i * kWaves, nch - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[i][mt] = ld4(Ar, (min(row0 + mt * 16 + r, T - 1) * K + 16 * cc + 4 * g) * 4);
    }
}
template <int MT, int HC>
__device__ __forceinline__ void colunit_mfma(const f32x4 (&a)[HC], const f32x4 (&b)[HC][MT], f32x4 (&acc)[MT]) {
#pragma unroll
    for (int i = 0; i < HC; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][m], b[i][mt][m], acc[mt], 0, 0, 0);
}
// The same product with the operand fragments fetched CB chunks at a time (HC x MT of them at once would not fit the registers
// of the wide shapes): W rows of stride ldw floats starting at Wrow0 (which already points at the first k of the range), A
// rows of stride lda floats from k offset koff, nch chunks of 16 in the range.
template <int CB>
__device__ __forceinline__ void colunit_load_w_batch(const float* __restrict__ Wrow0, int ldw, int nch, int i0, f32x4 (&wa)[CB], int lane,
                                                     int wave) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const int c = wave + (i0 + i) * kWaves, cc = min(c, nch - 1);
        const f32x4 v = *reinterpret_cast<const f32x4*>(Wrow0 + (int64_t)r * ldw + 16 * cc + 4 * g);
        wa[i] = c < nch ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
// w0 = the first batch's weight fragments, loaded by the caller BEFORE it waited for the activations
template <int MT, int HC, int CB>
__device__ __forceinline__ void colunit_accumulate(const float* __restrict__ Wrow0, int ldw, const f32x4 (&w0)[CB], rsrc_t Ar, int lda,
                                                   int koff, int nch, int T, int row0, f32x4 (&acc)[MT], int lane, int wave) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i0 = 0; i0 < HC; i0 += CB) {
        f32x4 wa[CB], xb[CB][MT];
        if (i0 == 0) {
#pragma unroll
            for (int i = 0; i < CB; ++i) wa[i] = w0[i];
        } else {
            colunit_load_w_batch<CB>(Wrow0, ldw, nch, i0, wa, lane, wave);
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int cc = min(wave + (i0 + i) * kWaves, nch - 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xb[i][mt] = ld4(Ar, (min(row0 + mt * 16 + r, T - 1) * lda + koff + 16 * cc + 4 * g) * 4);
        }
        colunit_mfma<MT, CB>(wa, xb, acc);
    }
}
// wave partials of a column unit -> LDS; after the barrier the partials are added in wave order
template <int MT>
__device__ __forceinline__ void colunit_publish(const f32x4 (&acc)[MT], f32x4* red, int lane, int wave) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[(wave * MT + mt) * 64 + lane] = acc[mt];
}
template <int MT>
__device__ __forceinline__ f32x4 colunit_total(const f32x4* red, int mt, int lane) {
    f32x4 s = red[mt * 64 + lane];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += red[(w * MT + mt) * 64 + lane];
    return s;
}

// ---- attention of one sentence's head from LDS tiles --------------------------------------------------------------------------
// Qs [rows][hd + 4] (scaled by log2(e) / sqrt(hd)), Ks [rows][hd + 4], Vt [hd][16 MT + 4] (V transposed), Cs [rows][hd + 4] receives
// the context.  Wave qi < ceil(len / 16) owns query tile qi.  Key tiles beyond the sentence must hold FINITE values (they are
// masked to -inf by index, and their V columns meet p = 0).
template <int MT>
__device__ __forceinline__ void attention_tile(const float* Qs, const float* Ks, const float* Vt, float* Cs, int hd, int len, int qi,
                                               int lane) {
    constexpr int VS = MT * 16 + 4;  // row stride of V^T: every key of the MT tiles + one 16-byte pad
    const int r = lane & 15, g = lane >> 4, hd4 = hd + 4;
    f32x4 sc[MT];
#pragma unroll
    for (int kj = 0; kj < MT; ++kj) sc[kj] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < (hd >> 4); ++c) {
        const f32x4 qb = *reinterpret_cast<const f32x4*>(Qs + (16 * qi + r) * hd4 + 16 * c + 4 * g);
#pragma unroll
        for (int kj = 0; kj < MT; ++kj) {
            const f32x4 ka = *reinterpret_cast<const f32x4*>(Ks + (16 * kj + r) * hd4 + 16 * c + 4 * g);
#pragma unroll
            for (int m = 0; m < 4; ++m) sc[kj] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[m], qb[m], sc[kj], 0, 0, 0);
        }
    }
    // sc[kj][v] = score(query 16 qi + r, key 16 kj + 4 g + v): softmax over the keys of this query
    float mx = -INFINITY;
#pragma unroll
    for (int kj = 0; kj < MT; ++kj)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            sc[kj][v] = 16 * kj + 4 * g + v < len ? sc[kj][v] : -INFINITY;
            mx = fmaxf(mx, sc[kj][v]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kj = 0; kj < MT; ++kj)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            sc[kj][v] = __builtin_amdgcn_exp2f(sc[kj][v] - mx);  // masked keys: 2^(-inf) = 0
            sum += sc[kj][v];
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kj = 0; kj < MT; ++kj) sc[kj] *= inv;
    // ctx^T = V^T P^T: A = V^T (i = d), B = P^T straight from the score accumulators (k = key)
    for (int dt = 0; dt < (hd >> 4); ++dt) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kj = 0; kj < MT; ++kj) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(Vt + (16 * dt + r) * VS + 16 * kj + 4 * g);
#pragma unroll
            for (int m = 0; m < 4; ++m) o = __builtin_amdgcn_mfma_f32_16x16x4f32(va[m], sc[kj][m], o, 0, 0, 0);
        }
        *reinterpret_cast<f32x4*>(Cs + (16 * qi + r) * hd4 + 16 * dt + 4 * g) = o;  // ctx[query 16 qi + r][16 dt + 4 g ..]
    }
}

// plane[h][s0 + query][16 nt ..] = sum_d ctx[query][d] Wo[16 nt ..][h hd + d] for one column tile; wa = the tile's Wo fragments
template <int MT>  // MT query tiles from query row q0 on
__device__ __forceinline__ void outproj_tile(const f32x4 (&wa)[4], const float* Cs, int hd, int len, rsrc_t PLr, int plane_row0, int H,
                                             int nt, int lane, int q0 = 0) {
    const int r = lane & 15, g = lane >> 4, hd4 = hd + 4, nc = hd >> 4;
#pragma unroll
    for (int qi = 0; qi < MT; ++qi) {  // query tiles beyond the sentence: never stored
        const int row = q0 + 16 * qi + r;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 cb = *reinterpret_cast<const f32x4*>(Cs + row * hd4 + 16 * min(c, nc - 1) + 4 * g);
#pragma unroll
            for (int m = 0; m < 4; ++m) o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][m], cb[m], o, 0, 0, 0);
        }
        if (row < len) st4(PLr, ((plane_row0 + row) * H + 16 * nt + 4 * g) * 4, o);
    }
}
__device__ __forceinline__ void outproj_load_w(const float* __restrict__ wo, int H, int h, int hd, int nt, f32x4 (&wa)[4], int lane) {
    const int r = lane & 15, g = lane >> 4, nc = hd >> 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // branch-free: a chunk beyond the head re-reads the last one, zeroed
        const f32x4 v = *reinterpret_cast<const f32x4*>(wo + (int64_t)(16 * nt + r) * H + h * hd + 16 * min(c, nc - 1) + 4 * g);
        wa[c] = c < nc ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ---- x[row] = LayerNorm(sum of `np` planes + bias + residual[row]) for the rows this workgroup owns (row = wg, wg + G, ...) ---
// Thread t = pg * (H / 4) + q sums planes pg, pg + npg, ... of column quad q (up to 24 loads in flight), the plane groups are
// added in group order through LDS, then the LayerNorm over the workgroup.  bias / gamma / beta were loaded before the wait.
struct LnWeights { f32x4 bias, gamma, beta; };
__device__ __forceinline__ LnWeights ln_load_w(const float* __restrict__ bias, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, int H, int tid) {
    LnWeights w;
    const int t = min(tid, (H >> 2) - 1);
    w.bias = *reinterpret_cast<const f32x4*>(bias + 4 * t);
    w.gamma = *reinterpret_cast<const f32x4*>(gamma + 4 * t);
    w.beta = *reinterpret_cast<const f32x4*>(beta + 4 * t);
    return w;
}
template <int RB>  // plane loads in flight per thread
__device__ __forceinline__ void reduce_ln_rows(const Args& a, rsrc_t PLr, int np, const LnWeights& lw, rsrc_t Rr, rsrc_t Or, int T,
                                               f32x4* comb, float* red8, int wg, int G, int tid, int lane, int wave,
                                               unsigned long long* trace, int layer) {
    const int H = a.H, HQ = H >> 2;
    const int npg = min(kWaves, kThreads / HQ);
    const int pg = tid / HQ, q = tid - pg * HQ;
    for (int row = wg; row < T; row += G) {
        const bool active = tid < HQ;
        f32x4 res = {0.f, 0.f, 0.f, 0.f};
        if (active) res = ld4(Rr, (row * H + 4 * tid) * 4);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (pg < npg) {
            for (int p0 = pg; p0 < np; p0 += RB * npg) {
                f32x4 v[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u)  // branch-free: a plane beyond np reads the last one and adds nothing
                    v[u] = ld4(PLr, ((min(p0 + u * npg, np - 1) * kTmax + row) * H + 4 * q) * 4);
#pragma unroll
                for (int u = 0; u < RB; ++u) s += p0 + u * npg < np ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            comb[pg * HQ + q] = s;
        }
        __syncthreads();
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            v = comb[tid];
            for (int j = 1; j < npg; ++j) v += comb[j * HQ + tid];
            v = (v + lw.bias) + res;
        }
        row_layernorm(v, active, tid, H, a.eps, lw.gamma, lw.beta, Or, row * H * 4, red8, lane, wave);
        __syncthreads();  // comb free for the next row
    }
}
// The batch of loads a thread keeps in flight is sized to its share of the planes (12 head planes over 5 plane groups = 3
// each; 96 FFN planes = 20 each): a fixed batch of 24 issued 21 duplicate loads of the last plane per thread for the head planes.
__device__ __forceinline__ void phase_reduce_ln(const Args& a, rsrc_t PLr, int np, const LnWeights& lw, rsrc_t Rr, rsrc_t Or, int T,
                                                f32x4* comb, float* red8, int wg, int G, int tid, int lane, int wave,
                                                unsigned long long* trace = nullptr, int layer = 0) {
    const int npg = min(kWaves, kThreads / (a.H >> 2));
    const int share = (np + npg - 1) / npg;
    if (share <= 4) reduce_ln_rows<4>(a, PLr, np, lw, Rr, Or, T, comb, red8, wg, G, tid, lane, wave, trace, layer);
    else if (share <= 8) reduce_ln_rows<8>(a, PLr, np, lw, Rr, Or, T, comb, red8, wg, G, tid, lane, wave, trace, layer);
    else if (share <= 12) reduce_ln_rows<12>(a, PLr, np, lw, Rr, Or, T, comb, red8, wg, G, tid, lane, wave, trace, layer);
    else reduce_ln_rows<20>(a, PLr, np, lw, Rr, Or, T, comb, red8, wg, G, tid, lane, wave, trace, layer);
}

// MT = row tiles (of 16 tokens) a column unit computes, RH = row groups: at more than 32 tokens the column units are split
// by rows too — unit (columns u, group hf) computes rows [32 hf, 32 hf + 32), RH = 2 up to 64 tokens, 4 up to 128 — so that a
// long sentence spreads over more workgroups and each loads and stores a share of the activations; AT = MT * RH row tiles
// for the attention, which needs every key of a sentence.  (33 - 64 slots as FOUR row tiles per unit and no row groups — half the
// workgroups, every unit's weights fetched once, roles placed: 0.416 ms against 0.375 with MT = 2, RH = 2.)  (Tried and removed: the QKV columns of a head computed by the head's own workgroup
// straight into the attention's LDS tiles, one phase fewer per layer — 12 CUs then do the whole QKV product on the fp32
// matrix pipe: 0.33 / 0.43 ms per forward at 8 / 32 tokens against 0.27 / 0.34 with QKV as its own 72-workgroup phase.)
template <int MT, int HC, int RH>
__global__ __launch_bounds__(kThreads) void encoder_walk_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // ---- LDS map -------------------------------------------------------------------------------------------------------
    int* s_tok_id = reinterpret_cast<int*>(lds);   // [128] packed token -> vocabulary id
    int* s_tok_pos = s_tok_id + kTmax;              // [128] packed token -> position id
    int* s_slot_p = s_tok_pos + kTmax;              // [128] token slot (b * S + t) -> packed token, or -1
    int* s_seq = s_slot_p + kTmax;                  // [B + 1 <= 129] first packed token of each sentence
    float* red8 = lds + kLdsHead - 8;               // [8]
    int* s_abort = reinterpret_cast<int*>(lds + kLdsHead - 9);  // set when a wait of this workgroup was released by kAbortBit
    if (threadIdx.x == 0) *s_abort = 0;
    const WaitCtx wc{__builtin_amdgcn_s_memrealtime(), a.deadline, a.deadline < 100000u ? 0u : 1023u, s_abort};
    float* work = lds + kLdsHead;                   // phase scratch (16-byte aligned)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x, G = gridDim.x;
    const int H = a.H, F = a.F, hd = a.hd;
    const int r = lane & 15, g = lane >> 4;

    // ---- packing: every workgroup derives it from the mask (B * S <= 128 slots = two ballots) ------------------------------
    if (wave == 0) {
        const int slots = a.B * a.S;
        const bool v0 = lane < slots && a.mask[lane] != 0;
        const bool v1 = lane + 64 < slots && a.mask[lane + 64] != 0;
        const unsigned long long m0 = __ballot(v0), m1 = __ballot(v1);
        const int n0 = __popcll(m0);
        // valid slots in [0, e)
        auto below = [&](int e) {
            const int lo = min(e, 64), hi = max(e - 64, 0);
            return __popcll(m0 & (lo >= 64 ? ~0ull : (1ull << lo) - 1ull)) + __popcll(m1 & (hi >= 64 ? ~0ull : (1ull << hi) - 1ull));
        };
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int slot = lane + 64 * half;
            const bool v = half ? v1 : v0;
            const int p = half ? n0 + __popcll(m1 & ((1ull << lane) - 1ull)) : __popcll(m0 & ((1ull << lane) - 1ull));
            const int b = slot < slots ? slot / a.S : 0, t = slot - b * a.S;
            const int sb = below(b * a.S);
            if (slot < slots) s_slot_p[slot] = v ? p : -1;
            if (v) {
                const int id = a.ids[slot];
                s_tok_id[p] = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
                s_tok_pos[p] = a.position_offset > 0 ? (p - sb) + a.position_offset : t;
            }
            if (slot <= a.B) s_seq[slot] = below(slot * a.S);
        }
        if (lane == 0) s_seq[a.B] = n0 + __popcll(m1);  // (B = 128 sentences of one token: slot 128 is nobody's)
    }
    if (wg == 0 && tid == 0 && a.flag) *a.flag = 0u;
    __syncthreads();
    const int T = s_seq[a.B];
    unsigned long long* trace = a.trace ? a.trace + (size_t)wg * kTraceSlots : nullptr;
    if (trace && tid == 0) {
        trace[kTraceSlots - 2] = __builtin_amdgcn_s_memrealtime();
        trace[kTraceSlots - 4] = __builtin_readcyclecounter();  // shader clock: (end - start) / realtime = the clock the launch ran at
    }
    const rsrc_t Xr = make_rsrc(a.X), X1r = make_rsrc(a.X1), Qr = make_rsrc(a.QKV), PLr = make_rsrc(a.PL);
    const int HQ = H >> 2;
    // ROLES.  A workgroup's index inside a phase is its distance from the phase's base (mod G): the host places the bases so
    // that, where the grid allows, no workgroup produces in one phase AND consumes in the next.  Such a workgroup reaches its
    // wait late (it has to drain its own stores first, then issues the next phase's weight loads, then polls): with every
    // phase based at workgroup 0 the row-owning workgroups 0 .. T-1 were released 1.4 - 2.2 us after a phase's last arrival
    // where everybody else was at 0.8 - 0.9 (profiles/r05_walk_roles_before.json) — and a phase ends with its last workgroup.
    const int wr = (wg - a.off_rows + G) % G, wq = (wg - a.off_qkv + G) % G, wa_ = (wg - a.off_attn + G) % G, wf = (wg - a.off_ffn + G) % G;
    // producers per phase (what the consumers of a phase wait for, per layer)
    const unsigned int prodRow = (unsigned int)min(T, G);                      // embeddings, sum + LN
    const int ntiles = H >> 4;
    const int ntu = (ntiles + a.nsplit - 1) / a.nsplit;
    const int attn_units = a.B * a.heads * a.nsplit * RH;  // (sentence, head, column split of the out-projection, query half)
    constexpr int AT = MT * RH;                                               // row tiles of the attention
    const int ncol = (3 * H) >> 4;                                            // column units of the QKV product
    const unsigned int prodQkv = (unsigned int)min(ncol * RH, G);
    const unsigned int prodAttn = (unsigned int)min(attn_units, G);
    // narrow shapes (H <= 384): the FFN is ONE phase — a workgroup owns 16-wide slices of F, GELU(x1 W1_slice^T) stays in LDS
    // and is multiplied by W2[:, slice] into the workgroup's partial plane.  Wide shapes (e5-large / bge-m3: H 1024, F 4096):
    // two phases — FFN1 as F / 16 column units into Hb, FFN2 as column units whose K = F is split kKP ways over workgroups
    // (kKP partial planes) — the one-phase form there was 27 us per layer of serialised slices and 128 planes to sum.
    constexpr bool kFfnSplit = HC > 4;
    const int nf = F >> 4;
    const int kKP = max(4, (nf + 63) >> 6);            // K parts of FFN2: at most 64 chunks (8 per wave) each
    const int nchp = (nf + kKP - 1) / kKP;             // chunks per part
    const unsigned int prodFfn1 = (unsigned int)min(nf * RH, G);
    const unsigned int prodFfn = kFfnSplit ? (unsigned int)min(ntiles * kKP * RH, G) : (unsigned int)min(a.np3 * RH, G);
    const int ffn_planes = kFfnSplit ? kKP : a.np3;

    if (T > 0) {
        // ---- embeddings + LayerNorm -> X (one workgroup per row) ---------------------------------------------------------------
        if (wr < (int)prodRow) {
            const bool active = tid < HQ;
            const int tq = min(tid, HQ - 1);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.embg + 4 * tq), b4 = *reinterpret_cast<const f32x4*>(a.embb + 4 * tq);
            for (int p = wr; p < T; p += G) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.word + (int64_t)s_tok_id[p] * H + 4 * tq);
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.type + 4 * tq);
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(a.pos + (int64_t)s_tok_pos[p] * H + 4 * tq);
                const f32x4 v = (w4 + t4) + p4;  // HF: inputs_embeds + token_type, then + position
                row_layernorm(v, active, tid, H, a.eps, g4, b4, Xr, p * H * 4, red8, lane, wave);
            }
            phase_arrive(a.bar, kCtrEmbed);
        }

        const float qscale = 1.4426950408889634f / sqrtf((float)hd);  // log2(e) / sqrt(hd): softmax by exp2
        const int hd4 = hd + 4;
        for (int layer = 0; layer < a.nlayers; ++layer) {
            const LayerPtrs L = a.layers[layer];
            const unsigned int lay1 = (unsigned int)layer + 1u;
            // what the first phase of a layer waits for: the embeddings, or the previous layer's last sum + LN
            const int in_ctr = layer == 0 ? kCtrEmbed : kCtrLn2;
            const unsigned int in_epochs = layer == 0 ? 1u : (unsigned int)layer;

            {
                // ---- QKV: column units over 3H -> QKV[T, 3H] ----------------------------------------------------------------------
                if (wq < (int)prodQkv) {
                    f32x4* red = reinterpret_cast<f32x4*>(work);
                    constexpr int CB = HC > 4 ? (MT > 1 ? 4 : 8) : HC;  // chunk batches: HC x MT operand fragments must fit the registers
                    bool first = true;
                    for (int uu = wq; uu < ncol * RH; uu += G) {
                        const int hf = uu / ncol, u = uu - hf * ncol, row0 = hf * MT * 16;  // columns u, row half hf
                        f32x4 acc[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        const f32x4 bias = *reinterpret_cast<const f32x4*>(L.bqkv + u * 16 + 4 * g);
                        if constexpr (CB == HC) {
                            f32x4 wa[HC], xb[HC][MT];
                            colunit_load_w<HC>(L.wqkv + (int64_t)u * 16 * H, H, wa, lane, wave);
                            if (first) {
                                stamp(trace, layer, 0, 0);
                                if (phase_wait(a.bar, in_ctr, in_epochs, prodRow, wc)) goto walk_out;
                                stamp(trace, layer, 0, 1);
                                first = false;
                            }
                            colunit_load_x<MT, HC>(Xr, H, T, xb, lane, wave, row0);
                            colunit_mfma<MT, HC>(wa, xb, acc);
                        } else {
                            f32x4 w0[CB];
                            colunit_load_w_batch<CB>(L.wqkv + (int64_t)u * 16 * H, H, H >> 4, 0, w0, lane, wave);
                            if (first) {
                                stamp(trace, layer, 0, 0);
                                if (phase_wait(a.bar, in_ctr, in_epochs, prodRow, wc)) goto walk_out;
                                stamp(trace, layer, 0, 1);
                                first = false;
                            }
                            colunit_accumulate<MT, HC, CB>(L.wqkv + (int64_t)u * 16 * H, H, w0, Xr, H, 0, H >> 4, T, row0, acc, lane, wave);
                        }
                        colunit_publish<MT>(acc, red, lane, wave);
                        __syncthreads();
                        if (wave < MT) {
                            const int tok = row0 + wave * 16 + r, n = u * 16 + 4 * g;
                            const f32x4 v = colunit_total<MT>(red, wave, lane) + bias;
                            if (tok < T) st4(Qr, (tok * 3 * H + n) * 4, v);
                        }
                        __syncthreads();
                    }
                    stamp(trace, layer, 0, 2);
                    phase_arrive(a.bar, kCtrQkv);
                }

                // ---- attention per (sentence, head, column split) + out-projection partial -> plane[head] --------------------------------
                if (wa_ < (int)prodAttn) {
                    constexpr int VS = AT * 16 + 4;          // row stride of V^T
                    float* Qs = work;                       // [AT * 16][hd + 4]  queries, scaled
                    float* Ks = Qs + AT * 16 * hd4;          // [AT * 16][hd + 4]
                    float* Vt = Ks + AT * 16 * hd4;          // [hd][AT * 16 + 4]  V transposed
                    float* Cs = Vt + hd * VS;               // [AT * 16][hd + 4]  context of this head
                    bool first = true;
                    for (int u = wa_; u < attn_units; u += G) {
                        const int qh = u % RH, u1 = u / RH;                 // query tiles [qh MT, qh MT + MT) of the sentence
                        const int ns = u1 % a.nsplit, bh = u1 / a.nsplit;
                        const int h = bh % a.heads, b = bh / a.heads;
                        const int s0 = s_seq[b], len = s_seq[b + 1] - s0;
                        const int nt0 = ns * ntu, nt1 = min(ntiles, nt0 + ntu);
                        f32x4 wo0[4];  // the first column tile's fragments before the wait
                        outproj_load_w(L.wo, H, h, hd, min(nt0 + wave, ntiles - 1), wo0, lane);
                        if (first) {
                            stamp(trace, layer, 1, 0);
                            if (phase_wait(a.bar, kCtrQkv, lay1, prodQkv, wc)) goto walk_out;
                            stamp(trace, layer, 1, 1);
                            first = false;
                        }
                        if (len <= 0) continue;  // uniform over the workgroup
                        const int mtb = (len + 15) >> 4;
                        // Q, K, V of (sentence b, head h): rows < len from QKV, the rest of the row tiles zero
                        const int q4 = hd >> 2;  // float4 per row
                        for (int e = tid; e < AT * 16 * q4; e += kThreads) {
                            const int row = e / q4, c4 = e - row * q4;
                            f32x4 qv = {0.f, 0.f, 0.f, 0.f}, kv = qv, vv = qv;
                            if (row < len) {
                                const int base = ((s0 + row) * 3 * H + h * hd + 4 * c4) * 4;
                                qv = ld4(Qr, base) * qscale;
                                kv = ld4(Qr, base + H * 4);
                                vv = ld4(Qr, base + 2 * H * 4);
                            }
                            *reinterpret_cast<f32x4*>(Qs + row * hd4 + 4 * c4) = qv;
                            *reinterpret_cast<f32x4*>(Ks + row * hd4 + 4 * c4) = kv;
#pragma unroll
                            for (int j = 0; j < 4; ++j) Vt[(4 * c4 + j) * VS + row] = vv[j];
                        }
                        __syncthreads();
                        if (wave < MT && qh * MT + wave < mtb) attention_tile<AT>(Qs, Ks, Vt, Cs, hd, len, qh * MT + wave, lane);
                        __syncthreads();
                        for (int nt = nt0 + wave; nt < nt1; nt += kWaves) {
                            f32x4 wa[4];
                            if (nt == nt0 + wave) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) wa[c] = wo0[c];
                            } else {
                                outproj_load_w(L.wo, H, h, hd, nt, wa, lane);
                            }
                            outproj_tile<MT>(wa, Cs, hd, len, PLr, h * kTmax + s0, H, nt, lane, qh * MT * 16);
                        }
                        __syncthreads();  // Q / K / V / ctx tiles free for the next unit
                    }
                    stamp(trace, layer, 1, 2);
                    phase_arrive(a.bar, kCtrAttn);
                }
            }

            // ---- x1 = LN(sum of head planes + bo + x) -> X1 ---------------------------------------------------------------------
            if (wr < (int)prodRow) {
                const LnWeights lw = ln_load_w(L.bo, L.ln1g, L.ln1b, H, tid);
                stamp(trace, layer, 2, 0);
                if (phase_wait(a.bar, kCtrAttn, lay1, prodAttn, wc)) goto walk_out;
                stamp(trace, layer, 2, 1);
                phase_reduce_ln(a, PLr, a.heads, lw, Xr, X1r, T, reinterpret_cast<f32x4*>(work), red8, wr, G, tid, lane, wave, trace, layer);
                stamp(trace, layer, 2, 2);
                phase_arrive(a.bar, kCtrLn1);
            }

            if constexpr (kFfnSplit) {
                const rsrc_t Hr = make_rsrc(a.Hb);
                constexpr int CB = MT > 1 ? 4 : 8;
                f32x4* red = reinterpret_cast<f32x4*>(work);
                // ---- FFN1: column units over F -> Hb[T, F] = GELU(x1 W1^T + b1) --------------------------------------------------
                if (wf < (int)prodFfn1) {
                    bool first = true;
                    for (int uu = wf; uu < nf * RH; uu += G) {
                        const int hf = uu / nf, j = uu - hf * nf, row0 = hf * MT * 16;
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(L.b1 + 16 * j + 4 * g);
                        f32x4 w0[CB];
                        colunit_load_w_batch<CB>(L.w1 + (int64_t)j * 16 * H, H, H >> 4, 0, w0, lane, wave);
                        if (first) {
                            stamp(trace, layer, 5, 0);
                            if (phase_wait(a.bar, kCtrLn1, lay1, prodRow, wc)) goto walk_out;
                            stamp(trace, layer, 5, 1);
                            first = false;
                        }
                        f32x4 acc[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        colunit_accumulate<MT, HC, CB>(L.w1 + (int64_t)j * 16 * H, H, w0, X1r, H, 0, H >> 4, T, row0, acc, lane, wave);
                        colunit_publish<MT>(acc, red, lane, wave);
                        __syncthreads();
                        if (wave < MT) {
                            const int tok = row0 + wave * 16 + r;
                            f32x4 v = colunit_total<MT>(red, wave, lane) + b1;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
                            if (tok < T) st4(Hr, (tok * F + 16 * j + 4 * g) * 4, v);
                        }
                        __syncthreads();
                    }
                    stamp(trace, layer, 5, 2);
                    phase_arrive(a.bar, kCtrFfn1);
                }
                // ---- FFN2: column units over H, K = F split kKP ways -> plane[kp][T, H] --------------------------------------------
                if (wf < (int)prodFfn) {
                    bool first = true;
                    for (int uu = wf; uu < ntiles * kKP * RH; uu += G) {
                        const int hf = uu / (ntiles * kKP), rem = uu - hf * ntiles * kKP;
                        const int kp = rem / ntiles, nt = rem - kp * ntiles, row0 = hf * MT * 16;
                        const int k0 = kp * nchp * 16, nch = max(1, min(nchp, nf - kp * nchp));
                        const bool live = nf - kp * nchp > 0;  // (a part beyond F when F / 16 is not a multiple of the parts: adds nothing)
                        f32x4 w0[CB];
                        colunit_load_w_batch<CB>(L.w2 + (int64_t)nt * 16 * F + (live ? k0 : 0), F, nch, 0, w0, lane, wave);
                        if (first) {
                            stamp(trace, layer, 3, 0);
                            if (phase_wait(a.bar, kCtrFfn1, lay1, prodFfn1, wc)) goto walk_out;
                            stamp(trace, layer, 3, 1);
                            first = false;
                        }
                        f32x4 acc[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (live) colunit_accumulate<MT, HC, CB>(L.w2 + (int64_t)nt * 16 * F + k0, F, w0, Hr, F, k0, nch, T, row0, acc, lane, wave);
                        colunit_publish<MT>(acc, red, lane, wave);
                        __syncthreads();
                        if (wave < MT) {
                            const int tok = row0 + wave * 16 + r;
                            const f32x4 v = colunit_total<MT>(red, wave, lane);
                            if (tok < T) st4(PLr, ((kp * kTmax + tok) * H + 16 * nt + 4 * g) * 4, v);
                        }
                        __syncthreads();
                    }
                    stamp(trace, layer, 3, 2);
                    phase_arrive(a.bar, kCtrFfn);
                }
            } else
            // ---- FFN: workgroup wg < np3 owns the 16-wide slices wg, wg + np3, ... of F -> plane[wg] ---------------------------------
            if (wf < (int)prodFfn) {
                f32x4* red = reinterpret_cast<f32x4*>(work);
                float* hbuf = work + kWaves * MT * 64 * 4;  // [MT * 16][20]: GELU(x1 W1_slice^T + b1)
                bool first = true;
              for (int uu = wf; uu < a.np3 * RH; uu += G) {   // units (plane, row group): more than one per workgroup above 64 tokens
                const int hf = uu / a.np3, pl = uu - hf * a.np3, row0 = hf * MT * 16;  // plane (= first slice) pl, row group hf
                f32x4 acc2[HC][MT];
#pragma unroll
                for (int i = 0; i < HC; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc2[i][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int j = pl; j < (F >> 4); j += a.np3) {
                    f32x4 acc[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(L.b1 + 16 * j + 4 * g);
                    f32x4 w2f[HC];
                    {
                        f32x4 wa[HC], xb[HC][MT];
                        colunit_load_w<HC>(L.w1 + (int64_t)j * 16 * H, H, wa, lane, wave);
#pragma unroll
                        for (int i = 0; i < HC; ++i) {  // a tile beyond H recomputes the last one and is never stored
                            const int nt = min(wave + i * kWaves, ntiles - 1);
                            w2f[i] = *reinterpret_cast<const f32x4*>(L.w2 + (int64_t)(16 * nt + r) * F + 16 * j + 4 * g);
                        }
                        if (first) {
                            stamp(trace, layer, 3, 0);
                            if (phase_wait(a.bar, kCtrLn1, lay1, prodRow, wc)) goto walk_out;
                            stamp(trace, layer, 3, 1);
                            first = false;
                        }
                        colunit_load_x<MT, HC>(X1r, H, T, xb, lane, wave, row0);
                        colunit_mfma<MT, HC>(wa, xb, acc);
                    }
                    colunit_publish<MT>(acc, red, lane, wave);
                    __syncthreads();
                    if (wave < MT) {
                        f32x4 v = colunit_total<MT>(red, wave, lane) + b1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
                        *reinterpret_cast<f32x4*>(hbuf + (wave * 16 + r) * 20 + 4 * g) = v;
                    }
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const f32x4 wa = w2f[i];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {  // row tiles beyond T: never stored
                            const f32x4 hb = *reinterpret_cast<const f32x4*>(hbuf + (mt * 16 + r) * 20 + 4 * g);
#pragma unroll
                            for (int m = 0; m < 4; ++m) acc2[i][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m], hb[m], acc2[i][mt], 0, 0, 0);
                        }
                    }
                    __syncthreads();  // red / hbuf free for the next slice
                }
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int nt = wave + i * kWaves;
                    if (nt < ntiles) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            if (row0 + mt * 16 + r < T) st4(PLr, ((pl * kTmax + row0 + mt * 16 + r) * H + 16 * nt + 4 * g) * 4, acc2[i][mt]);
                    }
                }
              }
                stamp(trace, layer, 3, 2);
                phase_arrive(a.bar, kCtrFfn);
            }

            // ---- x = LN(sum of FFN planes + b2 + x1) -> X -----------------------------------------------------------------------
            if (wr < (int)prodRow) {
                const LnWeights lw = ln_load_w(L.b2, L.ln2g, L.ln2b, H, tid);
                stamp(trace, layer, 4, 0);
                if (phase_wait(a.bar, kCtrFfn, lay1, prodFfn, wc)) goto walk_out;
                stamp(trace, layer, 4, 1);
                phase_reduce_ln(a, PLr, ffn_planes, lw, X1r, Xr, T, reinterpret_cast<f32x4*>(work), red8, wr, G, tid, lane, wave);
                stamp(trace, layer, 4, 2);
                phase_arrive(a.bar, kCtrLn2);
            }
        }
    }

    // ---- pooling + L2 normalise (average_pool + F.normalize(eps = 1e-12); pooling 1: first valid token) ------------------------
    if (wr < a.B || (a.hidden && wr < a.B * a.S)) {
        if (T > 0 && phase_wait(a.bar, kCtrLn2, (unsigned int)a.nlayers, prodRow, wc)) goto walk_out;
        for (int b = wr; b < a.B; b += G) {
            const int s0 = s_seq[b], len = s_seq[b + 1] - s0;
            const int span = a.pooling == 1 ? (len > 0 ? 1 : 0) : len;
            const bool active = tid < HQ;
            f32x4 e = {0.f, 0.f, 0.f, 0.f};
            if (active) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                int t = 0;
                for (; t + 8 <= span; t += 8) {  // eight loads in flight, added in token order
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = ld4(Xr, ((s0 + t + u) * H + 4 * tid) * 4);
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
                for (; t < span; ++t) s += ld4(Xr, ((s0 + t) * H + 4 * tid) * 4);
                e = s / (float)span;  // empty sentence -> NaN, as the reference's 0 / 0
            }
            const float sq = block_sum(active ? (e[0] * e[0] + e[1] * e[1]) + (e[2] * e[2] + e[3] * e[3]) : 0.f, red8, lane, wave);
            const float denom = fmaxf(sqrtf(sq), 1e-12f);
            if (active) *reinterpret_cast<f32x4*>(a.out + (int64_t)b * H + 4 * tid) = e / denom;
        }
        if (a.hidden) {  // hidden[b, t, :] = x[packed(b, t), :] for valid tokens, 0 for padding
            for (int slot = wr; slot < a.B * a.S; slot += G) {
                const int p = s_slot_p[slot];
                if (tid < HQ) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (p >= 0) v = ld4(Xr, (p * H + 4 * tid) * 4);
                    *reinterpret_cast<f32x4*>(a.hidden + (int64_t)slot * H + 4 * tid) = v;
                }
            }
        }
    }
    if (trace && tid == 0) {
        trace[kTraceSlots - 1] = __builtin_amdgcn_s_memrealtime();
        trace[kTraceSlots - 3] = __builtin_readcyclecounter();
    }
walk_out:
    // ---- the last workgroup out re-arms the counters for the next launch; an abandoned launch is counted and its output ----
    // ---- poisoned (NaN) so that nobody mistakes it for an embedding ----------------------------------------------------------
    __syncthreads();
    if (tid == 0) {
        const unsigned int left = __hip_atomic_fetch_add(ctr_word(a.bar, kCtrExit, 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int last = 0;
        if (left == (unsigned int)G - 1u) {
            const bool aborted = (__hip_atomic_load(ctr_word(a.bar, kCtrEmbed, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kAbortBit) != 0;
            last = aborted ? 2 : 1;
            for (int i = 0; i < kCtrCount * kReplicas; ++i)
                __hip_atomic_store(a.bar + i * kCtrStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (aborted) {
                const unsigned int n = __hip_atomic_fetch_add(a.aborts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                __hip_atomic_store(a.aborts_host, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (a.flag) *a.flag = 1u;  // what a caller of the device entry tests behind its own stream wait
            }
        }
        *s_abort = last;
    }
    __syncthreads();
    if (*s_abort == 2) {  // every other workgroup has left: nobody writes `out` any more
        for (int i = tid; i < a.B * H; i += kThreads) a.out[i] = __builtin_nanf("");
    }
}

// LDS bytes of a launch
inline size_t lds_bytes(int mt, int rh, int H, int hd) {
    const int at = mt * rh;                                                          // row tiles of the attention
    const size_t colunit = (size_t)kWaves * mt * 64 * 16 + (size_t)mt * 16 * 20 * 4;  // wave partials + GELU tile
    const size_t attn = (size_t)(3 * at * 16 * (hd + 4) + hd * (at * 16 + 4)) * 4;   // Q, K, ctx, V^T
    const size_t reduce = (size_t)kThreads * 16;                                     // plane-group partials
    // never less than half a CU's LDS + 1 KiB: two workgroups of a launch cannot share a CU (the measured sc1 hand-off is "one per CU")
    return std::max<size_t>((size_t)kLdsHead * 4 + std::max(std::max(colunit, attn), reduce), 81 * 1024);
}

}  // namespace walk
