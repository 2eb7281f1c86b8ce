// encoder_walk.hpp — the whole encoder forward of a SMALL batch as ONE launch that walks the layers itself.
//
// Reference shape: extract_embeddings(text) tokenises ONE sentence per call (minivectordb/embedding_model.py:62-71), i.e.
// 4 .. 60 tokens.  As a chain of per-op kernels that forward is ~76 launches of 4.5-9 us each (0.60 ms, round 4): the
// arithmetic (2.7 GFLOP at 64 tokens) and the weight bytes (85 MB, Infinity-Cache resident) are a few tens of us.  Here
// one persistent launch of G <= #CU workgroups (one per CU, all resident) runs
//     embeddings + LN | per layer: QKV | attention + out-proj partials | sum + LN | FFN partials | sum + LN | pooling
// with a grid-wide barrier between the phases and every hand-off through L2 (write-through `sc1` stores, L1-bypassing
// `sc1` loads: MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility", first table row).
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
constexpr int kTmax = 64;     // packed tokens (and token slots B * S) per launch
constexpr int kSc1 = 16;      // cache-policy bit of the buffer builtins on gfx950: sc1
constexpr int kMaxPlanes = 128;

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
    unsigned int* bar;        // [2]: arrivals of the running launch, exits
    float* out;               // [B, H]
    float* hidden;            // NULL or [B, S, H]
    int np3;                  // workgroups (= planes) of the FFN phase
    int nsplit;               // column splits of the out-projection per (sentence, head)
};

__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4(rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, kSc1));
}
__device__ __forceinline__ void st4(rsrc_t r, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, kSc1);
}

// Grid-wide barrier: one monotonic arrival counter.  Every wave drains its own (write-through) stores, the workgroup meets,
// one lane adds and polls with L1-bypassing loads, the workgroup meets again; every load of handed-off bytes after it is sc1.
__device__ __forceinline__ void grid_sync(unsigned int* bar, unsigned int& epoch, unsigned int nwg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    epoch += nwg;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
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
__device__ __forceinline__ void row_layernorm(f32x4 v, bool active, int t, int H, float eps, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, rsrc_t out, int row_off, float* red8, int lane,
                                              int wave) {
    f32x4 g4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        g4 = *reinterpret_cast<const f32x4*>(gamma + 4 * t);
        b4 = *reinterpret_cast<const f32x4*>(beta + 4 * t);
    }
    const float s = block_sum(active ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f, red8, lane, wave);
    const float mean = s / (float)H;
    f32x4 d = v - mean;
    const float q = block_sum(active ? (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]) : 0.f, red8, lane, wave);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);
    if (active) st4(out, row_off + 16 * t, d * rstd * g4 + b4);
}

// One column unit: acc[mt] += W[n0 .. n0 + 15][this wave's k chunks] . X[rows of tile mt][same k]^T.
// Wrow0 = &W[n0][0] (row-major [N, K]); Ar = the activations [T, K] (sc1 loads); rows beyond T - 1 read row T - 1.
template <int MT, int HC>
__device__ __forceinline__ void colunit_gemm(const float* __restrict__ Wrow0, int K, rsrc_t Ar, int T, int mtc, f32x4 (&acc)[MT],
                                             int lane, int wave) {
    const int r = lane & 15, g = lane >> 4;
    const int nch = K >> 4;
    constexpr int CB = HC > 4 ? (MT > 2 ? 2 : 4) : HC;  // chunks in flight per wave: 8 chunks x 4 row tiles of operands would not fit beside the FFN accumulators
#pragma unroll
    for (int i0 = 0; i0 < HC; i0 += CB) {
        f32x4 a[CB], b[CB][MT];
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int c = wave + (i0 + i) * kWaves;
            if (c < nch) {
                a[i] = *reinterpret_cast<const f32x4*>(Wrow0 + (int64_t)r * K + 16 * c + 4 * g);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (mt < mtc) b[i][mt] = ld4(Ar, (min(mt * 16 + r, T - 1) * K + 16 * c + 4 * g) * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int c = wave + (i0 + i) * kWaves;
            if (c < nch) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        if (mt < mtc) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][m], b[i][mt][m], acc[mt], 0, 0, 0);
            }
        }
    }
}

// wave partials of a column unit -> LDS; after the barrier, wave mt (< mtc) owns tile mt and adds the 8 partials in wave order
template <int MT>
__device__ __forceinline__ void colunit_publish(const f32x4 (&acc)[MT], int mtc, f32x4* red, int lane, int wave) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        if (mt < mtc) red[(wave * MT + mt) * 64 + lane] = acc[mt];
}
template <int MT>
__device__ __forceinline__ f32x4 colunit_total(const f32x4* red, int mt, int lane) {
    f32x4 s = red[mt * 64 + lane];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += red[(w * MT + mt) * 64 + lane];
    return s;
}

// x[row] = LayerNorm(sum of `np` planes + bias + residual[row]) for the rows this workgroup owns (row = wg, wg + G, ...).
// Thread t = pg * (H / 4) + q sums planes pg, pg + npg, ... of column quad q (eight loads in flight), the plane groups are
// added in group order through LDS, then the LayerNorm over the workgroup.
__device__ __forceinline__ void phase_reduce_ln(const Args& a, rsrc_t PLr, int np, const float* __restrict__ bias, rsrc_t Rr,
                                                const float* __restrict__ gamma, const float* __restrict__ beta, rsrc_t Or, int T,
                                                f32x4* comb, float* red8, int wg, int G, int tid, int lane, int wave) {
    const int H = a.H, HQ = H >> 2;
    const int npg = min(kWaves, kThreads / HQ);
    const int pg = tid / HQ, q = tid - pg * HQ;
    for (int row = wg; row < T; row += G) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (pg < npg) {
            for (int p0 = pg; p0 < np; p0 += 8 * npg) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int pl = p0 + u * npg;
                    if (pl < np) v[u] = ld4(PLr, ((pl * kTmax + row) * H + 4 * q) * 4);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (p0 + u * npg < np) s += v[u];
            }
            comb[pg * HQ + q] = s;
        }
        __syncthreads();
        const bool active = tid < HQ;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            v = comb[tid];
            for (int j = 1; j < npg; ++j) v += comb[j * HQ + tid];
            v = (v + *reinterpret_cast<const f32x4*>(bias + 4 * tid)) + ld4(Rr, (row * H + 4 * tid) * 4);
        }
        row_layernorm(v, active, tid, H, a.eps, gamma, beta, Or, row * H * 4, red8, lane, wave);
        __syncthreads();  // comb free for the next row
    }
}

template <int MT, int HC>
__global__ __launch_bounds__(kThreads) void encoder_walk_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // ---- LDS map -------------------------------------------------------------------------------------------------------
    int* s_tok_id = reinterpret_cast<int*>(lds);   // [64] packed token -> vocabulary id
    int* s_tok_pos = s_tok_id + 64;                 // [64] packed token -> position id
    int* s_slot_p = s_tok_pos + 64;                 // [64] token slot (b * S + t) -> packed token, or -1
    int* s_seq = s_slot_p + 64;                     // [B + 1 <= 65] first packed token of each sentence
    float* red8 = lds + 264;                        // [8]
    float* work = lds + 272;                        // phase scratch (16-byte aligned: 272 * 4 = 1088)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x, G = gridDim.x;
    const int H = a.H, F = a.F, hd = a.hd;
    const int r = lane & 15, g = lane >> 4;

    // ---- packing: every workgroup derives it from the mask (B * S <= 64 slots = one ballot) --------------------------------
    if (wave == 0) {
        const int slots = a.B * a.S;
        const bool v = lane < slots && a.mask[lane] != 0;
        const unsigned long long m = __ballot(v);
        const int p = __popcll(m & ((1ull << lane) - 1ull));
        const int b = lane < slots ? lane / a.S : 0, t = lane - b * a.S;
        const int lo = b * a.S;
        const int sb = __popcll(m & (lo >= 64 ? ~0ull : (1ull << lo) - 1ull));
        if (lane < slots) s_slot_p[lane] = v ? p : -1;
        if (v) {
            const int id = a.ids[lane];
            s_tok_id[p] = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
            s_tok_pos[p] = a.position_offset > 0 ? (p - sb) + a.position_offset : t;
        }
        if (lane <= a.B) {
            const int e = lane * a.S;
            s_seq[lane] = __popcll(m & (e >= 64 ? ~0ull : (1ull << e) - 1ull));
        }
    }
    __syncthreads();
    const int T = s_seq[a.B];
    const int mtc = (T + 15) >> 4;
    unsigned int epoch = 0;
    const rsrc_t Xr = make_rsrc(a.X), X1r = make_rsrc(a.X1), Qr = make_rsrc(a.QKV), PLr = make_rsrc(a.PL);

    if (T > 0) {
        // ---- embeddings + LayerNorm -> X (one workgroup per row) ---------------------------------------------------------------
        {
            const int HQ = H >> 2;
            for (int p = wg; p < T; p += G) {
                const bool active = tid < HQ;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (active) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.word + (int64_t)s_tok_id[p] * H + 4 * tid);
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.type + 4 * tid);
                    const f32x4 p4 = *reinterpret_cast<const f32x4*>(a.pos + (int64_t)s_tok_pos[p] * H + 4 * tid);
                    v = (w4 + t4) + p4;  // HF: inputs_embeds + token_type, then + position
                }
                row_layernorm(v, active, tid, H, a.eps, a.embg, a.embb, Xr, p * H * 4, red8, lane, wave);
            }
        }
        grid_sync(a.bar, epoch, G);

        const float qscale = 1.4426950408889634f / sqrtf((float)hd);  // log2(e) / sqrt(hd): softmax by exp2
        const int hd4 = hd + 4;
        for (int layer = 0; layer < a.nlayers; ++layer) {
            const LayerPtrs L = a.layers[layer];
            // ---- QKV: column units over 3H -> QKV[T, 3H] ----------------------------------------------------------------------
            {
                f32x4* red = reinterpret_cast<f32x4*>(work);
                for (int u = wg; u < (3 * H) >> 4; u += G) {
                    f32x4 acc[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    colunit_gemm<MT, HC>(L.wqkv + (int64_t)u * 16 * H, H, Xr, T, mtc, acc, lane, wave);
                    colunit_publish<MT>(acc, mtc, red, lane, wave);
                    __syncthreads();
                    if (wave < mtc) {
                        const int tok = wave * 16 + r, n = u * 16 + 4 * g;
                        const f32x4 v = colunit_total<MT>(red, wave, lane) + *reinterpret_cast<const f32x4*>(L.bqkv + n);
                        if (tok < T) st4(Qr, (tok * 3 * H + n) * 4, v);
                    }
                    __syncthreads();
                }
            }
            grid_sync(a.bar, epoch, G);

            // ---- attention per (sentence, head, column split) + out-projection partial -> plane[head] --------------------------------
            {
                float* Qs = work;                 // [64][hd + 4]  queries, scaled
                float* Ks = Qs + 64 * hd4;        // [64][hd + 4]
                float* Vt = Ks + 64 * hd4;        // [hd][68]      V transposed
                float* Cs = Vt + hd * 68;         // [64][hd + 4]  context of this head
                const int ntiles = H >> 4;
                const int ntu = (ntiles + a.nsplit - 1) / a.nsplit;
                const int units = a.B * a.heads * a.nsplit;
                for (int u = wg; u < units; u += G) {
                    const int ns = u % a.nsplit, bh = u / a.nsplit;
                    const int h = bh % a.heads, b = bh / a.heads;
                    const int s0 = s_seq[b], len = s_seq[b + 1] - s0;
                    if (len <= 0) continue;  // uniform over the workgroup
                    const int mtb = (len + 15) >> 4;
                    // Q, K, V of (sentence b, head h): rows < len from QKV, the rest of the 16-row tiles zero
                    const int q4 = hd >> 2;  // float4 per row
                    for (int e = tid; e < mtb * 16 * q4; e += kThreads) {
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
                        for (int j = 0; j < 4; ++j) Vt[(4 * c4 + j) * 68 + row] = vv[j];
                    }
                    __syncthreads();
                    if (wave < mtb) {
                        const int qi = wave;
                        f32x4 sc[MT];
#pragma unroll
                        for (int kj = 0; kj < MT; ++kj) {
                            sc[kj] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (kj < mtb) {
                                for (int c = 0; c < (hd >> 4); ++c) {
                                    const f32x4 ka = *reinterpret_cast<const f32x4*>(Ks + (16 * kj + r) * hd4 + 16 * c + 4 * g);
                                    const f32x4 qb = *reinterpret_cast<const f32x4*>(Qs + (16 * qi + r) * hd4 + 16 * c + 4 * g);
#pragma unroll
                                    for (int m = 0; m < 4; ++m)
                                        sc[kj] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[m], qb[m], sc[kj], 0, 0, 0);
                                }
                            }
                        }
                        // sc[kj][v] = score(query 16 qi + r, key 16 kj + 4 g + v): softmax over the keys of this query
                        float mx = -INFINITY;
#pragma unroll
                        for (int kj = 0; kj < MT; ++kj)
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const bool ok = kj < mtb && 16 * kj + 4 * g + v < len;
                                sc[kj][v] = ok ? sc[kj][v] : -INFINITY;
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
                                if (kj < mtb) {
                                    const f32x4 va = *reinterpret_cast<const f32x4*>(Vt + (16 * dt + r) * 68 + 16 * kj + 4 * g);
#pragma unroll
                                    for (int m = 0; m < 4; ++m) o = __builtin_amdgcn_mfma_f32_16x16x4f32(va[m], sc[kj][m], o, 0, 0, 0);
                                }
                            }
                            // o[v] = ctx[query 16 qi + r][d = 16 dt + 4 g + v]
                            *reinterpret_cast<f32x4*>(Cs + (16 * qi + r) * hd4 + 16 * dt + 4 * g) = o;
                        }
                    }
                    __syncthreads();
                    // plane[h][s0 + query][n] = sum_d ctx[query][d] * Wo[n][h * hd + d] for this unit's column tiles
                    const int nt0 = ns * ntu, nt1 = min(ntiles, nt0 + ntu);
                    for (int nt = nt0 + wave; nt < nt1; nt += kWaves) {
                        f32x4 wa[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < (hd >> 4))
                                wa[c] = *reinterpret_cast<const f32x4*>(L.wo + (int64_t)(16 * nt + r) * H + h * hd + 16 * c + 4 * g);
#pragma unroll
                        for (int qi = 0; qi < MT; ++qi) {
                            if (qi < mtb) {
                                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    if (c < (hd >> 4)) {
                                        const f32x4 cb = *reinterpret_cast<const f32x4*>(Cs + (16 * qi + r) * hd4 + 16 * c + 4 * g);
#pragma unroll
                                        for (int m = 0; m < 4; ++m) o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][m], cb[m], o, 0, 0, 0);
                                    }
                                }
                                if (16 * qi + r < len) st4(PLr, ((h * kTmax + s0 + 16 * qi + r) * H + 16 * nt + 4 * g) * 4, o);
                            }
                        }
                    }
                    __syncthreads();  // Q / K / V / ctx tiles free for the next unit
                }
            }
            grid_sync(a.bar, epoch, G);

            // ---- x1 = LN(sum of head planes + bo + x) -> X1 ---------------------------------------------------------------------
            phase_reduce_ln(a, PLr, a.heads, L.bo, Xr, L.ln1g, L.ln1b, X1r, T, reinterpret_cast<f32x4*>(work), red8, wg, G, tid, lane,
                            wave);
            grid_sync(a.bar, epoch, G);

            // ---- FFN: workgroup wg < np3 owns the 16-wide slices wg, wg + np3, ... of F -> plane[wg] ---------------------------------
            if (wg < a.np3) {
                f32x4* red = reinterpret_cast<f32x4*>(work);
                float* hbuf = work + kWaves * MT * 64 * 4;  // [64][20]: GELU(x1 W1_slice^T + b1)
                f32x4 acc2[HC][MT];
#pragma unroll
                for (int i = 0; i < HC; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc2[i][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int j = wg; j < (F >> 4); j += a.np3) {
                    f32x4 acc[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    colunit_gemm<MT, HC>(L.w1 + (int64_t)j * 16 * H, H, X1r, T, mtc, acc, lane, wave);
                    colunit_publish<MT>(acc, mtc, red, lane, wave);
                    __syncthreads();
                    if (wave < mtc) {
                        f32x4 v = colunit_total<MT>(red, wave, lane) + *reinterpret_cast<const f32x4*>(L.b1 + 16 * j + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
                        *reinterpret_cast<f32x4*>(hbuf + (wave * 16 + r) * 20 + 4 * g) = v;
                    }
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int nt = wave + i * kWaves;
                        if (nt < (H >> 4)) {
                            const f32x4 wa = *reinterpret_cast<const f32x4*>(L.w2 + (int64_t)(16 * nt + r) * F + 16 * j + 4 * g);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                if (mt < mtc) {
                                    const f32x4 hb = *reinterpret_cast<const f32x4*>(hbuf + (mt * 16 + r) * 20 + 4 * g);
#pragma unroll
                                    for (int m = 0; m < 4; ++m)
                                        acc2[i][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m], hb[m], acc2[i][mt], 0, 0, 0);
                                }
                            }
                        }
                    }
                    __syncthreads();  // red / hbuf free for the next slice
                }
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int nt = wave + i * kWaves;
                    if (nt < (H >> 4)) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            if (mt < mtc && mt * 16 + r < T) st4(PLr, ((wg * kTmax + mt * 16 + r) * H + 16 * nt + 4 * g) * 4, acc2[i][mt]);
                    }
                }
            }
            grid_sync(a.bar, epoch, G);

            // ---- x = LN(sum of FFN planes + b2 + x1) -> X -----------------------------------------------------------------------
            phase_reduce_ln(a, PLr, min(a.np3, F >> 4), L.b2, X1r, L.ln2g, L.ln2b, Xr, T, reinterpret_cast<f32x4*>(work), red8, wg, G, tid,
                            lane, wave);
            grid_sync(a.bar, epoch, G);
        }
    }

    // ---- pooling + L2 normalise (average_pool + F.normalize(eps = 1e-12); pooling 1: first valid token) ------------------------
    {
        const int HQ = H >> 2;
        for (int b = wg; b < a.B; b += G) {
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
            for (int slot = wg; slot < a.B * a.S; slot += G) {
                const int p = s_slot_p[slot];
                if (tid < HQ) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (p >= 0) v = ld4(Xr, (p * H + 4 * tid) * 4);
                    *reinterpret_cast<f32x4*>(a.hidden + (int64_t)slot * H + 4 * tid) = v;
                }
            }
        }
    }
    // ---- the last workgroup out re-arms the barrier words for the next launch -------------------------------------------------
    __syncthreads();
    if (tid == 0) {
        const unsigned int left = __hip_atomic_fetch_add(a.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (unsigned int)G - 1u) {
            __hip_atomic_store(a.bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// LDS bytes of a launch
inline size_t lds_bytes(int mt, int H, int hd) {
    const size_t colunit = (size_t)kWaves * mt * 64 * 16 + 64 * 20 * 4;             // wave partials + GELU tile
    const size_t attn = (size_t)(3 * 64 * (hd + 4) + hd * 68) * 4;                  // Q, K, ctx, V^T
    const size_t reduce = (size_t)kThreads * 16;                                    // plane-group partials
    return 272 * 4 + std::max(colunit, std::max(attn, reduce));
}

}  // namespace walk
