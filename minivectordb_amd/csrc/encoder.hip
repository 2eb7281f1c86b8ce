// encoder.hip — BERT-architecture sentence encoder forward (multilingual-e5-small/large shape) +
// masked mean pool + L2 normalise, hand-written for gfx950.
//
// Replaces, on MiniVectorDB's embed path (reference minivectordb/embedding_model.py):
//   outputs = self.model(**batch_dict)                      :66   (HF BertModel / XLMRobertaModel)
//   average_pool(last_hidden_state, attention_mask)         :50-53, :67
//   F.normalize(embeddings, p=2, dim=1)                     :70
// Op order follows transformers' modeling_bert.py (embeddings+LN; per layer QKV, scaled softmax
// attention with key mask, out-proj + residual + LN, FFN1 + erf-GELU, FFN2 + residual + LN).
//
// MI355X design
//   * varlen packing: only VALID tokens are materialised (T = sum of sequence lengths); padded
//     positions cost nothing in the GEMMs and need no mask in attention (keys of a sequence are
//     all valid).  Packing (counts, prefix sums, gather map) is computed on the device, so the
//     whole forward is enqueued without a host round trip; grids are sized for B*S and tiles
//     beyond T exit early.
//   * GEMMs on the exact-fp32 matrix cores: v_mfma_f32_32x32x2_f32, 128x128x16 (or 64x64x16) block
//     tile, 4 waves, LDS tiles stored k-major so every ds_read_b32 of a fragment is conflict-free,
//     register-staged global->LDS double buffering (T14 split).  Results are bit-for-bit an fp32
//     fmaf chain in k order (parity with the reference's fp32).
//   * attention on the same MFMA ("swapped" K·Q^T, softmax in registers, P^T taken from the
//     accumulator as the next product's operand); default arithmetic: split-precision fp16 x 3 (compute = 2, below).
//   * the whole forward is captured once per shape and replayed as one hipGraph.
//   * Q/K/V projections fused into one [3H,H] GEMM (weights concatenated once at create time).
//   * bias / erf-GELU / residual fused into the GEMM epilogue; LayerNorm and pooling are
//     one-wave-per-row kernels (HBM/L2-bound, tiny).
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <map>
#include <tuple>

#include "common.hpp"

using namespace mvdb;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

}  // namespace
#include "encoder_walk.hpp"
namespace {

// =================================================================================================
// packing
// =================================================================================================
// one wave per sequence: rank of every valid token inside its sequence, and the count
__global__ __launch_bounds__(64) void seq_rank_kernel(const int32_t* __restrict__ mask, int S,
                                                      int* __restrict__ rank, int* __restrict__ count) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int base = 0;
    for (int t0 = 0; t0 < S; t0 += 64) {
        const int t = t0 + lane;
        const bool v = t < S && mask[(int64_t)b * S + t] != 0;
        const unsigned long long m = __ballot(v);
        const int r = base + __popcll(m & ((1ull << lane) - 1ull));
        if (t < S) rank[(int64_t)b * S + t] = v ? r : -1;
        base += __popcll(m);
    }
    if (lane == 0) count[b] = base;
}

// single block: exclusive scan of count[B] -> seq_start[B+1]
__global__ __launch_bounds__(256) void seq_scan_kernel(const int* __restrict__ count, int B,
                                                       int* __restrict__ seq_start) {
    __shared__ int part[256];
    const int tid = threadIdx.x;
    const int per = (B + 255) / 256;
    const int lo = tid * per, hi = min(B, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += count[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) {
            const int v = part[i];
            part[i] = run;
            run += v;
        }
        seq_start[B] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int i = lo; i < hi; ++i) {
        seq_start[i] = run;
        run += count[i];
    }
}

// packed token p of sequence b: source id, position id, owning sequence
__global__ __launch_bounds__(256) void pack_fill_kernel(const int32_t* __restrict__ ids,
                                                        const int* __restrict__ rank,
                                                        const int* __restrict__ seq_start, int S,
                                                        int position_offset, int vocab,
                                                        int* __restrict__ tok_id,
                                                        int* __restrict__ tok_pos, int* __restrict__ tok_src) {
    const int b = blockIdx.x;
    const int s0 = seq_start[b];
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        const int r = rank[(int64_t)b * S + t];
        if (r >= 0) {
            const int p = s0 + r;
            const int id = ids[(int64_t)b * S + t];
            tok_id[p] = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never read outside the table
            // BERT: absolute position t.  XLM-R (position_offset = padding_idx + 1 > 0):
            // cumsum(mask) + padding_idx = (r + 1) + (position_offset - 1)
            tok_pos[p] = position_offset > 0 ? r + position_offset : t;
            tok_src[p] = b * S + t;
        }
    }
}

// The three kernels above as ONE launch for small batches (B <= kPackSmallB: one sentence per call is the reference's shape):
// four waves share the sentences, lane 0 of the workgroup scans the counts.  Also clears the forward's overflow word (a
// memset node of its own otherwise).  Same outputs, bit for bit.
constexpr int kPackSmallB = 64;
__global__ __launch_bounds__(256) void pack_small_kernel(const int32_t* __restrict__ mask, const int32_t* __restrict__ ids, int B, int S,
                                                         int position_offset, int vocab, int* __restrict__ rank, int* __restrict__ count,
                                                         int* __restrict__ seq_start, int* __restrict__ tok_id, int* __restrict__ tok_pos,
                                                         int* __restrict__ tok_src, unsigned int* __restrict__ flag) {
    __shared__ int s_start[kPackSmallB + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0 && flag) *flag = 0u;
    // (mask and ids may live in host-mapped memory — the host entry of one sentence per call —: eight words per lane are in
    //  flight at once, the ids fetched together with the mask; one load -> ballot at a time was 12 us for 256 tokens)
    const bool one_pass = B <= 4 && S <= 512;  // a wave meets ONE (sentence, 512-slot block): its ids stay in registers
    int iv0[8];
    for (int b = wave; b < B; b += 4) {
        int base = 0;
        for (int t00 = 0; t00 < S; t00 += 512) {
            int mv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t00 + 64 * u + lane;
                mv[u] = t < S ? mask[(int64_t)b * S + t] : 0;
                iv0[u] = one_pass && t < S ? ids[(int64_t)b * S + t] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t00 + 64 * u + lane;
                const bool v = mv[u] != 0;
                const unsigned long long m = __ballot(v);
                const int r = base + __popcll(m & ((1ull << lane) - 1ull));
                if (t < S) rank[(int64_t)b * S + t] = v ? r : -1;  // (read back below by the lane that wrote it)
                base += __popcll(m);
            }
        }
        if (lane == 0) {
            count[b] = base;
            s_start[b] = base;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < B; ++i) {
            const int v = s_start[i];
            s_start[i] = run;
            seq_start[i] = run;
            run += v;
        }
        s_start[B] = run;
        seq_start[B] = run;
    }
    __syncthreads();
    for (int b = wave; b < B; b += 4) {
        const int s0 = s_start[b];
        for (int t00 = 0; t00 < S; t00 += 512) {
            int iv[8], rv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t00 + 64 * u + lane;
                rv[u] = t < S ? rank[(int64_t)b * S + t] : -1;
                iv[u] = one_pass ? iv0[u] : (t < S ? ids[(int64_t)b * S + t] : 0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t00 + 64 * u + lane;
                if (rv[u] >= 0) {
                    const int p = s0 + rv[u];
                    tok_id[p] = iv[u] < 0 ? 0 : (iv[u] >= vocab ? vocab - 1 : iv[u]);  // never read outside the table
                    tok_pos[p] = position_offset > 0 ? rv[u] + position_offset : t;
                    tok_src[p] = b * S + t;
                }
            }
        }
    }
}

// One element per lane of a 32-column group -> the group's (hi | lo) line: `line` = the 128 bytes that hold columns
// c0 .. c0 + 31 of a row in the [row][K / 32][hi 32 | lo 32] image (as a float pointer: the fp32 position of (row, c0) —
// the image has the bytes of the fp32 matrix), c = the lane's column in the group.  Lanes c and c ^ 1 trade halves so that
// every lane stores ONE dword: even lanes the hi pieces of columns (c, c + 1), odd lanes the lo pieces of (c - 1, c).
// Both lanes of a pair must call it (`ok` = store or not, the same for both).
__device__ __forceinline__ void x3_pair_store(float* line, int c, float v, bool ok) {
    // v is made opaque first: otherwise hipcc may contract the producer's multiply-add INTO the conversion
    // (v_fma_mixlo_f16: one rounding of the exact fma to fp16) at some call sites and not at others, so that the same
    // row got different (hi, lo) bits on the branch-free and on the edge path of an epilogue — the result of a sentence
    // then depended, in the last bit, on where it sat in the batch
    asm volatile("" : "+v"(v));
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    union { _Float16 f[2]; uint32_t u; } mine;
    mine.f[0] = h;
    mine.f[1] = l;
    const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine.u, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
    const bool odd = c & 1;
    const uint32_t word = odd ? (other >> 16) | (mine.u & 0xffff0000u) : (mine.u & 0xffffu) | (other << 16);
    if (ok) reinterpret_cast<uint32_t*>(line)[odd ? 16 + (c >> 1) : (c >> 1)] = word;
}

// =================================================================================================
// row kernels: one wave per token row, VPT = ceil(H/64) values per lane in registers
// =================================================================================================
#ifdef MVDB_X3_ABLATE
// ablation build only (MVDB_LN_SKIP_X=1, results INVALID): the LayerNorm kernels do not store the fp32 row — the upper bound of
// what keeping the residual stream ONLY as the (hi | lo) image could save (round-5 review item; profiles/r06_residual_image_bound.txt)
__device__ int g_ln_skip_x = 0;
#define MVDB_LN_STORE_X (!g_ln_skip_x)
#else
#define MVDB_LN_STORE_X true
#endif
template <int VPT, bool FULL = false>
__device__ __forceinline__ void wave_layernorm(float (&v)[VPT], int H, int lane, float eps,
                                               const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float* __restrict__ out,
                                               float* __restrict__ outp) {
    // outp: NULL, or the row in the (hi | lo) fp16 image that feeds the split-precision GEMMs (needs H % 32 == 0)
    // gamma / beta are fetched BEFORE the reductions (their latency hides under the shuffles) and, when the row fills
    // every lane slot (H = 64 VPT: 384, 1024), the tail is branch-free: with a bounds test per element hipcc emitted a
    // loop of load, s_waitcnt vmcnt(0), store — six serialised round trips per row
    constexpr bool full = FULL;  // the row fills every lane slot: H == 64 VPT
    float g[VPT], bt[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = lane + i * 64;
        const int cc = full || c < H ? c : 0;
        g[i] = gamma[cc];
        bt[i] = beta[cc];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) s += (full || lane + i * 64 < H) ? v[i] : 0.f;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    const float mean = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const float dlt = (full || lane + i * 64 < H) ? v[i] - mean : 0.f;
        q += dlt * dlt;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);  // biased variance, eps inside the sqrt
    if (full) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const float r = (v[i] - mean) * rstd * g[i] + bt[i];
            if (MVDB_LN_STORE_X) out[lane + i * 64] = r;
            if (outp) x3_pair_store(outp + i * 64 + (lane & 32), lane & 31, r, true);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = lane + i * 64;
        const float r = (v[i] - mean) * rstd * g[i] + bt[i];
        if (c < H) out[c] = r;
        if (outp) x3_pair_store(outp + (c < H ? (c & ~31) : 0), lane & 31, r, c < H);
    }
}

// x[p,:] = LN(word[id] + pos[pos_id] + type[0])
template <int VPT>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int* __restrict__ tok_id,
                                                       const int* __restrict__ tok_pos,
                                                       const int* __restrict__ seq_start, int B,
                                                       const float* __restrict__ word,
                                                       const float* __restrict__ pos,
                                                       const float* __restrict__ type,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int H,
                                                       float* __restrict__ x, float* __restrict__ xp) {
    const int T = seq_start[B];
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= T) return;
    const float* w = word + (int64_t)tok_id[p] * H;
    const float* ps = pos + (int64_t)tok_pos[p] * H;
    float v[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = lane + i * 64;
        v[i] = c < H ? (w[c] + type[c]) + ps[c] : 0.f;  // HF: inputs_embeds + token_type, then + position
    }
    wave_layernorm<VPT>(v, H, lane, eps, gamma, beta, x + (int64_t)p * H, xp ? xp + (int64_t)p * H : nullptr);
}

// x[p,:] = LN(y[p,:])   (y already holds dense(...) + bias + residual from the GEMM epilogue)
template <int VPT, bool FULL>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ y,
                                                 const int* __restrict__ seq_start, int B,
                                                 const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, int H,
                                                 float* __restrict__ x, float* __restrict__ xp) {
    const int T = seq_start[B];
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= T) return;
    float v[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = lane + i * 64;
        if (FULL)
            v[i] = y[(int64_t)p * H + c];
        else
            v[i] = c < H ? y[(int64_t)p * H + c] : 0.f;
    }
    wave_layernorm<VPT, FULL>(v, H, lane, eps, gamma, beta, x + (int64_t)p * H, xp ? xp + (int64_t)p * H : nullptr);
}

// x[p,:] = LN(P[0][p,:] + ... + P[nparts - 1][p,:] + bias + x[p,:]): the planes of a split-K GEMM (EPI_PARTIAL), summed in plane
// order, then the bias and the residual row (x itself: read whole before it is overwritten, by the wave that owns the row)
// NP = planes (1 .. 4; 0: a run-time count): with the count known every load of a row is issued before the first add (the
// run-time loop compiled to load -> wait -> add per plane: 13.4 us per launch at T = 512, H = 1024)
template <int VPT, bool FULL, int NP = 0>
__global__ __launch_bounds__(256) void ln_partials_kernel(const float* __restrict__ P, int nparts, int64_t plane,
                                                          const float* __restrict__ bias, const int* __restrict__ seq_start, int B,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          int H, float* x, float* __restrict__ xp) {
    const int T = seq_start[B];
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= T) return;
    float v[VPT];
    if constexpr (NP > 0) {
        constexpr int GP = NP <= 4 ? NP : NP / 2;   // planes per group: every load of a group in flight before its adds
        float rv[VPT], bv[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = (FULL || lane + i * 64 < H) ? lane + i * 64 : 0;
            rv[i] = x[(int64_t)p * H + c];
            bv[i] = bias[c];
            v[i] = 0.f;
        }
#pragma unroll
        for (int g0 = 0; g0 < NP; g0 += GP) {
            float pv[GP][VPT];
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int c = (FULL || lane + i * 64 < H) ? lane + i * 64 : 0;
#pragma unroll
                for (int z = 0; z < GP; ++z) pv[z][i] = P[(g0 + z) * plane + (int64_t)p * H + c];
            }
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                float a = g0 == 0 ? pv[0][i] : v[i] + pv[0][i];   // plane order throughout
#pragma unroll
                for (int z = 1; z < GP; ++z) a += pv[z][i];
                v[i] = a;
            }
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) v[i] = (FULL || lane + i * 64 < H) ? (v[i] + bv[i]) + rv[i] : 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = lane + i * 64;
            const bool ok = FULL || c < H;
            float a = 0.f;
            if (ok) {
                a = P[(int64_t)p * H + c];
                for (int z = 1; z < nparts; ++z) a += P[z * plane + (int64_t)p * H + c];
                a = (a + bias[c]) + x[(int64_t)p * H + c];
            }
            v[i] = a;
        }
    }
    wave_layernorm<VPT, FULL>(v, H, lane, eps, gamma, beta, x + (int64_t)p * H, xp ? xp + (int64_t)p * H : nullptr);
}

// out[b,:] = normalize(mean over the sequence's tokens)   — average_pool + F.normalize(eps=1e-12)
// pooling == 1: the first valid token (CLS) instead of the mean — BGE-M3's dense_vecs
// flag: one word the launch ORs 1 into when a pooled row of a NON-empty sentence is not finite (an activation left the fp16 range
// of the split-precision GEMMs): callers of the device entry test it without reading the embeddings back
// Sentences of more than kPoolChunk tokens are summed in CHUNKS of kPoolChunk tokens on workgroups of their own (grid.y =
// ceil(S / kPoolChunk)): chunk partials in token order, then the partials in chunk order by the last workgroup to arrive — a
// sentence's sum depends on its length only, and up to kPoolChunk tokens it is the plain token-order sum.  (One workgroup
// walking 512 tokens eight at a time took 47.6 us of a 0.90 ms one-sentence forward; profiles/r06_long_sentence_chain.txt.)
// part: [B][grid.y][H] chunk partials, ctr: [B] arrival counters (zero between launches: the last workgroup re-arms its own).
constexpr int kPoolChunk = 64;
__global__ __launch_bounds__(256) void pool_norm_kernel(const float* __restrict__ x,
                                                        const int* __restrict__ seq_start, int H,
                                                        int pooling, float* __restrict__ out, unsigned int* __restrict__ flag,
                                                        float* part, unsigned int* ctr) {
    __shared__ float red[256];
    __shared__ int s_last;
    const int b = blockIdx.x, j = blockIdx.y;
    const int s0 = seq_start[b], len = seq_start[b + 1] - s0;
    const int span = pooling == 1 ? (len > 0 ? 1 : 0) : len;
    const int nch = max(1, (span + kPoolChunk - 1) / kPoolChunk);  // live chunks of this sentence
    if (j >= nch) return;
    const int t0 = j * kPoolChunk, t1 = min(span, t0 + kPoolChunk);
    float sq = 0.f;
    float ev[4];  // this thread's means (H <= 1024 = 4 x 256 columns): `out` may be host-mapped memory, never read back
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + i * 256;
        ev[i] = 0.f;
        if (c >= H) continue;
        float s = 0.f;
        int t = t0;
        // eight loads in flight, added in token order (the same sum as a plain loop, which hipcc compiles to one
        // dependent load -> add per token: 32 serialised round trips at S = 32)
        for (; t + 8 <= t1; t += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(int64_t)(s0 + t + u) * H + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; t < t1; ++t) s += x[(int64_t)(s0 + t) * H + c];
        if (nch == 1) {
            const float e = s / (float)span;  // empty sequence -> NaN, as the reference's 0/0
            ev[i] = e;
            sq += e * e;
        } else {
            __hip_atomic_store(part + ((int64_t)b * gridDim.y + j) * H + c, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (nch > 1) {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int seen = __hip_atomic_fetch_add(ctr + b, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = seen == (unsigned int)nch - 1u;
            if (s_last) __hip_atomic_store(ctr + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c >= H) continue;
            float s = __hip_atomic_load(part + ((int64_t)b * gridDim.y) * H + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int z = 1; z < nch; ++z)
                s += __hip_atomic_load(part + ((int64_t)b * gridDim.y + z) * H + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float e = s / (float)span;
            ev[i] = e;
            sq += e * e;
        }
    }
    red[threadIdx.x] = sq;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    const float denom = fmaxf(sqrtf(red[0]), 1e-12f);
    if (flag && threadIdx.x == 0 && span > 0 && !(red[0] < INFINITY)) atomicOr(flag, 1u);  // NaN or inf among the row's squares
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < H) out[(int64_t)b * H + c] = ev[i] / denom;
    }
}

// hidden[b,t,:] = x[packed(b,t),:] for valid tokens, 0 for padding
__global__ __launch_bounds__(256) void unpack_hidden_kernel(const float* __restrict__ x,
                                                            const int* __restrict__ rank,
                                                            const int* __restrict__ seq_start, int S,
                                                            int H, float* __restrict__ hidden) {
    const int bt = blockIdx.x;
    const int b = bt / S;
    const int r = rank[bt];
    const float* src = r >= 0 ? x + (int64_t)(seq_start[b] + r) * H : nullptr;
    for (int c = threadIdx.x; c < H; c += blockDim.x) hidden[(int64_t)bt * H + c] = src ? src[c] : 0.f;
}

// =================================================================================================
// GEMM  C[T,N] = A[T,K] · W[N,K]^T + bias (+ epilogue), exact fp32 on v_mfma_f32_32x32x2_f32
// =================================================================================================
// EPI_BIAS_QKV (split-precision GEMM only): bias, the first `qcols` columns (the queries) scaled by `qscale`
// (log2(e) / sqrt(head_dim): the attention kernel's score scale), output written as (hi | lo) fp16 lines
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESIDUAL = 2, EPI_BIAS_QKV = 3,
       EPI_PARTIAL = 4 };  // split-K (split-precision GEMM only): the bare partial sum acc / wscale of this workgroup's K range

constexpr int GBK = 16;  // k granularity of the exact fp32 GEMM tiles (hidden / intermediate must be multiples)

// TI = 32x32 MFMA tiles per wave and dimension: TI = 2 -> 128x128 block tile (best reuse), TI = 1 ->
// 64x64 (4x the blocks: used when the 128x128 grid would leave CUs idle, e.g. N = 384 at T = 8192).
// XCD-aware tile mapping: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own 4-MiB L2.
// With the natural order the tiles of one A row band land on all 8 XCDs and every L2 ends up streaming the whole of A
// and W (15 MB at T = 8192, H = 384); remapped, XCD k works on the k-th contiguous eighth of the ACTIVE tile list
// (row bands below T: a ragged batch must not leave the last XCDs idle), i.e. on 1/8 of A's rows plus W.
// Returns false for workgroups beyond the active tiles.
__device__ __forceinline__ bool xcd_tile(int T, int BM, int& bx, int& by) {
    const unsigned gx = gridDim.x;
    const unsigned active = gx * (unsigned)((T + BM - 1) / BM);
    unsigned lin = blockIdx.y * gx + blockIdx.x;
    if (lin >= active) return false;
    const unsigned chunk = active >> 3;
    if (lin < (chunk << 3)) lin = (lin & 7u) * chunk + (lin >> 3);
    bx = (int)(lin % gx);
    by = (int)(lin / gx);
    return true;
}

template <int EPI, int TI, int BKT>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const float* __restrict__ A,
                                                            const float* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ R,
                                                            float* __restrict__ C,
                                                            const int* __restrict__ Tptr, int N, int K) {
    constexpr int BM = 64 * TI, BN = 64 * TI, LD = BM + 4;  // LDS row stride (floats): +4 pad
    constexpr int KQ = BKT / 16;                             // 16-float k groups per LDS tile
    const int T = *Tptr;
    int bx, by;
    if (!xcd_tile(T, BM, bx, by)) return;
    const int m0 = by * BM;
    const int n0 = bx * BN;
    __shared__ float lds[2 * 2 * BKT * LD];  // [buf][A|B][k][row]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // global -> register staging: thread owns rows lr + 64*i of each tile, 4 consecutive k
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const float* a_ptr[TI];
    const float* w_ptr[TI];
    bool a_ok[TI], w_ok[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int ra = m0 + lr + i * 64, rw = n0 + lr + i * 64;
        a_ok[i] = ra < T;
        w_ok[i] = rw < N;
        a_ptr[i] = A + (int64_t)(a_ok[i] ? ra : 0) * K + lk;
        w_ptr[i] = W + (int64_t)(w_ok[i] ? rw : 0) * K + lk;
    }
    f32x4 ra[TI][KQ], rw[TI][KQ];
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int g = 0; g < KQ; ++g) {
                ra[i][g] = a_ok[i] ? *reinterpret_cast<const f32x4*>(a_ptr[i] + k0 + 16 * g) : f32x4{0, 0, 0, 0};
                rw[i][g] = w_ok[i] ? *reinterpret_cast<const f32x4*>(w_ptr[i] + k0 + 16 * g) : f32x4{0, 0, 0, 0};
            }
    };
    auto stage_write = [&](int buf) {
        float* As = lds + buf * (2 * BKT * LD);
        float* Bs = As + BKT * LD;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int r = lr + i * 64;
#pragma unroll
            for (int g = 0; g < KQ; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    As[(16 * g + lk + j) * LD + r] = ra[i][g][j];
                    Bs[(16 * g + lk + j) * LD + r] = rw[i][g][j];
                }
        }
    };

    f32x16 acc[TI][TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BKT;
    stage_load(0);
    stage_write(0);
    __syncthreads();
    const int fr = lane & 31, fk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage_load((kt + 1) * BKT);  // in flight during the MFMAs below
        const float* As = lds + buf * (2 * BKT * LD);
        const float* Bs = As + BKT * LD;
#pragma unroll
        for (int kk = 0; kk < BKT; kk += 2) {
            float av[TI], bv[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                av[i] = As[(kk + fk) * LD + wm * 32 * TI + i * 32 + fr];
                bv[i] = Bs[(kk + fk) * LD + wn * 32 * TI + i * 32 + fr];
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            stage_write(buf ^ 1);  // the other buffer: last read two iterations ago
            __syncthreads();
        }
    }

    // epilogue.  C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Tiles wholly inside the
    // matrix take a branch-free path (see gemm_x3_dma_kernel: per-element bounds tests serialise the stores).
    if (m0 + BM <= T && n0 + BN <= N) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TI; ++j) {
                const int col = n0 + wn * 32 * TI + j * 32 + fr;
                const float bvv = bias[col];
                const int row0 = m0 + wm * 32 * TI + i * 32 + 4 * fk;
                float res[16];
                if (EPI == EPI_BIAS_RESIDUAL) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) res[r] = R[(int64_t)(row0 + (r & 3) + 8 * (r >> 2)) * N + col];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bvv;
                    if (EPI == EPI_BIAS_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    if (EPI == EPI_BIAS_RESIDUAL) v += res[r];
                    C[(int64_t)(row0 + (r & 3) + 8 * (r >> 2)) * N + col] = v;
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const int col = n0 + wn * 32 * TI + j * 32 + fr;
            if (col >= N) continue;
            const float bvv = bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row < T) {
                    float v = acc[i][j][r] + bvv;
                    if (EPI == EPI_BIAS_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    if (EPI == EPI_BIAS_RESIDUAL) v += R[(int64_t)row * N + col];
                    C[(int64_t)row * N + col] = v;
                }
            }
        }
}

// =================================================================================================
// The same GEMM with LDS-DMA staging (default when K % 32 == 0): tiles go global -> LDS by global_load_lds (no
// VGPR staging, no ds_write: the register-staged kernel above pays 16 ds_write_b32 per thread and K-step, 9 % of the
// S = 32 forward), ROW-major in LDS (128 B = 32 k per row; the DMA image is lane-linear, so the bank swizzle is
// applied to the source: 16-byte slot p of row r receives the row's logical slot p ^ (r & 7)), three stages
// in flight, ONE bare s_barrier per K-step.  A fragment read is one ds_read_b128 per 32-row tile and four MFMA
// steps: lane (fr, fk) takes k = 8 c + 4 fk .. + 3 of row fr, MFMA step j pairs k = 8 c + j with 8 c + 4 + j.  Still an
// exact-fp32 product; only the order of the k terms inside each group of eight differs from the kernel above.
// =================================================================================================
typedef __attribute__((address_space(3))) void* enc_lds_ptr;
typedef const __attribute__((address_space(1))) void* enc_gbl_ptr;

template <int EPI, int TI>
__global__ __launch_bounds__(256) void gemm_f32_dma_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                           const float* __restrict__ bias, const float* __restrict__ R,
                                                           float* __restrict__ C, const int* __restrict__ Tptr, int N,
                                                           int K) {
    constexpr int BM = 64 * TI, BN = 64 * TI;
    constexpr int NST = 3;                          // ring depth
    constexpr int kStage = (BM + BN) * 128;         // bytes: BM + BN rows of 32 floats
    constexpr int NI = (BM + BN) / 8 / 4;           // DMA instructions per wave and stage (8 rows each)
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    const int T = *Tptr;
    int bx, by;
    if (!xcd_tile(T, BM, bx, by)) return;
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fk = lane >> 5;

    // DMA roles: instruction q (0 .. 4 NI - 1) of a stage moves tile rows 8 q .. 8 q + 7 (A rows first, then W rows);
    // wave w issues q = w NI .. w NI + NI - 1.  Per-lane source offsets never change: bases are scalars.
    int64_t voff[NI];  // T K 4 bytes can pass 4 GiB
    const float* sbase[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = wave * NI + i;
        const int row = 8 * q + (lane >> 3);        // row inside the stage image
        const int slot = (lane & 7) ^ (row & 7);
        const bool isA = 8 * q < BM;                // wave-uniform
        int g = isA ? m0 + row : n0 + (row - BM);   // global row of A / W
        const int lim = isA ? T : N;
        g = g < lim ? g : lim - 1;                  // rows past the edge: clamped (masked in the epilogue)
        sbase[i] = isA ? A : W;
        voff[i] = ((int64_t)g * K + 4 * slot) * 4;
    }
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const char* src = reinterpret_cast<const char*>(sbase[i]) + (int64_t)kt * 128;
            __builtin_amdgcn_global_load_lds((enc_gbl_ptr)(src + voff[i]),
                                             (enc_lds_ptr)(gsm + stage * kStage + (wave * NI + i) * 1024), 16, 0, 0);
        }
    };
    // fragment byte offsets inside a stage: A tile i of this wave, W tile j
    int a_off[TI], b_off[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int ra = wm * 32 * TI + i * 32 + fr;
        const int rb = BM + wn * 32 * TI + i * 32 + fr;
        a_off[i] = ra * 128;
        b_off[i] = rb * 128;
    }
    const int sw = fr & 7;  // (row & 7) of every fragment row of this lane: tile bases are multiples of 32

    f32x16 acc[TI][TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    const int nk = K / 32;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");  // this wave's part of stage kt has landed
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's part has; and every wave is done reading stage kt - 1
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) issue(kt + 2, st == 0 ? 2 : st - 1);  // into the buffer of stage kt - 1
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sb = gsm + st * kStage;
        // fragments of chunk c + 1 are read under the MFMAs of chunk c (register double buffer)
        f32x4 av[2][TI], bv[2][TI];
        auto read_chunk = [&](int c, int set) {
            const int so = ((2 * c + fk) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                av[set][i] = *reinterpret_cast<const f32x4*>(sb + a_off[i] + so);
                bv[set][i] = *reinterpret_cast<const f32x4*>(sb + b_off[i] + so);
            }
        };
        read_chunk(0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c + 1 < 4) read_chunk(c + 1, (c + 1) & 1);
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TI; ++j) {
                        if (TI == 1 && (j4 & 1))  // two independent accumulation chains per tile (summed at the end)
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][i][j4], bv[c & 1][j][j4], acc2, 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][i][j4], bv[c & 1][j][j4], acc[i][j], 0, 0, 0);
                    }
        }
        st = st == NST - 1 ? 0 : st + 1;
    }
    if (TI == 1)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += acc2[r];

    // epilogue.  C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Tiles wholly inside the
    // matrix take a branch-free path (see gemm_x3_dma_kernel: per-element bounds tests serialise the stores).
    if (m0 + BM <= T && n0 + BN <= N) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TI; ++j) {
                const int col = n0 + wn * 32 * TI + j * 32 + fr;
                const float bvv = bias[col];
                const int row0 = m0 + wm * 32 * TI + i * 32 + 4 * fk;
                float res[16];
                if (EPI == EPI_BIAS_RESIDUAL) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) res[r] = R[(int64_t)(row0 + (r & 3) + 8 * (r >> 2)) * N + col];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bvv;
                    if (EPI == EPI_BIAS_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    if (EPI == EPI_BIAS_RESIDUAL) v += res[r];
                    C[(int64_t)(row0 + (r & 3) + 8 * (r >> 2)) * N + col] = v;
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const int col = n0 + wn * 32 * TI + j * 32 + fr;
            if (col >= N) continue;
            const float bvv = bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row < T) {
                    float v = acc[i][j][r] + bvv;
                    if (EPI == EPI_BIAS_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    if (EPI == EPI_BIAS_RESIDUAL) v += R[(int64_t)row * N + col];
                    C[(int64_t)row * N + col] = v;
                }
            }
        }
}

// (A persistent form of this kernel — one DMA ring running across tile boundaries, static tile split per XCD — was built in
// round 2 and removed in round 3: +4 % on a full B = 256, S = 32 batch, -30 % on a ragged one; ticket counters -25 %.)
// (compute = 1 — GEMM operands rounded to ONE bf16 product, ~1e-3 on the embeddings — was removed in round 3: the
// split-precision mode below is both faster (2.0 vs 2.7 ms at B = 256, S = 32) and fp32-equivalent.)
constexpr int HBK = 32;       // k granularity of the split-precision images
// =================================================================================================
// compute = 2: split-precision GEMM on the fp16 matrix cores, fp32-equivalent to ~2^-21
//   a = ah + al + ra,  w = wh + wl + rw   (fp16 by RNE: |r| <= 2^-22 |.|, or 2^-25 absolute once the low part is a
//                                          subnormal — gfx950's matrix cores read fp16 subnormals exactly,
//                                          benchmarks/micro/mfma_f16_subnormal.hip)
//   a.w ~ al.wh + ah.wl + ah.wh          three v_mfma_f32_32x32x16_f16 per 16-k block, fp32 accumulate
// The fp32 matrix cores (157 TFLOP/s) bound the exact GEMM at T = 8192; the 16-bit cores are 16x faster, so three
// products cost 3/16 of the fp32 MFMA time.  With bf16 pieces (8 significant bits each) the same scheme left 2^-16 per
// operand: hidden states within 7.5e-5 of transformers' fp32 output, uncomfortably close to the 1e-4 tolerance; fp16
// pieces (11 bits each) leave 2^-22 at the same cost.  fp16's range is what has to be watched: weights are scaled per
// tensor by a power of two that puts max|w| in [2^13, 2^14) (their low parts stay normal numbers; the scale is divided
// out exactly in the epilogue), activations go in unscaled (|a| <= 65504: LayerNorm outputs, attention contexts and
// GELU outputs of a BERT-sized encoder are orders of magnitude below).  Activations stay fp32 in HBM and are split in
// registers on their way from LDS to the MFMA; weights are split once (ensure_x3_weights).
// =================================================================================================
typedef _Float16 x3_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 x3_h2 __attribute__((ext_vector_type(2)));
typedef float x3_f2 __attribute__((ext_vector_type(2)));
typedef unsigned int x3_u2 __attribute__((ext_vector_type(2)));

// max |in[i]| as the bits of a non-negative float (atomicMax on the integer image is order-preserving)
__global__ void absmax_kernel(const float* __restrict__ in, int64_t n, unsigned int* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(in[i]));
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// [N][K] fp32 -> [N][K / 32][hi 32 | lo 32] fp16 of scale * w: the W operand of gemm_x3_dma_kernel
__global__ void f32_split_interleave_kernel(const float* __restrict__ in, _Float16* __restrict__ out, int64_t n,
                                            float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = in[i] * scale;  // exact: power of two
        const _Float16 h = (_Float16)v;
        const int64_t o = (i >> 5) * 64 + (i & 31);  // K % 32 == 0: 32-k blocks never straddle rows
        out[o] = h;
        out[o + 32] = (_Float16)(v - (float)h);
    }
}

// erf to 1.5e-7 absolute (Abramowitz & Stegun 7.1.26: 1 - (a1 t + .. + a5 t^5) exp(-x^2), t = 1 / (1 + p |x|)) in ~14
// instructions; libm's erff is ~50 with branches and was a fifth of the FFN1 GEMM once its MFMAs ran on the 16-bit
// cores.  GELU error <= 0.5 |x| 1.5e-7: below the fp32 rounding noise of the GEMM that feeds it.  Used by the
// split-precision mode only; the exact mode keeps erff.
__device__ __forceinline__ float x3_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float r = fmaf(-p * t, e, 1.0f);
    return copysignf(r, x);
}

// (hi, lo) fp16 pairs of two fp32 values: 6 VALU (cvt_pk, 2 cvt back, 2 sub, cvt_pk)
__device__ __forceinline__ void x3_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    asm volatile("" : "+v"(a), "+v"(b));  // opaque: no contraction of the producer's multiply-add into the conversion
    union { x3_h2 v; uint32_t u; } h, l;
    h.v = __builtin_convertvector(x3_f2{a, b}, x3_h2);
    const x3_f2 back = __builtin_convertvector(h.v, x3_f2);
    l.v = __builtin_convertvector(x3_f2{a - back[0], b - back[1]}, x3_h2);
    hi = h.u;
    lo = l.u;
}

// ---- epilogues on TRANSPOSED accumulator tiles ------------------------------------------------------------------------
// The split-precision GEMMs issue their MFMAs with the WEIGHT fragment as the first operand: D = W_tile A_tile^T, so a
// lane's column of D is a TOKEN ROW of the output (m = lane & 31) and its 16 registers are 16 of the tile's 32 output
// columns: register r = column (r & 3) + 8 (r >> 2) + 4 fh — four groups of four consecutive columns, the other lane
// half (fh ^ 1) holding the groups in between.  What that buys over the natural orientation (one column, 16 rows per lane):
//   * fp32 outputs / residuals move as 16-byte pieces (4 per tile and lane instead of 16 dwords), the two lane halves
//     of one instruction writing adjacent pieces: whole 32-byte sectors;
//   * the (hi | lo) fp16 image of a row is written as 16-byte pieces as well (x3t_store_image: the lane halves trade
//     four-column groups by v_permlane32_swap so that each ends up with eight consecutive columns): 4 stores per tile and
//     lane instead of 16, no DPP pairing;
//   * a row's LayerNorm statistics are sums over the lane's OWN registers plus one exchange with the other half.
// The stores of a big tile were issue-bound: 128 dword stores per lane on a 256 x 256 tile, as long as its K loop at K = 384.
// Where a GEMM's time goes NOW (ablation build, make ABLATE=1: MVDB_GEMM_X3_DBG = 3 K loop only / 4 epilogue only; us per launch,
// full / K loop only / epilogue only): T = 131072  QKV 377 / 277 / 126, FFN1 525 / 352 / 182, N = H + LayerNorm 382 / 222 / 136;
// T = 8192  35.3 / 25.4 / 11.8, 39.4 / 29.1 / 13.1, 33.5 / 22.7 / 10.8 (launch ramp in both parts; profiles/r03_x3_ablations.txt).  The per-tile timeline
// (DBG = 5, benchmarks/x3_timeline.py) of FFN1 at T = 131072: K loop 25.6 us = 47.6k cycles at the 1.87 GHz the chip holds
// there, against 36.9k cycles of MFMA issue; epilogue 10 us until the FIRST wave has issued its stores and ~4.6 more until
// the last one has — GELU + the (hi, lo) split are ~25 VALU instructions per element, 128 elements per lane: the FFN1
// epilogue is bound by VALU issue, not by its 805 MB of stores (staging the outputs through LDS so that every store
// writes whole 128-byte lines changed nothing).
__device__ __forceinline__ void x3t_store_image(unsigned char* line, int fh, const float (&v)[16], bool ok) {
    // line: the 128 bytes [hi 32 | lo 32] of (this lane's row, this tile's 32 columns).  Every lane must take part in the
    // swaps (EXEC all ones); only the stores are predicated.
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
        uint32_t ha[2], la[2], hb[2], lb[2];  // a: columns 16 p2 + 4 fh + 0..3, b: columns 16 p2 + 8 + 4 fh + 0..3
        x3_split2(v[8 * p2], v[8 * p2 + 1], ha[0], la[0]);
        x3_split2(v[8 * p2 + 2], v[8 * p2 + 3], ha[1], la[1]);
        x3_split2(v[8 * p2 + 4], v[8 * p2 + 5], hb[0], lb[0]);
        x3_split2(v[8 * p2 + 6], v[8 * p2 + 7], hb[1], lb[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {  // fh = 0 keeps (own a, partner's a) = columns 16 p2 .. + 7, fh = 1 (partner's b, own b) = 16 p2 + 8 .. + 15
            const x3_u2 sh = __builtin_amdgcn_permlane32_swap(ha[e], hb[e], false, false);
            const x3_u2 sl = __builtin_amdgcn_permlane32_swap(la[e], lb[e], false, false);
            ha[e] = sh[0];
            hb[e] = sh[1];
            la[e] = sl[0];
            lb[e] = sl[1];
        }
        if (ok) {
            *reinterpret_cast<uint4*>(line + 32 * p2 + 16 * fh) = uint4{ha[0], ha[1], hb[0], hb[1]};
            *reinterpret_cast<uint4*>(line + 64 + 32 * p2 + 16 * fh) = uint4{la[0], la[1], lb[0], lb[1]};
        }
    }
}

// One K-step (32 k = two MFMA k-blocks) of a wave tile 32 TM x 32 TN on the stage at sb: fragments read from LDS as they
// stand, three products per k-block — the small cross terms first, the leading product last; the WEIGHT fragment is the
// first operand: acc[i][j] is the transposed tile (lane = token row, registers = output columns, see x3t_store_image).
// The order is laid down by hand: the fragments a product group needs are requested while the group BEFORE it runs, so
// the only LDS latency a wave sees is that of the first six reads behind the barrier.  (Left to hipcc the loop was
// [reads, s_waitcnt lgkmcnt(0), 4 - 8 MFMAs] six times per K-step — per-workgroup timeline of the 256 x 256 form at
// K = 384, benchmarks/x3_timeline.py: 25.9 us per K loop against 15.4 us of matrix-core time.)
typedef float f32x4s __attribute__((ext_vector_type(4)));
template <int SHAPE>
__device__ __forceinline__ void x3_mfma(const x3_h8& a, const x3_h8& b, f32x16& c, int p) {
    if (SHAPE == 0) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    } else {  // timing stand-in (DBG == 6): two 16x16x32 MFMAs on the same registers, results meaningless
        const int o = 8 * (p & 1);
        f32x4s c0 = {c[o], c[o + 1], c[o + 2], c[o + 3]}, c1 = {c[o + 4], c[o + 5], c[o + 6], c[o + 7]};
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c[o + e] = c0[e];
            c[o + 4 + e] = c1[e];
        }
    }
}

struct X3NoDma {
    __device__ __forceinline__ void operator()(int) const {}
};

// NP > 0: the wave's NP LDS-DMA instructions of the look-ahead stage are issued BETWEEN the MFMAs of the first k-block,
// one every (3 TM TN) / NP MFMAs, instead of as a burst behind the barrier: a global_load_lds costs its wave 60 - 180
// cycles of issue (address path), which is matrix-core time when both waves of a SIMD pay it at the same moment.
template <int TM, int TN, int SHAPE = 0, int NP = 0, class Dma = X3NoDma>
__device__ __forceinline__ void x3_kstep(const unsigned char* sb, const int (&a_off)[2][2], const int (&b_off)[2][2],
                                         f32x16 (&acc)[TM][TN], Dma dma = Dma()) {
    x3_h8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    auto read_first = [&](int ks) {  // what the first product (wh . al) needs
#pragma unroll
        for (int j = 0; j < TN; ++j) bh[ks][j] = *reinterpret_cast<const x3_h8*>(sb + b_off[ks][0] + j * 4096);
#pragma unroll
        for (int i = 0; i < TM; ++i) al[ks][i] = *reinterpret_cast<const x3_h8*>(sb + a_off[ks][1] + i * 4096);
    };
    auto read_rest = [&](int ks) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bl[ks][j] = *reinterpret_cast<const x3_h8*>(sb + b_off[ks][1] + j * 4096);
#pragma unroll
        for (int i = 0; i < TM; ++i) ah[ks][i] = *reinterpret_cast<const x3_h8*>(sb + a_off[ks][0] + i * 4096);
    };
    constexpr int SP = NP > 0 ? (3 * TM * TN / NP > 0 ? 3 * TM * TN / NP : 1) : 1;
    auto after = [&](int ks, int m) {  // m: MFMAs of this k-block issued so far
        if (NP > 0 && ks == 0 && m % SP == 0 && m / SP - 1 < NP) {
            __builtin_amdgcn_sched_barrier(0);
            dma(m / SP - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    read_first(0);
    read_rest(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#ifdef MVDB_X3_ONE_PRODUCT  // timing experiment (results = plain fp16 products): the two cross products left out
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                after(ks, i * TN + j + 1);
                after(ks, TM * TN + i * TN + j + 1);
            }
#else
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                x3_mfma<SHAPE>(bh[ks][j], al[ks][i], acc[i][j], 0);
                after(ks, i * TN + j + 1);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                x3_mfma<SHAPE>(bl[ks][j], ah[ks][i], acc[i][j], 1);
                after(ks, TM * TN + i * TN + j + 1);
            }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) read_first(1);  // under the leading product of k-block 0 ...
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                x3_mfma<SHAPE>(bh[ks][j], ah[ks][i], acc[i][j], 2);
                after(ks, 2 * TM * TN + i * TN + j + 1);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) read_rest(1);  // ... and under the first product of k-block 1 (at most 2 (TM + TN) fragments live)
        __builtin_amdgcn_sched_barrier(0);
    }
    if (NP > 0) {  // pieces the spacing did not reach (3 TM TN < NP)
#pragma unroll
        for (int p = 3 * TM * TN / SP; p < NP; ++p) dma(p);
    }
}

// The K-step as hipcc schedules it (reads and MFMAs of a k-block in source order, the compiler's own waits): what the
// one-tile-per-workgroup forms run — on their smaller wave tiles the hand-laid order below measured 2 - 3 % slower
// (e5-small forward, 256 x 32 tokens: 2.04 -> 2.08 ms).
template <int TM, int TN>
__device__ __forceinline__ void x3_kstep_plain(const unsigned char* sb, const int (&a_off)[2][2], const int (&b_off)[2][2],
                                               f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        x3_h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bh[j] = *reinterpret_cast<const x3_h8*>(sb + b_off[ks][0] + j * 4096);
            bl[j] = *reinterpret_cast<const x3_h8*>(sb + b_off[ks][1] + j * 4096);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            ah[i] = *reinterpret_cast<const x3_h8*>(sb + a_off[ks][0] + i * 4096);
            al[i] = *reinterpret_cast<const x3_h8*>(sb + a_off[ks][1] + i * 4096);
        }
#ifndef MVDB_X3_ONE_PRODUCT
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
    }
}

// Epilogue of a wave tile 32 TM x 32 TN on the transposed accumulator tiles: lane = token row, register r = column
// (r & 3) + 8 (r >> 2) + 4 fh of the tile.  Rows past T (last row band of a ragged batch) and column tiles past N
// (N % 32 == 0: whole tiles) only predicate the stores: one exec mask per tile, no branch per element.
template <int EPI, int TM, int TN>
__device__ __forceinline__ void x3_epilogue(const f32x16 (&acc)[TM][TN], int row0, int col0, int fr, int fh, int T, int N,
                                            float inv_wscale, const float* __restrict__ bias, const float* __restrict__ R,
                                            float* __restrict__ C, int qcols, float qscale) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = row0 + i * 32 + fr;
        const bool rok = row < T;
        const int64_t rbase = (int64_t)(rok ? row : T - 1) * N;  // clamped: loads stay in bounds
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cb = col0 + j * 32;  // first column of the tile
            const bool ok = rok && cb < N;
            const int cbc = cb < N ? cb : 0;
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (EPI != EPI_PARTIAL) b4 = *reinterpret_cast<const f32x4*>(bias + cbc + 8 * g + 4 * fh);
                f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
                if (EPI == EPI_BIAS_RESIDUAL) r4 = *reinterpret_cast<const f32x4*>(R + rbase + cbc + 8 * g + 4 * fh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][j][4 * g + e] * inv_wscale + b4[e];  // exact: the weight scale is a power of two
                    if (EPI == EPI_BIAS_GELU) x = 0.5f * x * (1.0f + x3_erf(x * 0.70710678118654752440f));
                    if (EPI == EPI_BIAS_RESIDUAL) x += r4[e];
                    if (EPI == EPI_BIAS_QKV) x = cb < qcols ? x * qscale : x;  // uniform per tile (qcols % 32 == 0)
                    v[4 * g + e] = x;
                }
            }
            if (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_QKV) {
                x3t_store_image(reinterpret_cast<unsigned char*>(C + rbase + cbc), fh, v, ok);
            } else if (ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(C + rbase + cbc + 8 * g + 4 * fh) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            }
        }
    }
}

// The planes of a split-K QKV / FFN1 GEMM (EPI_PARTIAL) -> the (hi | lo) image the fused epilogues write: sum in plane order,
// + bias, then EPI_BIAS_QKV's query scale (columns < qcols) resp. EPI_BIAS_GELU's erf-GELU (x3_epilogue's arithmetic).  One wave
// per (row, 256 columns): every load of the NP x 4 column groups in flight before the first add.  N % 64 == 0.
template <int EPI, int NP>
__global__ __launch_bounds__(256) void partials_image_kernel(const float* __restrict__ P, int64_t plane, const float* __restrict__ bias,
                                                             const int* __restrict__ seq_start, int B, int N, int qcols, float qscale,
                                                             float* __restrict__ Cimg) {
    const int T = seq_start[B];
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (p >= T) return;
    const int c0 = blockIdx.x * 256;
    float pv[NP][4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 64 < N ? c0 + u * 64 + lane : lane;   // (a group past N: re-reads the first, never stored)
        bv[u] = bias[c];
#pragma unroll
        for (int z = 0; z < NP; ++z) pv[z][u] = P[z * plane + (int64_t)p * N + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int cb = c0 + u * 64;          // wave-uniform
        float a = pv[0][u];
#pragma unroll
        for (int z = 1; z < NP; ++z) a += pv[z][u];
        a += bv[u];
        if (EPI == EPI_BIAS_GELU) a = 0.5f * a * (1.0f + x3_erf(a * 0.70710678118654752440f));
        if (EPI == EPI_BIAS_QKV) a = cb + (lane & 32) < qcols ? a * qscale : a;   // (qcols % 32 == 0: uniform per 32-column line)
        x3_pair_store(Cimg + (int64_t)p * N + cb + (lane & 32), lane & 31, a, cb < N);
    }
}

// Block tile 64 x 128 x 32, four waves 2 x 2.  BOTH operands arrive as (hi | lo) fp16 lines — A from the kernel that
// produced the activations (LayerNorm, attention, the GELU epilogue below: x3_pair_store), W from the one-time split of
// the weights — and go global -> LDS by global_load_lds into a three-stage ring (two K-steps in flight, ONE bare
// s_barrier per K-step, as gemm_f32_dma_kernel); the MFMA fragments are read from LDS as they stand: no VALU work in
// the K loop.  (First version: fp32 A rows, split in registers between LDS and the MFMA by every one of the N / 128
// workgroups that read them — 45 VALU per 12 MFMAs, 3 x the MFMA time in the compute-only ablation.)
//   stage = [A: BM rows x 128 B] [W: 128 rows x 128 B], a row's 128 bytes = 32 k of the hi plane | 32 k of the lo plane
//   ([row][K / 32][hi 32 | lo 32]: one full 128-byte line per row and K-step — with separate planes every request used
//   half a line and the L2 moved twice the W bytes: 4.9M line requests per QKV GEMM at T = 8192)
//   bank swizzle on the DMA source: slot p of row r holds the row's logical slot p ^ ((r >> 1) & 7)
// EPI_BIAS_GELU writes its output as (hi | lo) lines too (it only feeds the next GEMM); the other epilogues write fp32.
// dbg != 0: timing ablations (MVDB_GEMM_X3_DBG; results invalid): 1 = no fragment reads / MFMAs (the DMA ring,
// barriers and epilogue alone), 2 = no DMA (compute on whatever the LDS holds)
// The 256-row tile form pays off when its tiles make whole rounds of the CUs: at least one round, and either >= 4 rounds
// or a last round >= 85 % full.  Evaluated on the host with the padded token count (can the form apply at all?) and on the
// device with the packed one (does it?): a ragged batch of 256 x 128 token slots holds ~20k tokens, not 32k.
__host__ __device__ inline bool x3_big_form(int64_t T, int N, int bn, int cus) {
    const int64_t tiles = (int64_t)(N / bn) * ((T + 255) / 256);
    const int64_t rounds = (tiles + cus - 1) / cus;
    return tiles >= cus && (tiles >= 4 * (int64_t)cus || tiles * 100 >= rounds * cus * 85);
}

#ifdef MVDB_X3_ABLATE
// DBG == 5 (ablation build only): the real kernel plus a per-tile timeline — 16 words per tile:
// [tile id, HW_ID, XCC_ID, t start, t first stage landed, t K loop done, t stores issued, t stores acknowledged,
//  shader clock at "first stage landed", shader clock at "K loop done", ...], t = s_memrealtime (100 MHz, one clock for the
// whole chip), shader clock = s_memtime: (difference of the two) / (difference of t) x 100 MHz is the clock the chip held
// over the K loop.  Read back with mvdb_debug_x3_trace.  DBG == 6: the same with every v_mfma_f32_32x32x16_f16 of the K
// loop replaced by two v_mfma_f32_16x16x32_f16 on the same registers (same FLOPs, same LDS bytes, WRONG results): what the
// other MFMA shape would do to the K loop's time and clock, before anyone rewrites the epilogues for it.
constexpr int kX3TraceBlocks = 16384;
constexpr int kX3TraceWords = 16;
__device__ unsigned long long g_x3_trace[kX3TraceWords * kX3TraceBlocks];
__device__ __forceinline__ void x3_trace(int slot, bool first = false, int rec = -1) {
    if ((threadIdx.x & 63) != 0 || threadIdx.x >= 64) return;
    const unsigned b = rec >= 0 ? (unsigned)rec : blockIdx.y * gridDim.x + blockIdx.x;
    if (b >= (unsigned)kX3TraceBlocks) return;
    if (first) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_x3_trace[kX3TraceWords * b + 0] = b;
        g_x3_trace[kX3TraceWords * b + 1] = hw;
        g_x3_trace[kX3TraceWords * b + 2] = xcc;
    }
    g_x3_trace[kX3TraceWords * b + slot] = wall_clock64();
    if (slot == 4 || slot == 5) g_x3_trace[kX3TraceWords * b + slot + 4] = __builtin_readcyclecounter();
}
#else
__device__ __forceinline__ void x3_trace(int, bool = false, int = -1) {}
#endif

template <int EPI, int BM, int NST, int WAVES = 4, int BN = 128, int SPREAD = 0, int DBG = 0>
__global__ __launch_bounds__(WAVES * 64) void gemm_x3_dma_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ Wp,
                                                          float inv_wscale, const float* __restrict__ bias,
                                                          const float* __restrict__ R, float* __restrict__ C,
                                                          const int* __restrict__ Tptr, int N, int K, int sel_bn, int sel_cus,
                                                          int qcols, float qscale) {
    // DBG != 0 (builds with -DMVDB_X3_ABLATE + MVDB_GEMM_X3_DBG): timing ablations, results INVALID — 1: no fragment
    // reads / MFMAs, 2: no DMA, 3: no epilogue, 4: no K loop (epilogue only).  Compile-time: the same switches as a
    // kernel ARGUMENT cost the default path dearly (S = 512 FFN1 GEMM 553 -> 2,190 us: hipcc spilled around the branches)
    // sel_bn != 0: this launch is one of a PAIR — the 256-row form (sel_bn > 0) and its fallback (sel_bn < 0) — and the
    // number of packed tokens, known only on the device, decides which of the two does the work (x3_big_form)
    if (sel_bn != 0 && x3_big_form(*Tptr, N, sel_bn > 0 ? sel_bn : -sel_bn, sel_cus) != (sel_bn > 0)) return;
    constexpr int WM = WAVES / 2, TM = BM / (32 * WM), TN = BN / 64;  // WAVES / 2 x 2 waves, wave tile 32 TM x 32 TN
    constexpr int kA = BM * 128, kStage = kA + BN * 128;
    constexpr int NA = BM / 8;                  // DMA instructions of the A tile (8 rows each)
    constexpr int NI = (NA + BN / 8) / WAVES;   // per wave and stage (W: BN / 8 instructions of 8 rows each)
    static_assert((NA + BN / 8) % WAVES == 0, "whole DMA instructions per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    const int T = *Tptr;
    int bx, by;
    if (!xcd_tile(T, BM, bx, by)) return;
    const int m0 = by * BM, n0 = bx * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    if (DBG == 5) x3_trace(3, true);

    // DMA roles: instruction q of a stage; wave w issues q = w NI .. w NI + NI - 1 (the kind of q is wave-uniform)
    int64_t voff[NI];  // T K 4 bytes pass 4 GiB at 512 x 512 tokens of a 4096-wide FFN
    const char* sbase[NI];
    int kstep[NI], dsto[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = wave * NI + i;
        if (q < NA) {
            const int row = 8 * q + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            int g = m0 + row;
            g = g < T ? g : T - 1;  // rows past the edge: clamped (masked in the epilogue)
            sbase[i] = reinterpret_cast<const char*>(A);
            voff[i] = (int64_t)g * K * 4 + 16 * slot;
            kstep[i] = 128;
            dsto[i] = q * 1024;
        } else {
            const int qq = q - NA;  // 8 rows of [hi 64 B | lo 64 B] = one 128-byte line per row and K-step
            const int row = 8 * qq + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            int g = n0 + row;
            g = g < N ? g : N - 1;
            sbase[i] = reinterpret_cast<const char*>(Wp);
            voff[i] = (int64_t)g * K * 4 + 16 * slot;
            kstep[i] = 128;
            dsto[i] = kA + qq * 1024;
        }
    }
    // EPI_PARTIAL (split-K, grid z = the split): this workgroup sums K-steps [kt_base, kt_base + nk) and stores the bare partial
    // into plane z of C (planes `qcols` rows apart); ln_partials_kernel adds the planes, the bias and the residual
    const int nk_all = K / 32;
    int kt_base = 0, nk_mine = nk_all;
    if (EPI == EPI_PARTIAL) {
        const int per = (nk_all + (int)gridDim.z - 1) / (int)gridDim.z;
        kt_base = (int)blockIdx.z * per;
        nk_mine = nk_all - kt_base < per ? nk_all - kt_base : per;
        if (nk_mine < 0) nk_mine = 0;
        C += (int64_t)blockIdx.z * qcols * N;
    }
    auto issue_piece = [&](int kt, int stage, int i) {
        if (DBG == 2) return;
        const char* src = sbase[i] + (int64_t)(kt_base + kt) * kstep[i];
        __builtin_amdgcn_global_load_lds((enc_gbl_ptr)(src + voff[i]), (enc_lds_ptr)(xsm + stage * kStage + dsto[i]), 16, 0, 0);
    };
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int i = 0; i < NI; ++i) issue_piece(kt, stage, i);
    };
    // fragment byte offsets inside a stage, tile (0, 0) of the wave: row tile i / column tile j add 4096 i / 4096 j, folded into
    // the ds_read's immediate (the swizzle term depends on the row's low bits only)
    int a_off[2][2], b_off[2][2];
    const int ga = (fr >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            a_off[ks][pl] = (wm * (BM / WM) + fr) * 128 + (((4 * pl + 2 * ks + fh) ^ ga) << 4);
            b_off[ks][pl] = kA + (wn * (BN / 2) + fr) * 128 + (((4 * pl + 2 * ks + fh) ^ ga) << 4);
        }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = DBG == 4 ? 0 : nk_mine;
    // ring: NST - 1 stages in flight.  Stages past the end are issued too (clamped to the last K-step, into buffers
    // nobody reads) so that every counted wait sees a full ring.
    if (nk > 0) {
#pragma unroll
        for (int u = 0; u < NST - 1; ++u) issue(u < nk ? u : nk - 1, u);
    }
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NI) : "memory");  // this wave's part of stage kt has landed
        __builtin_amdgcn_s_barrier();  // every wave's part has; and every wave is done reading stage kt - 1
        __builtin_amdgcn_sched_barrier(0);
        if (DBG == 5 && kt == 0) x3_trace(4);
        const int ahead = kt + NST - 1, akt = ahead < nk ? ahead : nk - 1, abuf = st == 0 ? NST - 1 : st - 1;  // into the buffer of stage kt - 1
        if (!SPREAD || DBG == 1) issue(akt, abuf);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sb = xsm + st * kStage;
        if (DBG == 1) {
            acc[0][0][0] += (float)kt;
            st = st == NST - 1 ? 0 : st + 1;
            continue;
        }
        if (SPREAD)
            x3_kstep<TM, TN, 0, NI>(sb, a_off, b_off, acc, [&](int i) { issue_piece(akt, abuf, i); });
        else
            x3_kstep_plain<TM, TN>(sb, a_off, b_off, acc);
        st = st == NST - 1 ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped look-ahead DMAs must land before the LDS is released
    if (DBG == 3) {  // ablation: K loop only.  EVERY accumulator stays live through this never-taken store (with one
        float live = 0.f;  // of them hipcc drops the MFMAs of all the other tiles and the "K loop" costs 1 / (TM TN))
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) live += acc[i][j][r];
        if (live == 1.2345e-30f) C[0] = 0.f;
        return;
    }
    if (DBG == 5) x3_trace(5);
    x3_epilogue<EPI, TM, TN>(acc, m0 + wm * (BM / WM), n0 + wn * (BN / 2), fr, fh, T, N, inv_wscale, bias, R, C, qcols, qscale);
    if (DBG == 5) {
        __builtin_amdgcn_sched_barrier(0);
        x3_trace(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        x3_trace(7);
    }
}

// =================================================================================================
// The 256-row tile forms as ONE PERSISTENT workgroup per CU (eight waves 4 x 2, wave tile 64 x 32 TN, BN = 64 TN, two
// stages): workgroup w walks the tiles w, w + G, w + 2 G, ... of the XCD-aware tile list (xcd_tile's order, on the
// packed token count), and the K-step pipeline runs THROUGH the tile boundary — the look-ahead DMA of a tile's last
// K-step fetches the next tile's first stage, which lands under the epilogue.  Per-workgroup timeline of the
// one-tile-per-workgroup form at T = 131072, FFN1 (benchmarks/x3_timeline.py, 3072 workgroups, 12 per CU): 2.4 us launch ->
// first stage landed, 26.5 us K loop (51.2k cycles at 1.94 GHz for 36.9k cycles of MFMA issue), 9.9 us until wave 0 has
// issued its stores, and 4.6 us more until the next workgroup starts on the CU (the slowest wave's epilogue + the
// relaunch).  Persistent: the 2.4 us are gone, the 4.6 become the wait for the slowest wave at the next tile's first
// barrier; S = 512 forward 26.0 -> 25.7 ms.  With the DMA instructions spread between the MFMAs (x3_kstep): 47.6k cycles
// per K loop, at 1.87 GHz — the chip gives part of every saved cycle back as clock (25.7 -> 25.4 ms).  A 16x16x32 stand-in
// for the MFMAs (DBG = 6: same FLOPs, same LDS bytes) ran the K loop in 52.0k cycles at the same clock: not a lever here.
// =================================================================================================
template <int EPI, int BN, int SPREAD = 1, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_x3_big_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ Wp,
                                                          float inv_wscale, const float* __restrict__ bias,
                                                          const float* __restrict__ R, float* __restrict__ C,
                                                          const int* __restrict__ Tptr, int N, int K, int sel_bn, int sel_cus,
                                                          int qcols, float qscale) {
    if (sel_bn != 0 && x3_big_form(*Tptr, N, sel_bn > 0 ? sel_bn : -sel_bn, sel_cus) != (sel_bn > 0)) return;
    constexpr int BM = 256, WAVES = 8, WM = 4, TM = 2, TN = BN / 64;
    constexpr int kA = BM * 128, kStage = kA + BN * 128;
    constexpr int NA = BM / 8, NI = (NA + BN / 8) / WAVES;
    static_assert((NA + BN / 8) % WAVES == 0, "whole DMA instructions per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    const int T = *Tptr;
    const unsigned gx = (unsigned)(N / BN);
    const unsigned active = gx * (unsigned)((T + BM - 1) / BM), chunk = active >> 3;
    unsigned v = blockIdx.x;
    if (v >= active) return;
    auto tile_of = [&](unsigned vl, int& m0, int& n0) {
        unsigned lin = vl;
        if (lin < (chunk << 3)) lin = (lin & 7u) * chunk + (lin >> 3);
        m0 = (int)(lin / gx) * BM;
        n0 = (int)(lin % gx) * BN;
    };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    // DMA roles (gemm_x3_dma_kernel's): instruction q = wave NI + i of a stage moves 8 rows of A (q < NA) or of W.  Source =
    // a wave-uniform base of (tile, K-step) + a 32-bit lane offset inside the tile (the tile's rows are < 2 MiB apart).
    uint32_t lrow[NI], lslot[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = wave * NI + i;
        const int row = 8 * (q < NA ? q : q - NA) + (lane >> 3);
        lrow[i] = (uint32_t)row;
        lslot[i] = (uint32_t)(16 * ((lane & 7) ^ ((row >> 1) & 7)));
    }
    auto issue_piece = [&](int m0, int n0, int kt, int stage, int i) {
        if (DBG == 2) return;            // ablation: no DMA (the MFMAs run on whatever the LDS holds)
        if (DBG == 7) m0 = n0 = 0;       // ablation: every tile stages the SAME operand rows (always L2-resident)
        const int q = wave * NI + i;
        const char* base = q < NA ? reinterpret_cast<const char*>(A) + ((int64_t)m0 * K * 4 + (int64_t)kt * 128)
                                  : reinterpret_cast<const char*>(Wp) + ((int64_t)n0 * K * 4 + (int64_t)kt * 128);
        const uint32_t rlim = (uint32_t)(T - 1 - m0);  // rows past the edge: clamped (masked in the epilogue); N % BN == 0
        const uint32_t r = q < NA ? (lrow[i] < rlim ? lrow[i] : rlim) : lrow[i];
        const uint32_t off = r * (uint32_t)(K * 4) + lslot[i];
        __builtin_amdgcn_global_load_lds((enc_gbl_ptr)(base + off),
                                         (enc_lds_ptr)(xsm + stage * kStage + (q < NA ? q * 1024 : kA + (q - NA) * 1024)), 16, 0, 0);
    };
    auto issue = [&](int m0, int n0, int kt, int stage) {
#pragma unroll
        for (int i = 0; i < NI; ++i) issue_piece(m0, n0, kt, stage, i);
    };
    // fragment byte offsets inside a stage, tile (0, 0) of the wave: row tile i / column tile j add 4096 i / 4096 j, folded into
    // the ds_read's immediate (the swizzle term depends on the row's low bits only)
    int a_off[2][2], b_off[2][2];
    const int ga = (fr >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            a_off[ks][pl] = (wm * (BM / WM) + fr) * 128 + (((4 * pl + 2 * ks + fh) ^ ga) << 4);
            b_off[ks][pl] = kA + (wn * (BN / 2) + fr) * 128 + (((4 * pl + 2 * ks + fh) ^ ga) << 4);
        }

    const int nk = DBG == 4 ? 0 : K / 32;
    int m0, n0;
    tile_of(v, m0, n0);
    if (nk > 0) issue(m0, n0, 0, 0);
    int st = 0;
    for (;;) {
        if (DBG == 5 || DBG == 6) x3_trace(3, true, (int)v);
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const unsigned vn = v + gridDim.x;
        int m1 = m0, n1 = n0;
        if (vn < active) tile_of(vn, m1, n1);
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of stage kt has landed (and, at kt = 0, its stores of the tile before are acknowledged)
            __builtin_amdgcn_s_barrier();                       // every wave's part has; and every wave is done reading the other buffer
            __builtin_amdgcn_sched_barrier(0);
            if ((DBG == 5 || DBG == 6) && kt == 0) x3_trace(4, false, (int)v);
            // look-ahead: K-step kt + 1 of this tile, or the next tile's first stage (lands under this tile's epilogue);
            // issued piece by piece between the MFMAs (x3_kstep)
            // (the last tile's last K-step re-fetches its own first stage into the buffer nobody reads: no branch per piece)
            const bool more = kt + 1 < nk;
            const int lm = more ? m0 : m1, ln = more ? n0 : n1, lk = more ? kt + 1 : 0;
            auto dma = [&](int i) { issue_piece(lm, ln, lk, st ^ 1, i); };
            if (DBG == 1) {              // ablation: the DMAs and the barriers only (no fragment reads, no MFMAs)
#pragma unroll
                for (int i = 0; i < NI; ++i) dma(i);
                acc[0][0][0] += (float)kt;
                st ^= 1;
                continue;
            }
            if (SPREAD)
                x3_kstep<TM, TN, DBG == 6 ? 1 : 0, NI>(xsm + st * kStage, a_off, b_off, acc, dma);
            else {
#pragma unroll
                for (int i = 0; i < NI; ++i) dma(i);
                __builtin_amdgcn_sched_barrier(0);
                x3_kstep<TM, TN, DBG == 6 ? 1 : 0>(xsm + st * kStage, a_off, b_off, acc);
            }
            st ^= 1;
        }
        if (DBG == 5 || DBG == 6) x3_trace(5, false, (int)v);
        if (DBG == 3) {
            float live = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) live += acc[i][j][r];
            if (live == 1.2345e-30f) C[0] = 0.f;
        } else {
            x3_epilogue<EPI, TM, TN>(acc, m0 + wm * (BM / WM), n0 + wn * (BN / 2), fr, fh, T, N, inv_wscale, bias, R, C, qcols, qscale);
        }
        if (DBG == 5 || DBG == 6) {
            __builtin_amdgcn_sched_barrier(0);
            x3_trace(6, false, (int)v);
            x3_trace(7, false, (int)v);
        }
        if (vn >= active) break;
        v = vn;
        m0 = m1;
        n0 = n1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// =================================================================================================
// compute = 2, N == H GEMMs (attention output projection, FFN2): the split-precision GEMM with bias + residual +
// LayerNorm fused into the epilogue.  A workgroup owns WHOLE ROWS — tile BM x H, waves WM x 4, wave tile 32 TM x 32 TN,
// H = 128 TN — so the row statistics never leave the CU: the LayerNorm kernel that followed each of these GEMMs (two
// launches, an fp32 y written and read back, 11 % of the S = 32 forward) is gone.
//   v  = acc / wscale + bias + x        (x: the residual stream, read by the rows' owner only)
//   x' = (v - mean) * rstd * gamma + beta, two-pass statistics as ln_kernel / torch (mean first, then the squared
//        deviations).  The tiles are TRANSPOSED (x3t_store_image): a lane is a token row, so a row's sums are over the
//        lane's own TN x 16 registers, one v_permlane32_swap with the other lane half (the columns in between; the same
//        bits on both halves) and the four column waves through LDS (summed in wave order: deterministic)
//   x' is written IN PLACE over x (fp32, for the next residual and the pooling) and as the (hi | lo) fp16 image the next
//   GEMM's A operand is DMA'd from.
// K loop: gemm_x3_dma_kernel's (LDS-DMA ring, one bare s_barrier per K-step, fragments read as they stand).  The W tile
// is all H rows of the weight: BM = 32 keeps a T = 8192 batch on every CU (256 workgroups, ring of three stages =
// 156 KiB); BM = 64 / 128 (eight waves) serve longer batches with less W traffic per row.
// =================================================================================================
template <int BM, int WM, int TN, int NST, int SPREAD = 0, int DBG = 0>
__global__ __launch_bounds__(WM * 256) void gemm_x3_ln_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ Wp,
                                                             float inv_wscale, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, float* X, float* Xp, const int* __restrict__ Tptr, int K) {
    constexpr int WN = 4, WAVES = WM * WN, TM = BM / (32 * WM), BN = WN * TN * 32;
    constexpr int kA = BM * 128, kStage = kA + BN * 128;
    constexpr int NA = BM / 8;                  // DMA instructions of the A tile (8 rows each)
    constexpr int NI = (NA + BN / 8) / WAVES;   // per wave and stage
    static_assert((NA + BN / 8) % WAVES == 0, "whole DMA instructions per wave");
    // PERSISTENT (the two-stage forms, BM = 64 / 128): the grid is one workgroup per CU, workgroup w owns the row bands
    // w, w + G, ... and the K-step pipeline runs through the band boundary as in gemm_x3_big_kernel (the next band's first
    // stage lands under the LayerNorm epilogue; the row statistics have their own LDS behind the ring).  The three-stage
    // form (BM = 32: T = 8192) stays one band per workgroup: its counted waits would have to count the epilogue's loads.
    // (Tried on that form and dropped: four more waves that only issue the ring's DMAs, so that the compute waves — one per
    //  SIMD there — never stall in the vector-memory path: 1.91 ms per S = 32 forward either way.)
    constexpr bool PERS = NST == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    float* red = reinterpret_cast<float*>(xsm + NST * kStage);  // [2 passes][WN column waves][BM rows]
    const int T = *Tptr;
    int m0 = blockIdx.x * BM;
    if (m0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;

    // DMA roles: instruction q = wave NI + i of a stage moves 8 rows of A (q < NA) or of W; source = a wave-uniform base of
    // (band, K-step) + a 32-bit lane offset
    uint32_t lrow[NI], lslot[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = wave * NI + i;
        const int row = 8 * (q < NA ? q : q - NA) + (lane >> 3);
        lrow[i] = (uint32_t)row;
        lslot[i] = (uint32_t)(16 * ((lane & 7) ^ ((row >> 1) & 7)));
    }
    auto issue_piece = [&](int mb, int kt, int stage, int i) {
        if (DBG == 2) return;  // ablations as in gemm_x3_dma_kernel (results invalid)
        const int q = wave * NI + i;
        const char* base = q < NA ? reinterpret_cast<const char*>(A) + ((int64_t)mb * K * 4 + (int64_t)kt * 128)
                                  : reinterpret_cast<const char*>(Wp) + (int64_t)kt * 128;
        const uint32_t rlim = (uint32_t)(T - 1 - mb);  // rows past the edge: clamped (never stored); W: every row exists
        const uint32_t r = q < NA ? (lrow[i] < rlim ? lrow[i] : rlim) : lrow[i];
        __builtin_amdgcn_global_load_lds((enc_gbl_ptr)(base + (r * (uint32_t)(K * 4) + lslot[i])),
                                         (enc_lds_ptr)(xsm + stage * kStage + (q < NA ? q * 1024 : kA + (q - NA) * 1024)), 16, 0, 0);
    };
    auto issue = [&](int mb, int kt, int stage) {
#pragma unroll
        for (int i = 0; i < NI; ++i) issue_piece(mb, kt, stage, i);
    };
    // fragment byte offsets inside a stage, tile (0, 0) of the wave: row tile i / column tile j add 4096 i / 4096 j, folded into
    // the ds_read's immediate (the swizzle term depends on the row's low bits only)
    int a_off[2][2], b_off[2][2];
    const int ga = (fr >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            a_off[ks][pl] = (wm * (TM * 32) + fr) * 128 + (((4 * pl + 2 * ks + fh) ^ ga) << 4);
            b_off[ks][pl] = kA + (wn * (TN * 32) + fr) * 128 + (((4 * pl + 2 * ks + fh) ^ ga) << 4);
        }

    const int nk = DBG == 4 ? 0 : K / 32;
    const int mstep = PERS ? (int)gridDim.x * BM : 0;
    if (nk > 0) {
#pragma unroll
        for (int u = 0; u < NST - 1; ++u) issue(m0, u < nk ? u : nk - 1, u);
    }
    int st = 0;
    for (;;) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int m1 = m0 + mstep;                 // the next band of this workgroup
        const bool more_bands = PERS && m1 < T;
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NI) : "memory");  // this wave's part of stage kt has landed
            __builtin_amdgcn_s_barrier();  // every wave's part has; and every wave is done reading stage kt - 1
            __builtin_amdgcn_sched_barrier(0);
            // look-ahead into the buffer of stage kt - 1: K-step kt + NST - 1 of this band (clamped past the end: every counted
            // wait sees a full ring) or, persistent, the next band's first stage
            const int ahead = kt + NST - 1;
            const bool cross = PERS && ahead >= nk && more_bands;
            const int amb = cross ? m1 : m0, akt = cross ? 0 : (ahead < nk ? ahead : nk - 1), abuf = st == 0 ? NST - 1 : st - 1;
            if (!SPREAD || DBG == 1) issue(amb, akt, abuf);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* sb = xsm + st * kStage;
            if (DBG == 1) {
                acc[0][0][0] += (float)kt;
                st = st == NST - 1 ? 0 : st + 1;
                continue;
            }
            if (SPREAD)
                x3_kstep<TM, TN, 0, NI>(sb, a_off, b_off, acc, [&](int i) { issue_piece(amb, akt, abuf, i); });
            else
                x3_kstep_plain<TM, TN>(sb, a_off, b_off, acc);
            st = st == NST - 1 ? 0 : st + 1;
        }
        // ---- epilogue: v = acc / wscale + bias + residual, LayerNorm over the row, in place ------------------------------
        // transposed tiles (x3t_store_image): lane = token row, so a row's statistics are sums over the lane's own TN x 16
        // registers, one exchange with the other lane half (its columns in between) and the four column waves through LDS
        if (DBG == 3) {
            float live = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) live += acc[i][j][r];
            if (live == 1.2345e-30f) X[0] = 0.f;
            if (!more_bands) break;
            m0 = m1;
            continue;
        }
        int64_t rbase[TM];
        bool rok[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * (TM * 32) + i * 32 + fr;
            rok[i] = row < T;
            rbase[i] = (int64_t)(rok[i] ? row : T - 1) * BN;  // clamped: loads stay in bounds, stores are predicated
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c4 = wn * (TN * 32) + j * 32 + 8 * g + 4 * fh;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + c4);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const f32x4 r4 = *reinterpret_cast<const f32x4*>(X + rbase[i] + c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j][4 * g + e] = (acc[i][j][4 * g + e] * inv_wscale + b4[e]) + r4[e];  // exact scale: power of two
                }
            }
        float rstd[TM];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {  // 0: mean (then acc -= mean), 1: variance of the centred values
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) p += pass == 0 ? acc[i][j][r] : acc[i][j][r] * acc[i][j][r];
                // + the other lane half's columns: (lower, lower) + (upper, upper) on both halves, the same bits on both
                const x3_u2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
                p = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                if (fh == 0) red[(pass * WN + wn) * BM + wm * (TM * 32) + i * 32 + fr] = p;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the partial sums are in LDS
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) tot += red[(pass * WN + w) * BM + wm * (TM * 32) + i * 32 + fr];  // wave order: deterministic
                if (pass == 0) {
                    const float mean = tot / (float)BN;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] -= mean;
                } else {
                    rstd[i] = 1.0f / sqrtf(tot / (float)BN + eps);  // biased variance, eps inside the sqrt
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 g4[4], t4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c4 = wn * (TN * 32) + j * 32 + 8 * g + 4 * fh;
                g4[g] = *reinterpret_cast<const f32x4*>(gamma + c4);
                t4[g] = *reinterpret_cast<const f32x4*>(beta + c4);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float v[16];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * g + e] = acc[i][j][4 * g + e] * rstd[i] * g4[g][e] + t4[g][e];
                const int cb = wn * (TN * 32) + j * 32;
                if (rok[i]) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(X + rbase[i] + cb + 8 * g + 4 * fh) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                }
                x3t_store_image(reinterpret_cast<unsigned char*>(Xp + rbase[i] + cb), fh, v, rok[i]);
            }
        }
        if (!more_bands) break;
        m0 = m1;
        // (the statistics scratch is reused by the next band only after its K loop: a dozen barriers from here)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped look-ahead DMAs must land before the LDS is released
}

// =================================================================================================
// attention: one thread per query row, K/V tiles broadcast from LDS, chunked online softmax (fp32)
// =================================================================================================
constexpr int ATT_Q = 128;  // queries per block
constexpr int ATT_KT = 64;  // keys per LDS tile
constexpr int ATT_CH = 8;   // keys per softmax chunk

template <int HD>
__global__ __launch_bounds__(ATT_Q) void attention_kernel(const float* __restrict__ qkv,
                                                          const int* __restrict__ seq_start, int H,
                                                          float scale, float* __restrict__ ctx) {
    const int b = blockIdx.z, h = blockIdx.y;
    const int s0 = seq_start[b], len = seq_start[b + 1] - s0;
    const int q0 = blockIdx.x * ATT_Q;
    if (q0 >= len) return;
    __shared__ float Ks[ATT_KT * HD];
    __shared__ float Vs[ATT_KT * HD];
    const int qi = q0 + threadIdx.x;
    const bool active = qi < len;
    const int64_t ld = 3 * (int64_t)H;
    float q[HD], o[HD];
    {
        const float* qp = qkv + (int64_t)(s0 + (active ? qi : 0)) * ld + h * HD;
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qp + c);
            q[c] = v.x; q[c + 1] = v.y; q[c + 2] = v.z; q[c + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int kt = 0; kt < len; kt += ATT_KT) {
        const int nk = min(ATT_KT, len - kt);
        __syncthreads();
        for (int i = threadIdx.x; i < ATT_KT * (HD / 4); i += ATT_Q) {
            const int j = i / (HD / 4), c = (i % (HD / 4)) * 4;
            f32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
            if (j < nk) {
                const float* base = qkv + (int64_t)(s0 + kt + j) * ld + h * HD + c;
                kv = *reinterpret_cast<const f32x4*>(base + H);
                vv = *reinterpret_cast<const f32x4*>(base + 2 * H);
            }
            *reinterpret_cast<f32x4*>(&Ks[j * HD + c]) = kv;
            *reinterpret_cast<f32x4*>(&Vs[j * HD + c]) = vv;
        }
        __syncthreads();
        for (int j0 = 0; j0 < nk; j0 += ATT_CH) {
            float s[ATT_CH];
            float cmax = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < ATT_CH; ++jj) {
                const float* kr = &Ks[(j0 + jj) * HD];
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < HD; ++c) acc = fmaf(q[c], kr[c], acc);
                s[jj] = (j0 + jj < nk) ? acc * scale : -INFINITY;
                cmax = fmaxf(cmax, s[jj]);
            }
            const float mn = fmaxf(m, cmax);
            const float alpha = expf(m - mn);  // m = -inf on the first chunk -> 0
            l *= alpha;
#pragma unroll
            for (int c = 0; c < HD; ++c) o[c] *= alpha;
#pragma unroll
            for (int jj = 0; jj < ATT_CH; ++jj) {
                const float p = expf(s[jj] - mn);  // masked tail: exp(-inf) = 0
                l += p;
                const float* vr = &Vs[(j0 + jj) * HD];
#pragma unroll
                for (int c = 0; c < HD; ++c) o[c] = fmaf(p, vr[c], o[c]);
            }
            m = mn;
        }
    }
    if (active) {
        const float inv = 1.0f / l;
        float* op = ctx + (int64_t)(s0 + qi) * H + h * HD;
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
            f32x4 v = {o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv};
            *reinterpret_cast<f32x4*>(op + c) = v;
        }
    }
}

// =================================================================================================
// attention on the exact-fp32 matrix cores (default): "swapped" formulation, softmax in registers
// =================================================================================================
// One wave owns 32 query rows of one (sequence, head); the 4 waves of a block share K/V tiles in LDS.
//   S^T[key][query] = K · Q^T        v_mfma_f32_32x32x2_f32, A = K tile (k-major in LDS), B = Q^T (registers,
//                                    pre-scaled by 1/sqrt(hd)): the QUERY ends up on the lane (col = lane&31),
//                                    16 of the block's 32 keys on the registers of each lane half
//   softmax over keys                per lane: 16-register max/sum + one exchange with lane^32; online rescale
//   O^T[d][query] += V^T · P^T       A = V^T (V tile row-major in LDS: lane reads V[key][d = lane&31]),
//                                    B = P^T = accumulator register t AS IT STANDS: MFMA step t contracts the
//                                    two keys (t&3) + 8(t>>2) + {0,4} that the two lane halves hold in register t
// No P tile ever goes through LDS and no lane shuffles for the second product.
constexpr int AM_KT = 64;  // keys per LDS tile (2 MFMA key blocks)
// Scores are kept in the log2 domain (Q pre-scaled by log2(e)/sqrt(hd)) so that the softmax exponentials are
// single v_exp_f32 instructions: softmax(s) = 2^(s*log2e - max) / sum — the same function, ~1e-6 relative
// rounding difference from a libm expf (the precise expf expanded to ~15 instructions per element and cost
// as much as the MFMAs).
constexpr float kLog2e = 1.4426950408889634f;

template <int HD>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* __restrict__ qkv,
                                                             const int* __restrict__ seq_start, int H,
                                                             float scale, float* __restrict__ ctx) {
    constexpr int DT = HD / 32;         // 32-wide d tiles of the output
    constexpr int KLD = AM_KT + 1;      // Kt row stride (floats): +1 => the transposing writes are <= 2-way
    const int b = blockIdx.z, h = blockIdx.y;
    const int s0 = seq_start[b], len = seq_start[b + 1] - s0;
    const int q0 = blockIdx.x * 128;
    if (q0 >= len) return;
    __shared__ float Kt[HD * KLD];      // [k][key]
    __shared__ float Vs[AM_KT * HD];    // [key][d]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int64_t ld = 3 * (int64_t)H;

    // Q^T fragments: step kk (k = 2kk + fh) of query q0 + 32*wave + fr, pre-scaled
    const int qrow = q0 + wave * 32 + fr;
    const bool qvalid = qrow < len;
    float qf[HD / 2];
    {
        const float* qp = qkv + (int64_t)(s0 + (qvalid ? qrow : 0)) * ld + h * HD;
#pragma unroll
        for (int kk = 0; kk < HD / 2; ++kk) qf[kk] = qvalid ? qp[2 * kk + fh] * (scale * kLog2e) : 0.f;
    }
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const bool wave_active = q0 + wave * 32 < len;  // wave-uniform

    for (int kt = 0; kt < len; kt += AM_KT) {
        const int nk = min(AM_KT, len - kt);
        __syncthreads();
        // cooperative tile load: thread -> (key = e / (HD/4), float4 chunk c); K transposed, V straight
        for (int e = tid; e < AM_KT * (HD / 4); e += 256) {
            const int key = e / (HD / 4), c = (e % (HD / 4)) * 4;
            f32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
            if (key < nk) {
                const float* base = qkv + (int64_t)(s0 + kt + key) * ld + h * HD + c;
                kv = *reinterpret_cast<const f32x4*>(base + H);
                vv = *reinterpret_cast<const f32x4*>(base + 2 * H);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) Kt[(c + j) * KLD + key] = kv[j];
            *reinterpret_cast<f32x4*>(&Vs[key * HD + c]) = vv;
        }
        __syncthreads();
        if (!wave_active) continue;
#pragma unroll
        for (int kb = 0; kb < AM_KT / 32; ++kb) {
            if (kb * 32 >= nk) break;
            // ---- S^T block: 32 keys x 32 queries ----------------------------------------------------
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < HD / 2; ++kk) {
                const float ka = Kt[(2 * kk + fh) * KLD + kb * 32 + fr];  // A[i = key fr][k = fh]
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(ka, qf[kk], st, 0, 0, 0);
            }
            // ---- online softmax, query on the lane --------------------------------------------------
            float cmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                st[r] = key < nk ? st[r] : -INFINITY;
                cmax = fmaxf(cmax, st[r]);
            }
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
            const float mn = fmaxf(m, cmax);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] - mn);  // masked keys: 2^(-inf) = 0
                psum += st[r];
            }
            psum += __shfl_xor(psum, 32);
            l = l * alpha + psum;
            m = mn;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
            // ---- O^T += V^T · P^T ----------------------------------------------------------------------
#pragma unroll
            for (int t16 = 0; t16 < 16; ++t16) {
                const int key = kb * 32 + (t16 & 3) + 8 * (t16 >> 2) + 4 * fh;
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const float va = Vs[key * HD + t * 32 + fr];  // A[i = d fr][k = fh] = V[key][d]
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, st[t16], o[t], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_active) return;
    // O^T tile t: col = query fr (lane), row d = t*32 + (r&3) + 8(r>>2) + 4fh
    if (qvalid) {
        const float inv = 1.0f / l;
        float* op = ctx + (int64_t)(s0 + qrow) * H + h * HD;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v = {o[t][4 * r4] * inv, o[t][4 * r4 + 1] * inv, o[t][4 * r4 + 2] * inv,
                           o[t][4 * r4 + 3] * inv};
                *reinterpret_cast<f32x4*>(op + t * 32 + 8 * r4 + 4 * fh) = v;  // d = 8*r4 + 4fh + 0..3
            }
    }
}

// concat Wq,Wk,Wv ([H,H] each) and their biases into one [3H,H] / [3H]
// =================================================================================================
// compute = 2 attention: the same two products as attention_mfma_kernel on the fp16 matrix cores, split-precision
// (x = xh + xl in fp16, three v_mfma_f32_32x32x16_f16 per 16-deep step, fp32 accumulate: 2^-22 per operand).  With the
// GEMMs on the 16-bit cores the exact-fp32 attention had become the largest kernel of the S = 512 forward (28 %).
//   * K and V tiles are split ONCE per workgroup while they are staged: K as [key][d] (hi, lo) fp16 planes (rows padded
//     to 2 HD + 16 bytes: conflict-free b128 fragment reads), V TRANSPOSED as [d][key] planes (rows of 136 bytes);
//   * S^T = K Q^T: A = a K fragment (key on the lane row, 8 consecutive d), B = the Q fragment kept in registers (split
//     once, pre-scaled by log2(e) / sqrt(hd));
//   * softmax on the lane as before (the query is the lane's column);
//   * O^T += V^T P^T: the B operand of k-block kb is registers 8 kb .. 8 kb + 7 of the S^T accumulator AS THEY STAND,
//     split in place — register j of lane half fh holds key (j & 3) + 8 (j >> 2) + 4 fh of the 16-key block, and the V^T
//     fragment is gathered in that same key order (two 8-byte reads of four consecutive keys each), so no shuffle and no
//     P tile through LDS.
// =================================================================================================
constexpr int AX_KT = 64;  // keys per LDS tile

// WV = waves per workgroup (32 queries each): 4, or 1 for short sequences (S <= 32: one workgroup per (sentence, head) with
// no idle waves and a 32-key tile, twice as many resident)
template <int HD, int WV = 4>
__global__ __launch_bounds__(WV * 64) void attention_x3_kernel(const float* __restrict__ qkv,
                                                           const int* __restrict__ seq_start, int H, float scale,
                                                           float* __restrict__ ctx) {
    constexpr int DT = HD / 32, KB = HD / 16;
    constexpr int KP = HD * 2 + 16;     // K plane row pitch (bytes)
    constexpr int VP = (WV == 1 ? 32 : AX_KT) * 2 + 8;   // V^T plane row pitch (bytes)
    const int b = blockIdx.z, h = blockIdx.y;
    const int s0 = seq_start[b], len = seq_start[b + 1] - s0;
    constexpr int KT = WV == 1 ? 32 : AX_KT;  // keys per LDS tile
    const int q0 = blockIdx.x * (32 * WV);
    if (q0 >= len) return;
    __shared__ __attribute__((aligned(16))) unsigned char Kh[(WV == 1 ? 32 : AX_KT) * KP];
    __shared__ __attribute__((aligned(16))) unsigned char Kl[(WV == 1 ? 32 : AX_KT) * KP];
    __shared__ __attribute__((aligned(16))) unsigned char Vh[HD * VP];
    __shared__ __attribute__((aligned(16))) unsigned char Vl[HD * VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int64_t ld = 3 * (int64_t)H;

    // Q fragments (B operand): query q0 + 32 wave + fr, k = 16 kb + 8 fh .. + 7, pre-scaled, split once
    const int qrow = q0 + wave * 32 + fr;
    const bool qvalid = qrow < len;
    union Op {
        x3_h8 v;
        uint32_t w[4];
    };
    Op qh[KB], ql[KB];
    {
        const float* qp = qkv + (int64_t)(s0 + (qvalid ? qrow : 0)) * ld + h * HD;
        const float sc = scale * kLog2e;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(qp + 16 * kb + 8 * fh);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(qp + 16 * kb + 8 * fh + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p0 = (e < 2 ? a0[2 * e] : a1[2 * (e - 2)]) * sc, p1 = (e < 2 ? a0[2 * e + 1] : a1[2 * (e - 2) + 1]) * sc;
                x3_split2(qvalid ? p0 : 0.f, qvalid ? p1 : 0.f, qh[kb].w[e], ql[kb].w[e]);
            }
        }
    }
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const bool wave_active = q0 + wave * 32 < len;  // wave-uniform

    for (int kt = 0; kt < len; kt += KT) {
        const int nk = min(KT, len - kt);
        __syncthreads();
        // cooperative tile load + split: thread -> (key, 4 consecutive d)
        for (int e = tid; e < KT * (HD / 4); e += WV * 64) {
            const int key = e / (HD / 4), c = (e % (HD / 4)) * 4;
            f32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
            if (key < nk) {
                const float* base = qkv + (int64_t)(s0 + kt + key) * ld + h * HD + c;
                kv = *reinterpret_cast<const f32x4*>(base + H);
                vv = *reinterpret_cast<const f32x4*>(base + 2 * H);
            }
            uint32_t h0, l0, h1, l1;
            x3_split2(kv[0], kv[1], h0, l0);
            x3_split2(kv[2], kv[3], h1, l1);
            *reinterpret_cast<uint2*>(Kh + key * KP + c * 2) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(Kl + key * KP + c * 2) = uint2{l0, l1};
            x3_split2(vv[0], vv[1], h0, l0);
            x3_split2(vv[2], vv[3], h1, l1);
            *reinterpret_cast<uint16_t*>(Vh + (c + 0) * VP + key * 2) = (uint16_t)h0;
            *reinterpret_cast<uint16_t*>(Vh + (c + 1) * VP + key * 2) = (uint16_t)(h0 >> 16);
            *reinterpret_cast<uint16_t*>(Vh + (c + 2) * VP + key * 2) = (uint16_t)h1;
            *reinterpret_cast<uint16_t*>(Vh + (c + 3) * VP + key * 2) = (uint16_t)(h1 >> 16);
            *reinterpret_cast<uint16_t*>(Vl + (c + 0) * VP + key * 2) = (uint16_t)l0;
            *reinterpret_cast<uint16_t*>(Vl + (c + 1) * VP + key * 2) = (uint16_t)(l0 >> 16);
            *reinterpret_cast<uint16_t*>(Vl + (c + 2) * VP + key * 2) = (uint16_t)l1;
            *reinterpret_cast<uint16_t*>(Vl + (c + 3) * VP + key * 2) = (uint16_t)(l1 >> 16);
        }
        __syncthreads();
        if (!wave_active) continue;
#pragma unroll
        for (int kb32 = 0; kb32 < KT / 32; ++kb32) {
            if (kb32 * 32 >= nk) break;
            // ---- S^T block: 32 keys x 32 queries ----------------------------------------------------
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int off = (kb32 * 32 + fr) * KP + (16 * kb + 8 * fh) * 2;
                const x3_h8 kh = *reinterpret_cast<const x3_h8*>(Kh + off);
                const x3_h8 kl = *reinterpret_cast<const x3_h8*>(Kl + off);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[kb].v, st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[kb].v, st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[kb].v, st, 0, 0, 0);
            }
            // ---- online softmax, query on the lane --------------------------------------------------
            float cmax = -INFINITY;
            if (kb32 * 32 + 32 > nk) {  // only the last block of a sequence has keys to mask (workgroup-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb32 * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    st[r] = key < nk ? st[r] : -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) cmax = fmaxf(cmax, st[r]);
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
            const float mn = fmaxf(m, cmax);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] - mn);  // masked keys: 2^(-inf) = 0
                psum += st[r];
            }
            psum += __shfl_xor(psum, 32);
            l = l * alpha + psum;
            m = mn;
            if (__ballot(alpha != 1.0f) != 0ull) {  // once the running maxima have settled no lane rescales
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
            }
            // ---- O^T += V^T · P^T: k-block kb16 = accumulator registers 8 kb16 .. + 7, i.e. keys
            //      16 kb16 + {4 fh .. 4 fh + 3} and 16 kb16 + 8 + {4 fh .. 4 fh + 3} of this 32-key block
#pragma unroll
            for (int kb16 = 0; kb16 < 2; ++kb16) {
                Op ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) x3_split2(st[8 * kb16 + 2 * e], st[8 * kb16 + 2 * e + 1], ph.w[e], pl.w[e]);
                const int kbase = (kb32 * 32 + 16 * kb16 + 4 * fh) * 2;
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const int row = (t * 32 + fr) * VP + kbase;
                    union {
                        x3_h8 v;
                        uint2 u[2];
                    } vh, vl;
                    vh.u[0] = *reinterpret_cast<const uint2*>(Vh + row);
                    vh.u[1] = *reinterpret_cast<const uint2*>(Vh + row + 16);
                    vl.u[0] = *reinterpret_cast<const uint2*>(Vl + row);
                    vl.u[1] = *reinterpret_cast<const uint2*>(Vl + row + 16);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl.v, ph.v, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh.v, pl.v, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh.v, ph.v, o[t], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_active) return;
    // O^T tile t: col = query fr (lane), row d = t*32 + (r&3) + 8(r>>2) + 4fh
    if (qvalid) {
        const float inv = 1.0f / l;
        // the context only feeds the output projection: written as the (hi | lo) fp16 lines gemm_x3_dma_kernel reads
        // (h HD + 32 t is a multiple of 32: 32-column group t of the head = one 128-byte line of the image).  A lane holds
        // d = 8 r4 + 4 fh + 0 .. 3; the two lane halves trade their r4 = 2p + 1 / 2p pieces (v_permlane32_swap) so that
        // fh = 0 ends up with d = 16 p .. 16 p + 7 and fh = 1 with 16 p + 8 .. 16 p + 15: every store instruction then
        // writes whole 32-byte sectors (16 bytes per lane, the two halves adjacent), as the fp32 form did — stores of
        // half sectors cost 2.4 us per layer at T = 8192.
        unsigned char* op = reinterpret_cast<unsigned char*>(ctx + (int64_t)(s0 + qrow) * H + h * HD);
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                uint32_t ha[2], la[2], hb[2], lb[2];  // a: r4 = 2 p2, b: r4 = 2 p2 + 1
                x3_split2(o[t][8 * p2] * inv, o[t][8 * p2 + 1] * inv, ha[0], la[0]);
                x3_split2(o[t][8 * p2 + 2] * inv, o[t][8 * p2 + 3] * inv, ha[1], la[1]);
                x3_split2(o[t][8 * p2 + 4] * inv, o[t][8 * p2 + 5] * inv, hb[0], lb[0]);
                x3_split2(o[t][8 * p2 + 6] * inv, o[t][8 * p2 + 7] * inv, hb[1], lb[1]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {  // a.hi lanes <-> b.lo lanes: fh = 0 keeps (own a, partner's a), fh = 1 (partner's b, own b)
                    const x3_u2 sh = __builtin_amdgcn_permlane32_swap(ha[e], hb[e], false, false);
                    const x3_u2 sl = __builtin_amdgcn_permlane32_swap(la[e], lb[e], false, false);
                    ha[e] = sh[0];
                    hb[e] = sh[1];
                    la[e] = sl[0];
                    lb[e] = sl[1];
                }
                *reinterpret_cast<uint4*>(op + t * 128 + 32 * p2 + 16 * fh) = uint4{ha[0], ha[1], hb[0], hb[1]};
                *reinterpret_cast<uint4*>(op + t * 128 + 64 + 32 * p2 + 16 * fh) = uint4{la[0], la[1], lb[0], lb[1]};
            }
    }
}

// =================================================================================================
// compute = 2 attention on (hi | lo) IMAGES of Q, K, V (the QKV GEMM's EPI_BIAS_QKV epilogue writes them, Q already
// scaled by log2(e) / sqrt(hd)): same products, same softmax, same output as attention_x3_kernel, but no operand is split
// or transposed by this kernel —
//   * a head's 32 dimensions of one token are exactly one 128-byte line [hi 32 | lo 32] of the image (two lines at
//     hd = 64), so K and V tiles go global -> LDS by LDS-DMA as they stand (two stages: the next tile is in flight under
//     the current tile's MFMAs; ONE bare s_barrier per tile) and the Q fragments are plain 16-byte loads;
//   * the K fragment of S^T = K Q^T is a b128 read of a line (bank swizzle on the DMA source: 16-byte slot p of key r
//     holds the line's slot p ^ ((r >> 1) & 7); hd = 64: p ^ (r & 15) over the two lines);
//   * the V^T fragment of O^T += V^T P^T is read from the ROW-major V tile with ds_read_b64_tr_b16 (four keys x sixteen
//     dimensions per sixteen lanes, delivered transposed): keys 4 fh .. + 3 and 8 + 4 fh .. + 3 of the 16-key block, the
//     order in which the S^T accumulator holds P.  V lines are stored with their 64-byte halves exchanged on keys with
//     (key >> 1) & 1 (hd = 64: the four 64-byte quarters rotated by key & 3): the four keys of a read then sit on four
//     different bank groups.
// What is left on the VALU: the softmax and the (hi, lo) split of P.
// =================================================================================================
typedef short x3_s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) x3_s4 x3_lds_s4;

template <int HD, int WV>
__global__ __launch_bounds__(WV * 64) void attention_x3i_kernel(const float* __restrict__ qkv, const int* __restrict__ seq_start,
                                                               int H, float* __restrict__ ctx) {
    constexpr int DT = HD / 32, KB = HD / 16, LPK = HD / 32;  // lines per key
    constexpr int KT = WV == 1 ? 32 : 64;                     // keys per tile
    constexpr int kTile = KT * LPK * 128;                     // bytes of a K (or V) tile
    constexpr int NQ = 2 * KT * LPK / 8;                      // DMA instructions per tile (8 lines each): K then V
    constexpr int NI = NQ / WV;
    static_assert(NQ % WV == 0, "whole DMA instructions per wave");
    const int b = blockIdx.z, h = blockIdx.y;
    const int s0 = seq_start[b], len = seq_start[b + 1] - s0;
    const int q0 = blockIdx.x * (32 * WV);
    if (q0 >= len) return;
    __shared__ __attribute__((aligned(16))) unsigned char kv[2 * 2 * kTile];  // [stage][K | V][key][line][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int64_t pitch = 3 * (int64_t)H * 4;  // bytes per token row of the image
    const char* base = reinterpret_cast<const char*>(qkv) + (int64_t)s0 * pitch;

    // Q fragments (B operand): query q0 + 32 wave + fr, d = 16 kb + 8 fh .. + 7: hi at byte 2 d of its line, lo 64 further
    const int qrow = q0 + wave * 32 + fr;
    const bool qvalid = qrow < len;
    x3_h8 qh[KB], ql[KB];
    {
        const char* qp = base + (int64_t)(qvalid ? qrow : 0) * pitch + (int64_t)h * HD * 4;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int off = (kb >> 1) * 128 + (2 * (kb & 1) + fh) * 16;
            qh[kb] = *reinterpret_cast<const x3_h8*>(qp + off);
            ql[kb] = *reinterpret_cast<const x3_h8*>(qp + off + 64);
            if (!qvalid) {
#pragma unroll
                for (int e = 0; e < 8; ++e) qh[kb][e] = ql[kb][e] = (_Float16)0.f;
            }
        }
    }
    // DMA roles: instruction q of a tile moves lines 8 q' .. 8 q' + 7 of K (q < NQ / 2) or V; a lane's source is
    // (key row, line of the head, logical slot) for the physical slot it fills
    int dkey[NI], dsrc[NI], ddst[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = wave * NI + i;
        const bool isv = q >= NQ / 2;
        const int L = 8 * (isv ? q - NQ / 2 : q) + (lane >> 3);  // line of the tile
        const int key = L / LPK, sub = L % LPK;
        const int phys = sub * 8 + (lane & 7);                    // slot inside the key's SLOTS
        const int sw = isv ? (LPK == 1 ? ((key >> 1) & 1) << 2 : (key & 3) << 2) : (LPK == 1 ? (key >> 1) & 7 : key & 15);
        const int logical = phys ^ sw;
        dkey[i] = key;
        dsrc[i] = (isv ? 2 * H * 4 : H * 4) + h * HD * 4 + logical * 16;
        ddst[i] = (isv ? kTile : 0) + (isv ? q - NQ / 2 : q) * 1024;
    }
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int key = kt + dkey[i];
            key = key < len ? key : len - 1;  // past the sequence: any valid row (its scores are masked)
            __builtin_amdgcn_global_load_lds((enc_gbl_ptr)(base + (int64_t)key * pitch + dsrc[i]),
                                             (enc_lds_ptr)(kv + stage * 2 * kTile + ddst[i]), 16, 0, 0);
        }
    };
    // fragment addresses inside a stage.  K (A operand of S^T): key = 32 kb32 + fr, d = 16 kb + 8 fh
    int k_off[KB][2];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const int logical = (kb >> 1) * 8 + 4 * pl + 2 * (kb & 1) + fh;
            const int sw = LPK == 1 ? (fr >> 1) & 7 : fr & 15;  // key = 32 kb32 + fr: same low bits
            k_off[kb][pl] = fr * (LPK * 128) + ((logical ^ sw) << 4);
        }
    // V^T (A operand of O^T += V^T P^T) by transposed reads: the 16 lanes (g = lane >> 4) of a group point at keys
    // kbase + 4 (g >> 1) + q (q = (lane >> 2) & 3), dimensions 32 t + 16 (g & 1) + 4 p (p = lane & 3); lane li of the group
    // receives dimension 32 t + 16 (g & 1) + li = 32 t + fr of those four keys
    int v_off[DT][2][2];  // [t][plane][first / second key quad], for the key block at kbase = 0
    {
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const int key = 8 * hq + 4 * (g >> 1) + q;  // + 16 kb16 + 32 kb32: multiples of 16 leave the swizzle alone
                    const int logical = t * 8 + 4 * pl + 2 * (g & 1) + (pp >> 1);
                    const int sw = LPK == 1 ? ((key >> 1) & 1) << 2 : (key & 3) << 2;
                    v_off[t][pl][hq] = kTile + key * (LPK * 128) + ((logical ^ sw) << 4) + 8 * (pp & 1);
                }
    }

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const bool wave_active = q0 + wave * 32 < len;  // wave-uniform

    issue(0, 0);
    int stg = 0;
    for (int kt = 0; kt < len; kt += KT) {
        const int nk = min(KT, len - kt);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's lines of tile kt have landed
        __builtin_amdgcn_s_barrier();                     // everybody's have; everybody is done with tile kt - KT
        __builtin_amdgcn_sched_barrier(0);
        if (kt + KT < len) issue(kt + KT, stg ^ 1);       // under this tile's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sb = kv + stg * 2 * kTile;
        stg ^= 1;
        if (!wave_active) continue;
#pragma unroll
        for (int kb32 = 0; kb32 < KT / 32; ++kb32) {
            if (kb32 * 32 >= nk) break;
            const unsigned char* kbp = sb + kb32 * 32 * (LPK * 128);
            // ---- S^T block: 32 keys x 32 queries ----------------------------------------------------
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const x3_h8 kh = *reinterpret_cast<const x3_h8*>(kbp + k_off[kb][0]);
                const x3_h8 kl = *reinterpret_cast<const x3_h8*>(kbp + k_off[kb][1]);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[kb], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[kb], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[kb], st, 0, 0, 0);
            }
            // the V^T fragments of this key block do not depend on P: their reads fly under the softmax
            union VF {
                x3_h8 v;
                x3_s4 q[2];
            };
            VF vh[2][DT], vl[2][DT];
#pragma unroll
            for (int kb16 = 0; kb16 < 2; ++kb16)
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) {
                        const unsigned char* vp = kbp + kb16 * 16 * (LPK * 128);
                        vh[kb16][t].q[hq] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x3_lds_s4*)(vp + v_off[t][0][hq]));
                        vl[kb16][t].q[hq] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x3_lds_s4*)(vp + v_off[t][1][hq]));
                    }
            // ---- online softmax, query on the lane --------------------------------------------------
            float cmax = -INFINITY;
            if (kb32 * 32 + 32 > nk) {  // only the last block of a sequence has keys to mask (workgroup-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb32 * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    st[r] = key < nk ? st[r] : -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) cmax = fmaxf(cmax, st[r]);
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
            const float mn = fmaxf(m, cmax);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] - mn);  // masked keys: 2^(-inf) = 0
                psum += st[r];
            }
            psum += __shfl_xor(psum, 32);
            l = l * alpha + psum;
            m = mn;
            if (__ballot(alpha != 1.0f) != 0ull) {  // once the running maxima have settled no lane rescales
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
            }
            // ---- O^T += V^T . P^T: k-block kb16 = accumulator registers 8 kb16 .. + 7 ------------------
#pragma unroll
            for (int kb16 = 0; kb16 < 2; ++kb16) {
                union {
                    x3_h8 v;
                    uint32_t w[4];
                } ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) x3_split2(st[8 * kb16 + 2 * e], st[8 * kb16 + 2 * e + 1], ph.w[e], pl.w[e]);
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[kb16][t].v, ph.v, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[kb16][t].v, pl.v, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[kb16][t].v, ph.v, o[t], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_active || !qvalid) return;
    // O^T tile t: col = query fr (lane), row d = t*32 + (r&3) + 8(r>>2) + 4fh -> the (hi | lo) lines of the context image
    // (see attention_x3_kernel: lane halves trade pieces so that every store writes whole 32-byte sectors)
    const float inv = 1.0f / l;
    unsigned char* op = reinterpret_cast<unsigned char*>(ctx + (int64_t)(s0 + qrow) * H + h * HD);
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            uint32_t ha[2], la[2], hb[2], lb[2];  // a: r4 = 2 p2, b: r4 = 2 p2 + 1
            x3_split2(o[t][8 * p2] * inv, o[t][8 * p2 + 1] * inv, ha[0], la[0]);
            x3_split2(o[t][8 * p2 + 2] * inv, o[t][8 * p2 + 3] * inv, ha[1], la[1]);
            x3_split2(o[t][8 * p2 + 4] * inv, o[t][8 * p2 + 5] * inv, hb[0], lb[0]);
            x3_split2(o[t][8 * p2 + 6] * inv, o[t][8 * p2 + 7] * inv, hb[1], lb[1]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const x3_u2 sh = __builtin_amdgcn_permlane32_swap(ha[e], hb[e], false, false);
                const x3_u2 sl = __builtin_amdgcn_permlane32_swap(la[e], lb[e], false, false);
                ha[e] = sh[0];
                hb[e] = sh[1];
                la[e] = sl[0];
                lb[e] = sl[1];
            }
            *reinterpret_cast<uint4*>(op + t * 128 + 32 * p2 + 16 * fh) = uint4{ha[0], ha[1], hb[0], hb[1]};
            *reinterpret_cast<uint4*>(op + t * 128 + 64 + 32 * p2 + 16 * fh) = uint4{la[0], la[1], lb[0], lb[1]};
        }
}

__global__ void concat3_kernel(const float* a, const float* b, const float* c, int64_t n, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[i] = a[i];
        out[n + i] = b[i];
        out[2 * n + i] = c[i];
    }
}

struct LayerW {
    const float *wqkv, *bqkv;  // fused (library owned)
    const float *wo, *bo, *ln1g, *ln1b, *w1, *b1, *w2, *b2, *ln2g, *ln2b;
    // compute = 2: (hi | lo) fp16 pieces of scale * w, interleaved per 32-k block, and 1 / scale
    _Float16 *wqkv_p = nullptr, *wo_p = nullptr, *w1_p = nullptr, *w2_p = nullptr;
    float wqkv_is = 1.f, wo_is = 1.f, w1_is = 1.f, w2_is = 1.f;
};

}  // namespace

// split-K GEMMs of small batches (x3_splitk_parts): most planes, and the floats of a lane's plane buffer — planes x tokens x N
// never pass one 64 x 128 tile per CU (tiles x planes <= CUs), whatever the width
constexpr int kSplitKMax = 8;
inline int64_t x3_plane_floats(int cus) { return (int64_t)cus * 64 * 128 + 64 * 1024; }

struct GraphKey {
    int B, S, compute;
    const void *ids, *mask, *out;
    bool operator<(const GraphKey& o) const {
        return std::tie(B, S, compute, ids, mask, out) < std::tie(o.B, o.S, o.compute, o.ids, o.mask, o.out);
    }
};

struct mvdb_encoder {
    mvdb_encoder_cfg cfg;
    std::map<GraphKey, hipGraphExec_t> graphs;  // captured forwards, keyed by shape + buffer addresses
    void drop_graphs() {
        for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
        graphs.clear();
    }
    int device = 0;
    // A/B switches, read from the environment when the encoder is created (so that one process can hold encoders of both
    // kinds): MVDB_GEMM_LN_FUSED (0 never, 1 by batch size, 2 always), MVDB_ATTENTION_IMG (0: fp32 qkv + per-tile split)
    int opt_ln_fused = 1;
    bool opt_img_attn = true;
    const float *word = nullptr, *pos = nullptr, *type = nullptr, *embg = nullptr, *embb = nullptr;
    std::vector<LayerW> layers;
    std::vector<float*> owned;  // fused qkv weights / biases
    std::vector<void*> owned_h; // (hi | lo) fp16 weight images (compute = 2)
    bool have_x3 = false;
    // workspace (grown on demand), guarded by mu: one forward at a time per encoder
    std::mutex mu;
    uint64_t ws_gen = 0;                  // bumped whenever the workspace is reallocated
    int64_t cap_tokens = 0, cap_b = 0;    // lane 0 (a whole batch)
    int64_t cap1_tokens = 0, cap1_b = 0;  // lane 1 (the larger half of a split batch): its own capacity — ceil(B / 2) * S of
                                          // a later call can exceed half of what lane 0 was sized for (odd B, longer S)
    // Two lanes: lane 0 holds a whole batch; when a batch is split in two halves that run concurrently on two streams
    // (enqueue_split), lane 1 holds the second half.
    struct Lane {
        int *rank = nullptr, *count = nullptr, *seq_start = nullptr, *tok_id = nullptr, *tok_pos = nullptr,
            *tok_src = nullptr;
        float *x = nullptr, *y = nullptr, *qkv = nullptr, *ctx = nullptr, *ffn = nullptr;
        float* xp = nullptr;  // compute = 2: the (hi | lo) fp16 image of x (same bytes as x)
        unsigned int* pool_ctr = nullptr;  // [B] arrival counters of pool_norm_kernel's chunks (zero between launches)
        float* planes = nullptr;           // x3_plane_floats(CUs): [planes][padded tokens][H] partial sums of the split-K GEMMs
        void release() {
            void* ptrs[] = {rank, count, seq_start, tok_id, tok_pos, tok_src, x, y, qkv, ctx, ffn, xp, pool_ctr, planes};
            for (void* p : ptrs)
                if (p) (void)hipFree(p);
            rank = count = seq_start = tok_id = tok_pos = tok_src = nullptr;
            x = y = qkv = ctx = ffn = xp = nullptr;
            pool_ctr = nullptr;
            planes = nullptr;
        }
    } lane[2];
    hipStream_t stream2 = nullptr;             // second half of a split batch
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int32_t *ids_stage = nullptr, *mask_stage = nullptr;
    float* out_stage = nullptr;
    int64_t stage_cap = 0, out_cap = 0;
    hipStream_t stream = nullptr;
    // small batches (<= walk::kTmax token slots): the layer-walking persistent launch (encoder_walk.hpp)
    int opt_walk = 1;                       // MVDB_ENCODER_WALK as read when the encoder was created (0: the per-op kernels)
    int opt_walk_roles = 1;                 // MVDB_WALK_ROLES (0: every phase based at workgroup 0, the form before role placement)
    int opt_walk_pinned = 1;                // MVDB_WALK_PINNED (0: the host entry stages ids / mask / out through device buffers)
    walk::LayerPtrs* walk_layers = nullptr; // device copy of the per-layer weight pointers
    float *walk_x = nullptr, *walk_x1 = nullptr, *walk_qkv = nullptr, *walk_pl = nullptr, *walk_h = nullptr;
    unsigned int* walk_bar = nullptr;
    unsigned long long* walk_trace = nullptr;  // ablation build only
    unsigned int* overflow_flag = nullptr;     // device word: 1 after a forward whose pooled rows were not all finite
    PinnedBuf walk_pin;                        // host entry of the walking launch: [ids | mask | out] in ONE host-mapped buffer
    hipEvent_t walk_done = nullptr;            // recorded behind every forward of the device entry: the next one — on whatever
                                               // stream — waits for it (forwards share the staging buffers, the workspace and the
                                               // walking launch's phase counters: two resident walking grids would never finish)
    int walk_np3 = 0, walk_grid = 0, walk_grid_env = 0;
    // bounded waits (encoder_walk.hpp, Args::deadline): launches abandoned so far, on the device and mirrored into host-mapped
    // memory; calls still to be served by the per-op kernels after an abandoned launch (the GPU is being shared with
    // something that keeps the launch's workgroups from all being resident: do not pay the deadline on every call)
    unsigned int* walk_aborts = nullptr;
    unsigned int* walk_aborts_host = nullptr;  // hipHostMalloc'ed word (its device alias goes into the launch)
    unsigned int walk_deadline = 0;            // ticks of s_memrealtime (100 MHz)
    int walk_suspended = 0;
    unsigned int walk_aborts_seen = 0;         // value of *walk_aborts_host the suspension was last decided on
    unsigned long long walk_fallbacks = 0;     // forwards re-run on the per-op kernels after an abandoned launch

    void free_ws() {
        lane[0].release();
        lane[1].release();
        cap_tokens = cap_b = cap1_tokens = cap1_b = 0;
    }
};

namespace {

constexpr int kFixedWeights = 5, kPerLayer = 16;
const char* kLayerNames[kPerLayer] = {
    "attention.self.query.weight",  "attention.self.query.bias",  "attention.self.key.weight",
    "attention.self.key.bias",      "attention.self.value.weight", "attention.self.value.bias",
    "attention.output.dense.weight", "attention.output.dense.bias", "attention.output.LayerNorm.weight",
    "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
    "output.dense.weight",          "output.dense.bias",          "output.LayerNorm.weight",
    "output.LayerNorm.bias"};
const char* kFixedNames[kFixedWeights] = {
    "embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
    "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"};

int check_cfg(const mvdb_encoder_cfg* c) {
    if (!c) return fail(MVDB_ERR_ARG, "cfg is NULL");
    if (c->hidden <= 0 || c->layers <= 0 || c->heads <= 0 || c->intermediate <= 0 || c->vocab_size <= 0 ||
        c->max_positions <= 0 || c->type_vocab <= 0)
        return fail(MVDB_ERR_ARG, "encoder cfg has a non-positive size");
    if (c->hidden % c->heads) return fail(MVDB_ERR_ARG, "hidden %% heads != 0");
    const int hd = c->hidden / c->heads;
    if (hd != 32 && hd != 64) return fail(MVDB_ERR_ARG, "head_dim %d not supported (32 or 64)", hd);
    if (c->hidden % GBK || c->intermediate % GBK)
        return fail(MVDB_ERR_ARG, "hidden and intermediate must be multiples of %d", GBK);
    if (c->hidden > 1024) return fail(MVDB_ERR_ARG, "hidden > 1024 not supported");
    if (c->pooling != 0 && c->pooling != 1) return fail(MVDB_ERR_ARG, "unknown pooling mode %d", c->pooling);
    return 0;
}

template <typename T>
int dev_alloc(T** p, int64_t n) {
    MVDB_HIP(hipMalloc((void**)p, (size_t)std::max<int64_t>(n, 1) * sizeof(T)));
    return 0;
}

int alloc_lane(mvdb_encoder* e, mvdb_encoder::Lane& w, int64_t B, int64_t tokens) {
    const int64_t H = e->cfg.hidden, F = e->cfg.intermediate;
    MVDB_TRY(dev_alloc(&w.rank, tokens));
    MVDB_TRY(dev_alloc(&w.count, B));
    MVDB_TRY(dev_alloc(&w.seq_start, B + 1));
    MVDB_TRY(dev_alloc(&w.tok_id, tokens));
    MVDB_TRY(dev_alloc(&w.tok_pos, tokens));
    MVDB_TRY(dev_alloc(&w.tok_src, tokens));
    MVDB_TRY(dev_alloc(&w.x, tokens * H));
    MVDB_TRY(dev_alloc(&w.y, tokens * H));
    MVDB_TRY(dev_alloc(&w.qkv, tokens * 3 * H));
    MVDB_TRY(dev_alloc(&w.ctx, tokens * H));
    MVDB_TRY(dev_alloc(&w.ffn, tokens * std::max(F, H)));
    MVDB_TRY(dev_alloc(&w.xp, tokens * H));
    MVDB_TRY(dev_alloc(&w.planes, x3_plane_floats(device_cus(e->device))));
    MVDB_TRY(dev_alloc(&w.pool_ctr, B));
    MVDB_HIP(hipMemset(w.pool_ctr, 0, (size_t)std::max<int64_t>(B, 1) * sizeof(unsigned int)));
    return 0;
}

int ensure_ws(mvdb_encoder* e, int B, int S) {
    const int64_t tokens = (int64_t)B * S;
    const int64_t b1 = B - B / 2;  // the larger half of a split batch
    if (tokens <= e->cap_tokens && B <= e->cap_b && b1 * S <= e->cap1_tokens && b1 <= e->cap1_b) return 0;
    // grow, never shrink: each lane gets the larger of what it had and what this call needs
    const int64_t t0 = std::max(tokens, e->cap_tokens), c0 = std::max<int64_t>(B, e->cap_b);
    const int64_t t1 = std::max(b1 * S, e->cap1_tokens), c1 = std::max(b1, e->cap1_b);
    e->free_ws();
    ++e->ws_gen;
    MVDB_TRY(alloc_lane(e, e->lane[0], c0, t0));
    MVDB_TRY(alloc_lane(e, e->lane[1], c1, t1));
    e->cap_tokens = t0;
    e->cap_b = c0;
    e->cap1_tokens = t1;
    e->cap1_b = c1;
    return 0;
}

template <int EPI>
void launch_gemm(const float* A, const float* W, const float* bias, const float* R, float* C,
                 const int* Tptr, int64_t Tmax, int N, int K, int cus, hipStream_t s) {
    const int64_t big = (int64_t)((N + 127) / 128) * ((Tmax + 127) / 128);
    // 128x128 tiles only when they fill the chip's resident slots (2 blocks per CU) several times over;
    // otherwise the last, partly filled round dominates (T = 8192: 576-768 tiles = 1.1-1.5 rounds) and
    // 64x64 tiles (4x the blocks) are faster despite their lower per-block reuse.
    static const int rounds = []() {
        const char* v = getenv("MVDB_GEMM_BIG_TILE_ROUNDS");
        return v && *v ? atoi(v) : 3;
    }();
    static const bool big_bk32 = []() {
        const char* v = getenv("MVDB_GEMM_BIG_BK32");
        return v && *v == '1';
    }();
    static const bool small_bk32 = []() {
        const char* v = getenv("MVDB_GEMM_SMALL_BK32");
        return !(v && *v == '0');
    }();
    static const bool use_dma = []() {
        const char* v = getenv("MVDB_GEMM_DMA");
        return !(v && *v == '0');
    }();
    if (use_dma && K % 32 == 0 && big < (int64_t)rounds * 2 * cus) {
        // 64x64 tiles, LDS-DMA staged, three 16-KiB stages: 3 blocks per CU — T = 8192: 9 / 12 / 3 tiles per CU for
        // N = 1152 / 1536 / 384, whole rounds in every GEMM (the register-staged kernel fits 4 blocks: 2.25 rounds
        // at N = 1152).  Measured 4.12 vs 4.26 ms per forward at B = 256, S = 32.  With 128x128 tiles (96 KiB of LDS,
        // one block per CU) it loses to the register-staged kernel (81 vs 67 ms at S = 512): not used there.
        constexpr int lds = 3 * 128 * 128;
        dim3 grid((N + 63) / 64, (unsigned)((Tmax + 63) / 64));
        hipLaunchKernelGGL((gemm_f32_dma_kernel<EPI, 1>), grid, dim3(256), lds, s, A, W, bias, R, C, Tptr, N, K);
        return;
    }
    if (big >= (int64_t)rounds * 2 * cus) {
        dim3 grid((N + 127) / 128, (unsigned)((Tmax + 127) / 128));
        if (K % 32 == 0 && big_bk32)
            hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, 2, 32>), grid, dim3(256), 0, s, A, W, bias, R, C, Tptr, N, K);
        else
            hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, 2, 16>), grid, dim3(256), 0, s, A, W, bias, R, C, Tptr, N, K);
    } else {
        dim3 grid((N + 63) / 64, (unsigned)((Tmax + 63) / 64));
        // 64x64 tiles do 8 MFMAs per wave and 16-deep step: step twice as deep to halve the barriers
        if (K % 32 == 0 && small_bk32)
            hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, 1, 32>), grid, dim3(256), 0, s, A, W, bias, R, C, Tptr, N, K);
        else
            hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, 1, 16>), grid, dim3(256), 0, s, A, W, bias, R, C, Tptr, N, K);
    }
}

// compute = 2: LDS-DMA kernel, 64 x 128 tiles, three stages, two workgroups per CU — measured best at every shape
// (B = 256: S = 32 2.68 ms vs 3.04 with 128-row tiles for the wide GEMMs, 3.15 register-staged; S = 512 43.8 ms vs
// 55.7 / 45.3; rings of 2, 4 or 6 stages, i.e. 3 or 1 workgroups per CU: within 1 %)
// Raises a kernel's dynamic-LDS limit once per (kernel, device).
int x3_set_lds(const void* kern, int lds, int device) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> done;
    std::lock_guard<std::mutex> lk(mu);
    int& have = done[{kern, device}];
    if (lds > have) {
        MVDB_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        have = lds;
    }
    return 0;
}

// Timing ablations of the split-precision GEMMs (DBG template parameter: 1 no fragment reads / MFMAs, 2 no DMA, 3 no
// epilogue, 4 no K loop; results invalid) exist only in builds with -DMVDB_X3_ABLATE (make ABLATE=1), selected at run time
// by MVDB_GEMM_X3_DBG; the normal build instantiates the real kernels only.
#ifdef MVDB_X3_ABLATE
#define X3_KERN(...)                                                                                                  \
    (dbg == 1 ? gemm_x3_dma_kernel<__VA_ARGS__, 1> : dbg == 2 ? gemm_x3_dma_kernel<__VA_ARGS__, 2> :                 \
     dbg == 3 ? gemm_x3_dma_kernel<__VA_ARGS__, 3> : dbg == 4 ? gemm_x3_dma_kernel<__VA_ARGS__, 4> :                 \
     dbg == 5 ? gemm_x3_dma_kernel<__VA_ARGS__, 5> : gemm_x3_dma_kernel<__VA_ARGS__, 0>)
#define X3_BIG_KERN(...)                                                                                              \
    (dbg == 1 ? gemm_x3_big_kernel<__VA_ARGS__, 1> : dbg == 2 ? gemm_x3_big_kernel<__VA_ARGS__, 2> :                 \
     dbg == 7 ? gemm_x3_big_kernel<__VA_ARGS__, 7> :                                                                  \
     dbg == 3 ? gemm_x3_big_kernel<__VA_ARGS__, 3> : dbg == 4 ? gemm_x3_big_kernel<__VA_ARGS__, 4> :                 \
     dbg == 5 ? gemm_x3_big_kernel<__VA_ARGS__, 5> : dbg == 6 ? gemm_x3_big_kernel<__VA_ARGS__, 6> : gemm_x3_big_kernel<__VA_ARGS__, 0>)
#define X3_LN_KERN(...)                                                                                               \
    (dbg == 1 ? gemm_x3_ln_kernel<__VA_ARGS__, 1> : dbg == 2 ? gemm_x3_ln_kernel<__VA_ARGS__, 2> :                   \
     dbg == 3 ? gemm_x3_ln_kernel<__VA_ARGS__, 3> : dbg == 4 ? gemm_x3_ln_kernel<__VA_ARGS__, 4> : gemm_x3_ln_kernel<__VA_ARGS__, 0>)
#else
#define X3_KERN(...) gemm_x3_dma_kernel<__VA_ARGS__, 0>
#define X3_BIG_KERN(...) gemm_x3_big_kernel<__VA_ARGS__, 0>
#define X3_LN_KERN(...) gemm_x3_ln_kernel<__VA_ARGS__, 0>
#endif

template <int EPI>
int launch_gemm_x3(const float* Aimg, const _Float16* Wp, float inv_wscale, const float* bias, const float* R, float* C,
                   const int* Tptr, int64_t Tmax, int N, int K, int device, hipStream_t s, int qcols = 0, float qscale = 1.f) {
    const _Float16* A = reinterpret_cast<const _Float16*>(Aimg);  // [T][K / 32][hi 32 | lo 32]: the bytes of a [T][K] fp32 matrix
    static const int dbg = []() { const char* v = getenv("MVDB_GEMM_X3_DBG"); return v ? atoi(v) : 0; }();
    (void)dbg;
    static const bool spread_small = []() { const char* v = getenv("MVDB_GEMM_X3_SPREAD_SMALL"); return !(v && *v == '0'); }();
    auto kern = spread_small ? X3_KERN(EPI, 64, 3, 4, 128, 1) : X3_KERN(EPI, 64, 3, 4, 128, 0);
    constexpr int lds = 3 * (64 * 128 + 128 * 128);
    MVDB_TRY(x3_set_lds((const void*)kern, lds, device));
    dim3 grid((N + 127) / 128, (unsigned)((Tmax + 63) / 64));
    // Enough 128 x 128 tiles to fill every resident slot (two workgroups per CU) at least once: 128 x 128 tiles on FOUR
    // waves — wave tile 64 x 64, 8 fragment reads per 12 MFMAs where the 32 x 64 wave tile of the default needs 12, and a
    // third less L2 -> LDS traffic per output —, two stages (64 KiB), two workgroups per CU.  e5-small, S = 512: 29.3 ms
    // per forward vs 33.1 with the default tiles and 32.2 with 128 x 128 tiles on eight waves of 32 x 64
    // (MVDB_GEMM_X3_W8=1).  The threshold (MVDB_GEMM_X3_MANY, in units of the CU count, counted on the padded batch):
    // at 2 an e5-large-shaped forward (H = 1024, 24 layers) of 256 x 32 tokens takes 16.7 ms against 18.3 at 8, 17.4 at 3
    // (ragged: 12.9 / 12.9 / 12.5); e5-small at T = 8192 — QKV and FFN1 qualify, 1.1 / 1.5 rounds — is within the
    // run-to-run spread either way (1.97 vs 1.99 ms, ragged 1.64 vs 1.62).  Forcing them onto the N = 384 GEMMs as well
    // (192 tiles) costs a ragged batch 30 %.
    static const int big4env = []() { const char* v = getenv("MVDB_GEMM_X3_BM128W4"); return v ? atoi(v) : -1; }();
    static const bool w8 = []() { const char* v = getenv("MVDB_GEMM_X3_W8"); return v && *v == '1'; }();
    static const int many_x = []() { const char* v = getenv("MVDB_GEMM_X3_MANY"); return v && *v ? atoi(v) : 2; }();
    const bool many = (int64_t)((N + 127) / 128) * ((Tmax + 127) / 128) >= (int64_t)many_x * device_cus(device);
    // N a multiple of 256 and whole rounds of 256 x 256 tiles: ONE workgroup of eight waves per CU on a 256 x 256 tile (wave
    // tile 64 x 128: 24 fragment reads per 48 MFMAs; two 64-KiB stages) — half the L2 -> LDS bytes per output of two
    // 128 x 128 workgroups.  PMC on the e5-large shape at 256 x 512 tokens: matrix cores 61 % busy in the QKV GEMM, 60 %
    // in the N = 1024 ones, 53 % in FFN1 (GELU epilogue).  Forward of that shape (24 layers): 294.9 -> 265.2 ms (ragged
    // 189.4 -> 177.9), S = 256 137.0 -> 122.9, S = 128 66.8 -> 60.2 (ragged 45.0 -> 46.4), S = 64 33.7 -> 29.8; at S = 32 its
    // 384 / 128 tiles per GEMM are 1.5 / 0.5 rounds of the 256 CUs and it loses (16.6 -> 17.9 ms, ragged 12.9 -> 16.8) — hence
    // the rule: at least one round, and either >= 4 rounds or a last round that is >= 85 % full (counted on the padded
    // batch).  MVDB_GEMM_X3_BIG: 1 forces it wherever N % 256 == 0 or N % 192 == 0, 0 disables it.
    static const int big8env = []() { const char* v = getenv("MVDB_GEMM_X3_BIG"); return v ? atoi(v) : -1; }();
    // N % 256 != 0 but N % 192 == 0 (every GEMM of a 384-wide model): the same form on 256 x 192 tiles (wave tile 64 x 96)
    const int bign = N % 256 == 0 ? 256 : N % 192 == 0 ? 192 : 0;
    const int cus = device_cus(device);
    // forced (1): unconditional; default: launched as a pair with the fallback below when the padded batch could qualify
    const bool big_pair = bign != 0 && big8env < 0 && x3_big_form(Tmax, N, bign, cus);
    int sel_bn = 0;  // the fallback's selector
    if (!w8 && big4env < 0 && bign != 0 && (big8env == 1 || big_pair)) {
        const dim3 gridb(N / bign, (unsigned)((Tmax + 255) / 256));
        const int sel_big = big8env == 1 ? 0 : bign;
        // one persistent workgroup per CU walking the tile list (gemm_x3_big_kernel); MVDB_GEMM_X3_PERSIST=0: one workgroup per tile
        static const bool persist = []() { const char* v = getenv("MVDB_GEMM_X3_PERSIST"); return !(v && *v == '0'); }();
        if (persist) {
            const unsigned gp = (unsigned)std::min<int64_t>((int64_t)gridb.x * gridb.y, cus);
            static const bool spread = []() { const char* v = getenv("MVDB_GEMM_X3_SPREAD"); return !(v && *v == '0'); }();
            if (bign == 256) {
                auto kernp = spread ? X3_BIG_KERN(EPI, 256, 1) : X3_BIG_KERN(EPI, 256, 0);
                constexpr int ldsp = 2 * (256 * 128 + 256 * 128);
                MVDB_TRY(x3_set_lds((const void*)kernp, ldsp, device));
                hipLaunchKernelGGL(kernp, dim3(gp), dim3(512), ldsp, s, A, Wp, inv_wscale, bias, R, C, Tptr, N, K, sel_big, cus, qcols, qscale);
            } else {
                auto kernp = spread ? X3_BIG_KERN(EPI, 192, 1) : X3_BIG_KERN(EPI, 192, 0);
                constexpr int ldsp = 2 * (256 * 128 + 192 * 128);
                MVDB_TRY(x3_set_lds((const void*)kernp, ldsp, device));
                hipLaunchKernelGGL(kernp, dim3(gp), dim3(512), ldsp, s, A, Wp, inv_wscale, bias, R, C, Tptr, N, K, sel_big, cus, qcols, qscale);
            }
        } else if (bign == 256) {
            auto kernb = X3_KERN(EPI, 256, 2, 8, 256, 0);
            constexpr int ldsb = 2 * (256 * 128 + 256 * 128);
            MVDB_TRY(x3_set_lds((const void*)kernb, ldsb, device));
            hipLaunchKernelGGL(kernb, gridb, dim3(512), ldsb, s, A, Wp, inv_wscale, bias, R, C, Tptr, N, K, sel_big, cus, qcols, qscale);
        } else {
            auto kernc = X3_KERN(EPI, 256, 2, 8, 192, 0);
            constexpr int ldsc = 2 * (256 * 128 + 192 * 128);
            MVDB_TRY(x3_set_lds((const void*)kernc, ldsc, device));
            hipLaunchKernelGGL(kernc, gridb, dim3(512), ldsc, s, A, Wp, inv_wscale, bias, R, C, Tptr, N, K, sel_big, cus, qcols, qscale);
        }
        if (big8env == 1) return 0;
        sel_bn = -bign;  // the fallback runs only where the packed token count rules the 256-row form out
    }
    // (Tried in round 3 and dropped: the persistent eight-wave kernel on 128 x 192 / 128 x 256 tiles (wave tile 32 x 96 / 32 x 128) for
    // batches too small for the 256-row forms — S = 32 forward 2.37 ms vs 2.04 with the 128 x 128 four-wave tiles below.)
    // (Tried in round 3 and dropped: 128 x 192 tiles on four waves, two 40-KiB stages, TWO workgroups per CU so that one
    // workgroup's stores run under the other's K loop — S = 512 forward 26.7 ms vs 25.9 with the 256-row forms, S = 32 2.01
    // vs 1.96: the epilogue is not what the 256-row forms wait for.)
    if (!w8 && (big4env >= 0 ? big4env == 1 : many)) {
        auto kern4 = spread_small ? X3_KERN(EPI, 128, 2, 4, 128, 1) : X3_KERN(EPI, 128, 2, 4, 128, 0);
        constexpr int lds4 = 2 * (128 * 128 + 128 * 128);
        MVDB_TRY(x3_set_lds((const void*)kern4, lds4, device));
        dim3 grid4((N + 127) / 128, (unsigned)((Tmax + 127) / 128));
        hipLaunchKernelGGL(kern4, grid4, dim3(256), lds4, s, A, Wp, inv_wscale, bias, R, C, Tptr, N, K, sel_bn, cus, qcols, qscale);
        return 0;
    }
    if (w8) {
        auto kern8 = X3_KERN(EPI, 128, 3, 8, 128, 0);
        constexpr int lds8 = 3 * (128 * 128 + 128 * 128);
        MVDB_TRY(x3_set_lds((const void*)kern8, lds8, device));
        dim3 grid8((N + 127) / 128, (unsigned)((Tmax + 127) / 128));
        hipLaunchKernelGGL(kern8, grid8, dim3(512), lds8, s, A, Wp, inv_wscale, bias, R, C, Tptr, N, K, sel_bn, cus, qcols, qscale);
        return 0;
    }
    // Few tiles (one long sentence, a handful of short ones): 64 x 64 tiles, twice the workgroups — a K-step of a 64 x 128 tile is
    // 12 MFMAs per wave (~0.33 us), and with a quarter of the CUs busy that, not the memory system, is what the GEMM takes
    static const bool bn64 = []() { const char* v = getenv("MVDB_GEMM_X3_BN64"); return !(v && *v == '0'); }();
    // (one workgroup per CU at most; up to 1.5 / 2 per CU measured: +-2 %, benchmarks/mid_batch_probe.py)
    if (bn64 && sel_bn == 0 && N % 64 == 0 && (int64_t)grid.x * grid.y * 2 <= cus) {
        auto kern64 = X3_KERN(EPI, 64, 3, 4, 64, 1);
        constexpr int lds64 = 3 * (64 * 128 + 64 * 128);
        MVDB_TRY(x3_set_lds((const void*)kern64, lds64, device));
        dim3 grid64(N / 64, (unsigned)((Tmax + 63) / 64));
        hipLaunchKernelGGL(kern64, grid64, dim3(256), lds64, s, A, Wp, inv_wscale, bias, R, C, Tptr, N, K, 0, cus, qcols, qscale);
        return 0;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, A, Wp, inv_wscale, bias, R, C, Tptr, N, K, sel_bn, cus, qcols, qscale);
    return 0;
}

// ---- N == H GEMM with bias + residual + LayerNorm in the epilogue (gemm_x3_ln_kernel) --------------------------------
bool x3_ln_fusable(int H) { return H % 128 == 0 && H <= 512; }

template <int BM, int WM, int TN>
int launch_gemm_x3_ln_inst(const _Float16* A, const _Float16* Wp, float inv_wscale, const float* bias, const float* gamma,
                           const float* beta, float eps, float* X, float* Xp, const int* Tptr, int64_t Tmax, int K, int device,
                           hipStream_t s) {
    constexpr int kStage = (BM + 128 * TN) * 128;
    constexpr int NST = 3 * kStage <= 160 * 1024 ? 3 : 2;
    static_assert(2 * kStage <= 160 * 1024, "two stages must fit the CU's LDS");
    static const int dbg = []() { const char* v = getenv("MVDB_GEMM_X3_DBG"); return v ? atoi(v) : 0; }();
    (void)dbg;
    static const bool spread = []() { const char* v = getenv("MVDB_GEMM_LN_SPREAD"); return !(v && *v == '0'); }();
    auto kern = spread ? X3_LN_KERN(BM, WM, TN, NST, 1) : X3_LN_KERN(BM, WM, TN, NST, 0);
    constexpr int lds = NST * kStage + 2 * 4 * BM * 4;  // the ring + the row statistics
    static_assert(lds <= 160 * 1024, "LDS budget of a CU");
    MVDB_TRY(x3_set_lds((const void*)kern, lds, device));
    // two-stage forms: persistent, one workgroup per CU (a multiple of the row bands would leave a last round part full anyway)
    const int64_t bands = (Tmax + BM - 1) / BM;
    static const bool persist = []() { const char* v = getenv("MVDB_GEMM_LN_PERSIST"); return !(v && *v == '0'); }();  // 0: one band per workgroup (A/B)
    const unsigned grid = (unsigned)(NST == 2 && persist ? std::min<int64_t>(bands, device_cus(device)) : bands);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * 256), lds, s, A, Wp, inv_wscale, bias, gamma, beta, eps, X, Xp, Tptr, K);
    return 0;
}

template <int TN>
int launch_gemm_x3_ln_tn(int bm, const _Float16* A, const _Float16* Wp, float inv_wscale, const float* bias, const float* gamma,
                         const float* beta, float eps, float* X, float* Xp, const int* Tptr, int64_t Tmax, int K, int device,
                         hipStream_t s) {
    // (H = 512 at 128 rows: two 80-KiB stages are the whole LDS, no room for the row statistics behind them: 64 rows there)
    if constexpr (TN < 4)
        if (bm == 128) return launch_gemm_x3_ln_inst<128, 2, TN>(A, Wp, inv_wscale, bias, gamma, beta, eps, X, Xp, Tptr, Tmax, K, device, s);
    if (bm == 128) bm = 64;
    if (bm == 64) return launch_gemm_x3_ln_inst<64, 2, TN>(A, Wp, inv_wscale, bias, gamma, beta, eps, X, Xp, Tptr, Tmax, K, device, s);
    return launch_gemm_x3_ln_inst<32, 1, TN>(A, Wp, inv_wscale, bias, gamma, beta, eps, X, Xp, Tptr, Tmax, K, device, s);
}

// X (in: the residual rows; out: LayerNorm(A W^T + bias + X), fp32) and Xp (out: its (hi | lo) image); N = H = 128 TN.
// Rows per workgroup: the largest of 128 / 64 / 32 that still gives every CU a workgroup (counted on the padded batch) —
// a workgroup streams ALL of W whatever its height, so taller tiles mean less L2 -> LDS traffic per output row, but at
// T = 8192 only 32-row tiles reach all 256 CUs.  MVDB_GEMM_LN_BM overrides.
int launch_gemm_x3_ln(const float* Aimg, const _Float16* Wp, float inv_wscale, const float* bias, const float* gamma,
                      const float* beta, float eps, float* X, float* Xp, const int* Tptr, int64_t Tmax, int H, int K, int device,
                      hipStream_t s) {
    const _Float16* A = reinterpret_cast<const _Float16*>(Aimg);
    static const int bm_env = []() { const char* v = getenv("MVDB_GEMM_LN_BM"); return v && *v ? atoi(v) : 0; }();
    const int cus = device_cus(device);
    int bm = (Tmax + 127) / 128 >= cus ? 128 : (Tmax + 63) / 64 >= cus ? 64 : 32;
    if (bm_env == 32 || bm_env == 64 || bm_env == 128) bm = bm_env;
    switch (H / 128) {
        case 1: return launch_gemm_x3_ln_tn<1>(bm, A, Wp, inv_wscale, bias, gamma, beta, eps, X, Xp, Tptr, Tmax, K, device, s);
        case 2: return launch_gemm_x3_ln_tn<2>(bm, A, Wp, inv_wscale, bias, gamma, beta, eps, X, Xp, Tptr, Tmax, K, device, s);
        case 3: return launch_gemm_x3_ln_tn<3>(bm, A, Wp, inv_wscale, bias, gamma, beta, eps, X, Xp, Tptr, Tmax, K, device, s);
        case 4: return launch_gemm_x3_ln_tn<4>(bm, A, Wp, inv_wscale, bias, gamma, beta, eps, X, Xp, Tptr, Tmax, K, device, s);
        default: return fail(MVDB_ERR_ARG, "no LayerNorm-fused GEMM for H = %d", H);
    }
}

// one tensor: max|w| -> power-of-two scale with max|w| * scale in [2^13, 2^14) -> interleaved (hi | lo) fp16 pieces
int make_interleaved(mvdb_encoder* e, const float* src, int64_t n, _Float16** out, float* inv_scale, unsigned int* scratch,
                     hipStream_t s) {
    MVDB_HIP(hipMemsetAsync(scratch, 0, sizeof(unsigned int), s));
    hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, s, src, n, scratch);
    unsigned int bits = 0;
    MVDB_HIP(hipMemcpyAsync(&bits, scratch, sizeof(bits), hipMemcpyDeviceToHost, s));
    MVDB_HIP(hipStreamSynchronize(s));
    float mx;
    memcpy(&mx, &bits, sizeof(mx));
    if (!(mx < 3.0e38f)) return fail(MVDB_ERR_ARG, "encoder weights hold a non-finite value");
    int ex = 0;
    if (mx > 0.f) (void)std::frexp(mx, &ex);          // mx = m 2^ex, m in [0.5, 1)
    int sexp = mx > 0.f ? 14 - ex : 0;
    sexp = std::max(-100, std::min(100, sexp));
    const float scale = std::ldexp(1.f, sexp);
    *inv_scale = std::ldexp(1.f, -sexp);
    _Float16* p = nullptr;
    MVDB_HIP(hipMalloc((void**)&p, (size_t)n * 2 * sizeof(_Float16)));
    e->owned_h.push_back(p);
    hipLaunchKernelGGL(f32_split_interleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, p, n, scale);
    *out = p;
    return 0;
}

int ensure_x3_weights(mvdb_encoder* e, hipStream_t s) {
    if (e->have_x3) return 0;
    const int64_t H = e->cfg.hidden, F = e->cfg.intermediate;
    unsigned int* scratch = nullptr;
    MVDB_HIP(hipMalloc((void**)&scratch, sizeof(unsigned int)));
    int rc = 0;
    for (LayerW& L : e->layers) {
        if (!rc) rc = make_interleaved(e, L.wqkv, 3 * H * H, &L.wqkv_p, &L.wqkv_is, scratch, s);
        if (!rc) rc = make_interleaved(e, L.wo, H * H, &L.wo_p, &L.wo_is, scratch, s);
        if (!rc) rc = make_interleaved(e, L.w1, F * H, &L.w1_p, &L.w1_is, scratch, s);
        if (!rc) rc = make_interleaved(e, L.w2, H * F, &L.w2_p, &L.w2_is, scratch, s);
    }
    (void)hipFree(scratch);
    if (rc) return rc;
    MVDB_HIP(hipGetLastError());
    e->have_x3 = true;
    return 0;
}

// Small batches (one sentence per call is the reference's own API shape): the N = H GEMM over K = F is 3 column tiles x a
// handful of row bands, each walking all F / 32 K-steps alone — 21 us of a 60-us layer at T = 16.  Split over K instead:
// `parts` planes of bare partial sums (EPI_PARTIAL), added up with the bias and the residual by ln_partials_kernel.
// Used when the unsplit grid would leave three quarters of the CUs idle and K is long (x3_splitk_parts); the summation order over K
// then differs from the unsplit kernel's (rounding-level differences between a sentence embedded alone and in a large batch).
// MVDB_GEMM_X3_SPLITK=0 switches it off.
// Planes: K-steps are a latency chain (one barrier + one DMA round trip each, ~0.4 us), so a GEMM whose 64 x 128 tiles do not fill
// the CUs is split over K into as many planes as (a) leave each >= 4 K-steps (MVDB_GEMM_X3_SPLITK_MINSTEPS; 8 / 3 / 2 measured:
// 8 loses 4-7 % on the e5-small shape, whose K = 384 GEMM then stays unsplit, 3 and 2 change nothing), (b) keep the launch at
// one workgroup per CU (tiles x planes <= CUs: 8 planes x 64 tiles were slower than 3 at T = 512) and (c) ln_partials_kernel
// has an unrolled form for (2, 3, 4, 6, 8).  Until round 6: three planes, only below a quarter of the CUs in tiles.  One long
// sentence (profiles/r06_long_sentence_chain.txt): e5-small 256 tokens 0.71 -> 0.65 ms, large 129 tokens 2.22 -> 2.02; batches
// of 32 - 64 short sentences (benchmarks/mid_batch_probe.py): e5-small 64 x 32 tokens 0.92 -> 0.78 ms, large 32 x 32 3.93 -> 3.20.
// MVDB_GEMM_X3_SPLITK_PARTS=3 forces three planes (A/B), MVDB_GEMM_X3_SPLITK=0 switches the split off.
int x3_splitk_parts(int64_t Tmax, int N, int K, int cus) {
    static const bool on = []() { const char* v = getenv("MVDB_GEMM_X3_SPLITK"); return !(v && *v == '0'); }();
    static const int forced = []() { const char* v = getenv("MVDB_GEMM_X3_SPLITK_PARTS"); return v && *v ? atoi(v) : 0; }();
    static const int min_steps = []() { const char* v = getenv("MVDB_GEMM_X3_SPLITK_MINSTEPS"); return v && *v ? std::max(1, atoi(v)) : 4; }();
    if (!on) return 0;
    const int64_t tiles = ((Tmax + 63) / 64) * ((N + 127) / 128);
    int64_t parts = std::min<int64_t>({(int64_t)kSplitKMax, (int64_t)(K / 32 / min_steps), (int64_t)cus / std::max<int64_t>(tiles, 1)});
    if (forced >= 2) parts = std::min<int64_t>({(int64_t)forced, (int64_t)kSplitKMax, (int64_t)(K / 32)});
    if (parts < 2 || parts * Tmax * N > x3_plane_floats(cus)) return 0;
    return parts == 5 ? 4 : parts == 7 ? 6 : (int)parts;
}

int launch_gemm_x3_splitk(const float* Aimg, const _Float16* Wp, float inv_wscale, float* planes, const int* Tptr, int64_t Tmax,
                          int N, int K, int parts, int device, hipStream_t s) {
    const _Float16* A = reinterpret_cast<const _Float16*>(Aimg);
    auto kern = gemm_x3_dma_kernel<EPI_PARTIAL, 64, 3, 4, 128, 1, 0>;
    constexpr int lds = 3 * (64 * 128 + 128 * 128);
    MVDB_TRY(x3_set_lds((const void*)kern, lds, device));
    dim3 grid((N + 127) / 128, (unsigned)((Tmax + 63) / 64), (unsigned)parts);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, A, Wp, inv_wscale, (const float*)nullptr, (const float*)nullptr, planes, Tptr, N, K,
                       0, device_cus(device), (int)Tmax, 1.f);
    return 0;
}

// the planes of a split-K QKV / FFN1 GEMM -> bias (+ query scale / GELU) -> (hi | lo) image (partials_image_kernel)
template <int EPI>
int launch_partials_image(const float* planes, int parts, const float* bias, const int* seq_start, int B, int64_t Tmax, int N, int qcols,
                          float qscale, float* Cimg, hipStream_t s) {
    const dim3 grid((unsigned)((N + 255) / 256), (unsigned)((Tmax + 3) / 4));
    const int64_t plane = Tmax * N;
#define MVDB_PI(NPV) hipLaunchKernelGGL((partials_image_kernel<EPI, NPV>), grid, dim3(256), 0, s, planes, plane, bias, seq_start, B, N, qcols, qscale, Cimg)
    switch (parts) {
        case 2: MVDB_PI(2); break;
        case 3: MVDB_PI(3); break;
        case 4: MVDB_PI(4); break;
        case 6: MVDB_PI(6); break;
        case 8: MVDB_PI(8); break;
        default: return fail(MVDB_ERR_ARG, "internal: no partials_image_kernel for %d planes", parts);
    }
#undef MVDB_PI
    MVDB_HIP(hipGetLastError());
    return 0;
}
// ... worth its extra launch (~5 us) where the GEMM walks many K-steps: K >= 768 (the wide shapes) and three planes or more
int x3_splitk_parts_wide(int64_t Tmax, int N, int K, int cus) {
    static const bool on = []() { const char* v = getenv("MVDB_GEMM_X3_SPLITK_WIDE"); return !(v && *v == '0'); }();
    if (!on || K < 768 || N % 64) return 0;
    const int parts = x3_splitk_parts(Tmax, N, K, cus);
    return parts >= 3 ? parts : 0;
}

template <int VPT>
void launch_ln_partials(const float* planes, int parts, int64_t plane, const float* bias, const int* seq_start, int B, const float* g,
                        const float* b, float eps, int H, float* x, float* xp, int64_t Tmax, hipStream_t s) {
    const dim3 grid((unsigned)((Tmax + 3) / 4));
#define MVDB_LNP(FULLV, NPV)                                                                                                          \
    hipLaunchKernelGGL((ln_partials_kernel<VPT, FULLV, NPV>), grid, dim3(256), 0, s, planes, parts, plane, bias, seq_start, B, g, b, \
                       eps, H, x, xp)
    if (H == VPT * 64) {
        if (parts == 3) { MVDB_LNP(true, 3); return; }
        // (the other unrolled forms only at the widths of the reference's models: H = 384 / 768 / 1024)
        if constexpr (VPT == 6 || VPT == 12 || VPT == 16) {
            if (parts == 2) { MVDB_LNP(true, 2); return; }
            if (parts == 4) { MVDB_LNP(true, 4); return; }
            if (parts == 6) { MVDB_LNP(true, 6); return; }
            if (parts == 8) { MVDB_LNP(true, 8); return; }
        }
        MVDB_LNP(true, 0);
    } else {
        if (parts == 3) MVDB_LNP(false, 3);
        else MVDB_LNP(false, 0);
    }
#undef MVDB_LNP
}

template <int VPT>
void launch_ln(const float* y, const int* seq_start, int B, const float* g, const float* b, float eps,
               int H, float* x, float* xp, int64_t Tmax, hipStream_t s) {
    if (H == VPT * 64)  // branch-free rows (384, 1024, ...)
        hipLaunchKernelGGL((ln_kernel<VPT, true>), dim3((unsigned)((Tmax + 3) / 4)), dim3(256), 0, s, y, seq_start, B,
                           g, b, eps, H, x, xp);
    else
        hipLaunchKernelGGL((ln_kernel<VPT, false>), dim3((unsigned)((Tmax + 3) / 4)), dim3(256), 0, s, y, seq_start, B,
                           g, b, eps, H, x, xp);
}


// Walking launches are persistent grids whose workgroups spin on each other: two of them resident at once can each hold CUs the
// other's missing workgroups need.  Three guards, from cheap to last resort:
//  1. IN THE PROCESS: one walking launch at a time per DEVICE — an event behind every launch, waited for by the next (whatever
//     encoder, stream or host thread it comes from), and g_walk_mu held across [wait for that event, launch, record it] so
//     that two host threads cannot both enqueue behind the same predecessor (WalkTurn).
//  2. ACROSS PROCESSES: an advisory flock on a file named after the GPU's UUID, held from before the launch until the launch
//     has completed (the host entry waits for its stream anyway; the device entry releases it from a host callback behind
//     the launch).  It is advisory: a process that cannot have it within kGateWaitUs runs that forward on the per-op kernels.
//  3. IN THE KERNEL: every wait is bounded (encoder_walk.hpp, Args::deadline).  A launch that cannot complete — a foreign
//     persistent kernel, a process that ignores the lock — abandons itself, is counted (mvdb_encoder_walk_aborts) and its
//     forward re-runs on the per-op kernels; the GPU is never wedged.
// A caller that CAPTURES the stream gets the per-op kernels: a captured launch can take part in none of 1 and 2.
std::mutex g_walk_mu;   // the enqueue order of walking launches
std::mutex g_gate_mu;   // the gates' bookkeeping only: never held across a HIP call (gate_release runs in a host callback)
struct WalkOrder {
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;  // stream of the last walking launch: the next one on the SAME stream is ordered by the stream
    bool any = false;
};
struct WalkGate {
    int fd = -2;       // -2: not opened yet, -1: no lock file to be had (guard 3 alone)
    int inflight = 0;  // walking launches of THIS process between acquire and release: the flock is held while > 0
};
std::map<int, WalkOrder> g_walk_order;
std::map<int, WalkGate> g_walk_gate;
constexpr int kGateWaitUs = 50000;
constexpr int kPinSlots = 512;          // token slots the host entry serves from ONE host-mapped buffer (no copies)
constexpr int kGateBusyCalls = 64;      // forwards on the per-op kernels after the gate could not be had within kGateWaitUs
constexpr int kWalkSuspendCalls = 256;  // forwards on the per-op kernels after an abandoned launch, before the next attempt

int gate_open(int device) {
    const char* off = getenv("MVDB_WALK_LOCK");
    if (off && *off == '0') return -1;
    hipUUID u;
    char name[80];
    if (hipDeviceGetUuid(&u, device) == hipSuccess) {
        char hex[33];
        for (int i = 0; i < 16; ++i) snprintf(hex + 2 * i, 3, "%02x", (unsigned char)u.bytes[i]);
        snprintf(name, sizeof(name), "mvdb_walk_%s.lock", hex);
    } else {
        (void)hipGetLastError();
        char bus[32] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) {
            (void)hipGetLastError();
            snprintf(bus, sizeof(bus), "dev%d", device);
        }
        for (char* c = bus; *c; ++c)
            if (*c == ':' || *c == '.' || *c == '/') *c = '_';
        snprintf(name, sizeof(name), "mvdb_walk_%s.lock", bus);
    }
    const char* dirs[] = {getenv("MVDB_WALK_LOCK_DIR"), "/dev/shm", "/tmp"};
    for (const char* d : dirs) {
        if (!d || !*d) continue;
        char path[256];
        snprintf(path, sizeof(path), "%s/%s", d, name);
        const int fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC, 0666);
        if (fd >= 0) {
            (void)fchmod(fd, 0666);  // other users' processes share the GPU too
            return fd;
        }
    }
    return -1;
}

// 1: this process holds the device's gate (or there is no lock file); 0: another process kept it for kGateWaitUs
int gate_acquire(int device) {
    std::lock_guard<std::mutex> lk(g_gate_mu);
    WalkGate& g = g_walk_gate[device];
    if (g.fd == -2) g.fd = gate_open(device);
    if (g.fd >= 0 && g.inflight == 0) {
        const auto t0 = std::chrono::steady_clock::now();
        while (flock(g.fd, LOCK_EX | LOCK_NB) != 0) {
            if (errno != EWOULDBLOCK && errno != EINTR) {  // the file system does not lock: guard 3 alone from now on
                close(g.fd);
                g.fd = -1;
                break;
            }
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > kGateWaitUs) return 0;
            usleep(20);
        }
    }
    ++g.inflight;
    return 1;
}
void gate_release(int device) {
    std::lock_guard<std::mutex> lk(g_gate_mu);
    WalkGate& g = g_walk_gate[device];
    if (g.inflight > 0 && --g.inflight == 0 && g.fd >= 0) (void)flock(g.fd, LOCK_UN);
}
void gate_release_cb(void* p) { gate_release((int)(intptr_t)p); }

// guard 1: constructed before the launch, `launched(s)` records the order event; holds g_walk_mu in between
struct WalkTurn {
    std::unique_lock<std::mutex> lk;
    WalkOrder* o = nullptr;
    int begin(int device, hipStream_t s) {
        lk = std::unique_lock<std::mutex>(g_walk_mu);
        o = &g_walk_order[device];
        if (!o->done) MVDB_HIP(hipEventCreateWithFlags(&o->done, hipEventDisableTiming));
        if (o->any && o->last != s) MVDB_HIP(hipStreamWaitEvent(s, o->done, 0));
        return 0;
    }
    int launched(hipStream_t s) {
        MVDB_HIP(hipEventRecord(o->done, s));
        o->last = s;
        o->any = true;
        return 0;
    }
};

// ---- the layer-walking launch for small batches (encoder_walk.hpp) ------------------------------------------------------
// Eligible: at most kWalkSlots = 64 token slots (the kernel itself serves walk::kTmax = 128: row groups of 32) — beyond, the per-op
// kernels win since round 6 gave them split-K planes and 64 x 64 tiles for small batches: one sentence, host in / host out,
// e5-small shape 96 / 128 tokens 0.49 / 0.51 ms against the walking launch's 0.53 / 0.56 (64 tokens: 0.47 against 0.38); wide
// shapes 96 / 128 tokens 1.57 / 1.83 against 2.21 / 2.38 (profiles/r06_long_sentence_chain.txt) —, widths the column units tile
// (H, F multiples of 16, H <= 1024).
constexpr int kWalkSlots = 64;
int walk_max_slots(const mvdb_encoder_cfg&) { return kWalkSlots; }
bool walk_eligible(const mvdb_encoder* e, int B, int S) {
    const mvdb_encoder_cfg& c = e->cfg;
    return e->opt_walk && (int64_t)B * S <= walk_max_slots(c) && c.hidden % 16 == 0 && c.intermediate % 16 == 0 && c.hidden <= 1024;
}

int ensure_walk(mvdb_encoder* e) {
    if (e->walk_layers) return 0;
    const mvdb_encoder_cfg& c = e->cfg;
    const int64_t H = c.hidden, F = c.intermediate;
    const int cus = device_cus(e->device);
    e->walk_np3 = (int)std::min<int64_t>(std::min<int64_t>(F / 16, walk::kMaxPlanes), cus);
    {
        const char* v = getenv("MVDB_WALK_PLANES");  // A/B: workgroups (= partial planes) of the FFN phase
        if (v && *v) e->walk_np3 = std::max(1, std::min(e->walk_np3, atoi(v)));
    }
    {
        const char* v = getenv("MVDB_WALK_GRID");    // A/B: workgroups of the launch (0: by shape, launch_walk)
        e->walk_grid_env = v && *v ? atoi(v) : 0;
    }
    std::vector<walk::LayerPtrs> lp;
    for (const LayerW& L : e->layers)
        lp.push_back(walk::LayerPtrs{L.wqkv, L.bqkv, L.wo, L.bo, L.ln1g, L.ln1b, L.w1, L.b1, L.w2, L.b2, L.ln2g, L.ln2b});
    const int64_t planes = std::max<int64_t>({(int64_t)c.heads, (int64_t)e->walk_np3, 4, (F / 16 + 63) / 64});
    MVDB_TRY(dev_alloc(&e->walk_h, walk::kTmax * F));
    MVDB_TRY(dev_alloc(&e->walk_x, walk::kTmax * H));
    MVDB_TRY(dev_alloc(&e->walk_x1, walk::kTmax * H));
    MVDB_TRY(dev_alloc(&e->walk_qkv, walk::kTmax * 3 * H));
    MVDB_TRY(dev_alloc(&e->walk_pl, planes * walk::kTmax * H));
    MVDB_TRY(dev_alloc(&e->walk_bar, walk::kCtrCount * walk::kReplicas * walk::kCtrStride));
    MVDB_HIP(hipMemset(e->walk_bar, 0, walk::kCtrCount * walk::kReplicas * walk::kCtrStride * sizeof(unsigned int)));
    MVDB_TRY(dev_alloc(&e->walk_aborts, 1));
    MVDB_HIP(hipMemset(e->walk_aborts, 0, sizeof(unsigned int)));
    MVDB_HIP(hipHostMalloc((void**)&e->walk_aborts_host, sizeof(unsigned int), hipHostMallocMapped));
    *e->walk_aborts_host = 0u;
    {
        // no wait of a launch outlasts this (default 20 ms >> the 0.25 - 2.4 ms of a forward, << anything a watchdog would notice)
        const char* v = getenv("MVDB_WALK_DEADLINE_US");
        const long long us = v && *v ? atoll(v) : 20000;
        e->walk_deadline = (unsigned int)std::min<long long>(std::max<long long>(us, 0) * 100, 0x7fffffffLL);  // s_memrealtime: 100 MHz
    }
    walk::LayerPtrs* dev = nullptr;
    MVDB_TRY(dev_alloc(&dev, (int64_t)lp.size()));
    MVDB_HIP(hipMemcpy(dev, lp.data(), lp.size() * sizeof(walk::LayerPtrs), hipMemcpyHostToDevice));
    e->walk_layers = dev;
    return 0;
}

template <int MT, int HC, int RH>
int launch_walk_inst(mvdb_encoder* e, const walk::Args& a, size_t lds, int grid, hipStream_t s) {
    auto kern = walk::encoder_walk_kernel<MT, HC, RH>;
    MVDB_TRY(x3_set_lds((const void*)kern, (int)lds, e->device));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(walk::kThreads), lds, s, a);
    MVDB_HIP(hipGetLastError());
    return 0;
}

int launch_walk(mvdb_encoder* e, const int32_t* ids, const int32_t* mask, int B, int S, float* out, float* hidden, hipStream_t s) {
    MVDB_TRY(ensure_walk(e));
    const mvdb_encoder_cfg& c = e->cfg;
    walk::Args a;
    a.ids = ids;
    a.mask = mask;
    a.B = B;
    a.S = S;
    a.H = c.hidden;
    a.F = c.intermediate;
    a.heads = c.heads;
    a.hd = c.hidden / c.heads;
    a.nlayers = c.layers;
    a.position_offset = c.position_offset;
    a.vocab = c.vocab_size;
    a.pooling = c.pooling;
    a.eps = c.ln_eps;
    a.word = e->word;
    a.pos = e->pos;
    a.type = e->type;
    a.embg = e->embg;
    a.embb = e->embb;
    a.layers = e->walk_layers;
    a.X = e->walk_x;
    a.X1 = e->walk_x1;
    a.QKV = e->walk_qkv;
    a.PL = e->walk_pl;
    a.Hb = e->walk_h;
    a.bar = e->walk_bar;
    a.flag = e->overflow_flag;
    a.out = out;
    a.hidden = hidden;
    a.np3 = e->walk_np3;
    a.deadline = e->walk_deadline;
    a.aborts = e->walk_aborts;
    MVDB_HIP(hipHostGetDevicePointer((void**)&a.aborts_host, e->walk_aborts_host, 0));
    const int ntiles = c.hidden / 16;
    const int slots = B * S;
    // <= 16 / <= 32 token slots: one / two row tiles per column unit; more: two tiles and the units split by row groups of 32 too
    const int hc = c.hidden <= 128 ? 1 : c.hidden <= 384 ? 3 : 8;
    int mt = slots <= 16 ? 1 : 2, rh = slots <= 32 ? 1 : slots <= 64 ? 2 : 4;
    // wide shapes: a column unit's weights are 64 KB — four row tiles per unit (its weights fetched once for all 64 rows)
    // instead of two units of two (one sentence of 64 tokens on the large shape: 1.59 -> 1.1 ms)
    if (hc == 8 && slots > 32) {
        mt = 4;
        rh = slots <= 64 ? 1 : 2;
    }
    const int cus = device_cus(e->device);
    int grid = std::min(cus, std::max<int>({16, 3 * c.hidden / 16 * rh, e->walk_np3 * rh}));  // one workgroup per CU: all resident
    if (hc == 8) grid = cus;  // wide shapes: FFN1 has F / 16 column units, FFN2 (H / 16) x 4 (encoder_walk.hpp): every CU
    if (e->walk_grid_env > 0) grid = std::min(cus, std::max(e->walk_grid_env, e->walk_np3));
    e->walk_grid = grid;
    a.nsplit = std::max(1, std::min(std::max(grid, 96) / std::max(1, B * c.heads * rh), std::max(1, ntiles / walk::kWaves)));
    // Role placement (encoder_walk.hpp "ROLES"): rows | QKV | attention side by side, the FFN over the QKV / attention workgroups
    // (not adjacent phases) — every producer / consumer pair of adjacent phases on disjoint workgroups where the CUs allow:
    // e5-small shape, <= 32 slots: 32 + 72 + 36 = 140 workgroups; 64 slots: 64 + 144 + 72 wraps at 256 (24 attention
    // units land on row-owning workgroups).  Wide shapes use every CU in every GEMM phase: no separation to be had.
    a.off_rows = a.off_qkv = a.off_attn = a.off_ffn = 0;
    // (A/B on the host clock, `profiles/r05_walk_roles_ab.txt`: 5 - 7 us per forward at <= 32 slots; at 64 slots the grid would
    //  grow to every CU and the forward LOSES 15 us: roles are placed up to 32 slots only)
    if (hc != 8 && rh == 1 && !e->walk_grid_env && e->opt_walk_roles) {
        const int r = std::min(slots, cus), q = 3 * c.hidden / 16 * rh, at = B * c.heads * a.nsplit * rh, f = e->walk_np3 * rh;
        grid = std::min(cus, std::max({16, r + q + at, r + f}));
        a.off_qkv = r % grid;
        a.off_attn = (r + q) % grid;
        a.off_ffn = r % grid;
        e->walk_grid = grid;
    }
    a.trace = nullptr;
#ifdef MVDB_X3_ABLATE
    if (!e->walk_trace) MVDB_TRY(dev_alloc(&e->walk_trace, (int64_t)cus * walk::kTraceSlots));
    MVDB_HIP(hipMemsetAsync(e->walk_trace, 0, sizeof(unsigned long long) * cus * walk::kTraceSlots, s));
    a.trace = e->walk_trace;
#endif
    const size_t lds = walk::lds_bytes(mt, rh, c.hidden, a.hd);
#define MVDB_WALK_CASE(M, C, R) if (mt == M && hc == C && rh == R) return launch_walk_inst<M, C, R>(e, a, lds, grid, s)
    MVDB_WALK_CASE(1, 1, 1); MVDB_WALK_CASE(2, 1, 1); MVDB_WALK_CASE(2, 1, 2);
    MVDB_WALK_CASE(1, 3, 1); MVDB_WALK_CASE(2, 3, 1); MVDB_WALK_CASE(2, 3, 2);
    MVDB_WALK_CASE(1, 8, 1); MVDB_WALK_CASE(2, 8, 1); MVDB_WALK_CASE(4, 8, 1);
#undef MVDB_WALK_CASE
    return fail(MVDB_ERR_ARG, "no walker instantiation for this shape");
}

// Enqueue every kernel of one forward on `s` (no allocation, no host sync: capturable in a hipGraph).
int enqueue_lane(mvdb_encoder* e, mvdb_encoder::Lane& w, const int32_t* ids, const int32_t* mask, int B, int S,
                 int compute, float* out, float* hidden, hipStream_t s, bool clear_flag = false) {
    const mvdb_encoder_cfg& c = e->cfg;
    const int H = c.hidden, F = c.intermediate, hd = H / c.heads;
    const int64_t Tmax = (int64_t)B * S;
    const int vpt = (H + 63) / 64;
    const int* Tptr = w.seq_start + B;

    if (clear_flag && B <= kPackSmallB) {
        hipLaunchKernelGGL(pack_small_kernel, dim3(1), dim3(256), 0, s, mask, ids, B, S, c.position_offset, c.vocab_size, w.rank, w.count,
                           w.seq_start, w.tok_id, w.tok_pos, w.tok_src, e->overflow_flag);
    } else {
        if (clear_flag) MVDB_HIP(hipMemsetAsync(e->overflow_flag, 0, sizeof(unsigned int), s));  // (a memset node of the captured graph)
        hipLaunchKernelGGL(seq_rank_kernel, dim3(B), dim3(64), 0, s, mask, S, w.rank, w.count);
        hipLaunchKernelGGL(seq_scan_kernel, dim3(1), dim3(256), 0, s, w.count, B, w.seq_start);
        hipLaunchKernelGGL(pack_fill_kernel, dim3(B), dim3(256), 0, s, ids, w.rank, w.seq_start, S,
                           c.position_offset, c.vocab_size, w.tok_id, w.tok_pos, w.tok_src);
    }
    const dim3 rowgrid((unsigned)((Tmax + 3) / 4));
    // compute = 2: every GEMM input is a (hi | lo) fp16 image written by its producer — x by the LayerNorms (beside the
    // fp32 x the residuals and the pooling read), the context by the attention kernel, the GELU output by FFN1's epilogue
    float* xp = compute == 2 ? w.xp : nullptr;
#define MVDB_VPT_SWITCH(CALL)                                      \
    switch (vpt) {                                                 \
        case 1: CALL(1); break;  case 2: CALL(2); break;           \
        case 3: CALL(3); break;  case 4: CALL(4); break;           \
        case 5: CALL(5); break;  case 6: CALL(6); break;           \
        case 7: CALL(7); break;  case 8: CALL(8); break;           \
        case 9: CALL(9); break;  case 10: CALL(10); break;         \
        case 11: CALL(11); break; case 12: CALL(12); break;        \
        case 13: CALL(13); break; case 14: CALL(14); break;        \
        case 15: CALL(15); break; default: CALL(16); break;        \
    }
#define EMBED_CALL(V)                                                                                  \
    hipLaunchKernelGGL(embed_ln_kernel<V>, rowgrid, dim3(256), 0, s, w.tok_id, w.tok_pos, w.seq_start, \
                       B, e->word, e->pos, e->type, e->embg, e->embb, c.ln_eps, H, w.x, xp)
    MVDB_VPT_SWITCH(EMBED_CALL)
#undef EMBED_CALL

    const float scale = 1.0f / sqrtf((float)hd);
    const int cus = device_cus(e->device);
    static const bool attn_valu = []() {
        const char* v = getenv("MVDB_ENCODER_ATTENTION");
        return v && v[0] == 'v';
    }();
    static const bool x3_attention = []() {
        const char* v = getenv("MVDB_ATTENTION_X3");
        return !(v && *v == '0');
    }();
    // sequences longer than this run 256-query workgroups (eight waves): each K / V tile is loaded and split for twice as
    // many queries — e5-small, B = 256: S = 512 28.9 -> 27.7 ms per forward (ragged 19.4 -> 18.6), S = 256 13.4 -> 13.0; sixteen
    // waves (one workgroup per (sentence, head) at S = 512) add nothing: 27.4 vs 27.3, ragged 18.6 vs 18.3
    static const int x3_wide_from = []() {
        const char* v = getenv("MVDB_ATTENTION_X3_WIDE_FROM");
        return v && *v ? atoi(v) : 128;
    }();
    static const bool x3_short = []() {
        const char* v = getenv("MVDB_ATTENTION_X3_SHORT");
        return !(v && *v == '0');
    }();

    const dim3 agrid((S + ATT_Q - 1) / ATT_Q, c.heads, B);
    const bool ctx_is_image = compute == 2 && x3_attention && !attn_valu;
    // Q / K / V as (hi | lo) images straight from the QKV GEMM's epilogue (Q pre-scaled), attention_x3i_kernel on them:
    // MVDB_ATTENTION_IMG=0 keeps the fp32 qkv + attention_x3_kernel (splits K / V per workgroup) as the A/B reference
    const bool img_attn = ctx_is_image && e->opt_img_attn;  // MVDB_ATTENTION_IMG as read when the encoder was created
    // bias + residual + LayerNorm in the epilogue of the N = H GEMMs (MVDB_GEMM_LN_FUSED=0: separate ln_kernel launches;
    // =2: at every batch size).  A row-owning workgroup streams ALL of W and there is one workgroup per CU, so the fused
    // kernel needs rows: measured on e5-small, B = 256 (full / ragged batch, ms per forward, fused vs GEMM + ln_kernel):
    // S = 32 1.97 / 1.54 vs 1.98 / 1.34, S = 64 3.24 / 2.43 vs 3.36 / 2.20, S = 128 5.71 / 4.23 vs 6.11 / 4.08,
    // S = 256 11.85 / 8.50 vs 12.73 / 8.30, S = 512 25.6 / 17.0 vs 27.1 / 18.1 — ahead on full batches from S = 64, on
    // ragged ones (64 % of the token slots filled; the row count per workgroup is chosen from the PADDED count, the only
    // one the host knows) only at S = 512.  Default: from 128 token slots per CU.
    const int ln_env = e->opt_ln_fused;  // MVDB_GEMM_LN_FUSED as read when the encoder was created
    const bool ln_fused = compute == 2 && x3_ln_fusable(H) && ln_env != 0 && (ln_env == 2 || Tmax >= 128 * (int64_t)cus);
    const int ffn2_parts = compute == 2 && !ln_fused ? x3_splitk_parts(Tmax, H, F, cus) : 0;
    const int wo_parts = compute == 2 && !ln_fused ? x3_splitk_parts(Tmax, H, H, cus) : 0;
    // QKV and FFN1 the same way on the wide shapes (K = H >= 768: 32 K-steps), their epilogues as a launch of their own
    const int qkv_parts = compute == 2 ? x3_splitk_parts_wide(Tmax, 3 * H, H, cus) : 0;
    const int ffn1_parts = compute == 2 ? x3_splitk_parts_wide(Tmax, F, H, cus) : 0;
    for (const LayerW& L : e->layers) {
        if (compute == 2 && img_attn && qkv_parts) {   // small batch, K >= 768: split over K, then bias + query scale + image
            MVDB_TRY(launch_gemm_x3_splitk(xp, L.wqkv_p, L.wqkv_is, w.planes, Tptr, Tmax, 3 * H, H, qkv_parts, e->device, s));
            MVDB_TRY(launch_partials_image<EPI_BIAS_QKV>(w.planes, qkv_parts, L.bqkv, w.seq_start, B, Tmax, 3 * H, H, scale * kLog2e, w.qkv, s));
        } else if (compute == 2 && img_attn)
            MVDB_TRY(launch_gemm_x3<EPI_BIAS_QKV>(xp, L.wqkv_p, L.wqkv_is, L.bqkv, nullptr, w.qkv, Tptr, Tmax, 3 * H, H, e->device, s,
                                                  H, scale * kLog2e));
        else if (compute == 2)
            MVDB_TRY(launch_gemm_x3<EPI_BIAS>(xp, L.wqkv_p, L.wqkv_is, L.bqkv, nullptr, w.qkv, Tptr, Tmax, 3 * H, H, e->device, s));
        else
            launch_gemm<EPI_BIAS>(w.x, L.wqkv, L.bqkv, nullptr, w.qkv, Tptr, Tmax, 3 * H, H, cus, s);
        if (attn_valu) {  // MVDB_ENCODER_ATTENTION=valu: the thread-per-query VALU kernel (A/B reference)
            if (hd == 32)
                hipLaunchKernelGGL(attention_kernel<32>, agrid, dim3(ATT_Q), 0, s, w.qkv, w.seq_start, H, scale,
                                   w.ctx);
            else
                hipLaunchKernelGGL(attention_kernel<64>, agrid, dim3(ATT_Q), 0, s, w.qkv, w.seq_start, H, scale,
                                   w.ctx);
        } else if (img_attn) {
            if (S <= 32 && x3_short) {  // one wave per (sentence, head)
                const dim3 sgrid(1, c.heads, B);
                if (hd == 32)
                    hipLaunchKernelGGL((attention_x3i_kernel<32, 1>), sgrid, dim3(64), 0, s, w.qkv, w.seq_start, H, w.ctx);
                else
                    hipLaunchKernelGGL((attention_x3i_kernel<64, 1>), sgrid, dim3(64), 0, s, w.qkv, w.seq_start, H, w.ctx);
            } else if (S > x3_wide_from) {  // eight waves = 256 queries per workgroup share each K / V tile
                const dim3 wgrid((S + 255) / 256, c.heads, B);
                if (hd == 32)
                    hipLaunchKernelGGL((attention_x3i_kernel<32, 8>), wgrid, dim3(512), 0, s, w.qkv, w.seq_start, H, w.ctx);
                else
                    hipLaunchKernelGGL((attention_x3i_kernel<64, 8>), wgrid, dim3(512), 0, s, w.qkv, w.seq_start, H, w.ctx);
            } else if (hd == 32)
                hipLaunchKernelGGL((attention_x3i_kernel<32, 4>), agrid, dim3(256), 0, s, w.qkv, w.seq_start, H, w.ctx);
            else
                hipLaunchKernelGGL((attention_x3i_kernel<64, 4>), agrid, dim3(256), 0, s, w.qkv, w.seq_start, H, w.ctx);
        } else if (compute == 2 && x3_attention) {
            if (S <= 32 && x3_short) {  // one wave per (sentence, head)
                const dim3 sgrid(1, c.heads, B);
                if (hd == 32)
                    hipLaunchKernelGGL((attention_x3_kernel<32, 1>), sgrid, dim3(64), 0, s, w.qkv, w.seq_start, H, scale, w.ctx);
                else
                    hipLaunchKernelGGL((attention_x3_kernel<64, 1>), sgrid, dim3(64), 0, s, w.qkv, w.seq_start, H, scale, w.ctx);
            } else if (S > x3_wide_from) {  // eight waves = 256 queries per workgroup: each K / V tile is loaded and split for twice as many queries
                const dim3 wgrid((S + 255) / 256, c.heads, B);
                if (hd == 32)
                    hipLaunchKernelGGL((attention_x3_kernel<32, 8>), wgrid, dim3(512), 0, s, w.qkv, w.seq_start, H, scale, w.ctx);
                else
                    hipLaunchKernelGGL((attention_x3_kernel<64, 8>), wgrid, dim3(512), 0, s, w.qkv, w.seq_start, H, scale, w.ctx);
            } else if (hd == 32)
                hipLaunchKernelGGL(attention_x3_kernel<32>, agrid, dim3(256), 0, s, w.qkv, w.seq_start, H, scale, w.ctx);
            else
                hipLaunchKernelGGL(attention_x3_kernel<64>, agrid, dim3(256), 0, s, w.qkv, w.seq_start, H, scale, w.ctx);
        } else if (hd == 32) {
            hipLaunchKernelGGL(attention_mfma_kernel<32>, agrid, dim3(256), 0, s, w.qkv, w.seq_start, H, scale,
                               w.ctx);
        } else {
            hipLaunchKernelGGL(attention_mfma_kernel<64>, agrid, dim3(256), 0, s, w.qkv, w.seq_start, H, scale,
                               w.ctx);
        }
        if (compute == 2) {
            const float* ctx_img = w.ctx;
            if (!ctx_is_image) {  // an fp32 attention kernel ran (A/B switches): split its output into the image form
                const int64_t n = Tmax * H;
                hipLaunchKernelGGL(f32_split_interleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w.ctx,
                                   reinterpret_cast<_Float16*>(w.ffn), n, 1.0f);
                ctx_img = w.ffn;
            }
            if (ln_fused)
                MVDB_TRY(launch_gemm_x3_ln(ctx_img, L.wo_p, L.wo_is, L.bo, L.ln1g, L.ln1b, c.ln_eps, w.x, xp, Tptr, Tmax, H, H, e->device, s));
            else if (wo_parts)  // small batch, long K (H >= 768): split over K like FFN2 (attention is done with qkv)
                MVDB_TRY(launch_gemm_x3_splitk(ctx_img, L.wo_p, L.wo_is, w.planes, Tptr, Tmax, H, H, wo_parts, e->device, s));
            else
                MVDB_TRY(launch_gemm_x3<EPI_BIAS_RESIDUAL>(ctx_img, L.wo_p, L.wo_is, L.bo, w.x, w.y, Tptr, Tmax, H, H, e->device, s));
        }
        else
            launch_gemm<EPI_BIAS_RESIDUAL>(w.ctx, L.wo, L.bo, w.x, w.y, Tptr, Tmax, H, H, cus, s);
#define LN1_CALL(V) launch_ln<V>(w.y, w.seq_start, B, L.ln1g, L.ln1b, c.ln_eps, H, w.x, xp, Tmax, s)
#define LN1P_CALL(V) launch_ln_partials<V>(w.planes, wo_parts, Tmax * H, L.bo, w.seq_start, B, L.ln1g, L.ln1b, c.ln_eps, H, w.x, xp, Tmax, s)
        if (!ln_fused && wo_parts) { MVDB_VPT_SWITCH(LN1P_CALL) }
        else if (!ln_fused) { MVDB_VPT_SWITCH(LN1_CALL) }
#undef LN1P_CALL
#undef LN1_CALL
        if (compute == 2) {
            if (ffn1_parts) {
                MVDB_TRY(launch_gemm_x3_splitk(xp, L.w1_p, L.w1_is, w.planes, Tptr, Tmax, F, H, ffn1_parts, e->device, s));
                MVDB_TRY(launch_partials_image<EPI_BIAS_GELU>(w.planes, ffn1_parts, L.b1, w.seq_start, B, Tmax, F, 0, 1.f, w.ffn, s));
            } else
                MVDB_TRY(launch_gemm_x3<EPI_BIAS_GELU>(xp, L.w1_p, L.w1_is, L.b1, nullptr, w.ffn, Tptr, Tmax, F, H, e->device, s));
            if (ln_fused)
                MVDB_TRY(launch_gemm_x3_ln(w.ffn, L.w2_p, L.w2_is, L.b2, L.ln2g, L.ln2b, c.ln_eps, w.x, xp, Tptr, Tmax, H, F, e->device, s));
            else if (ffn2_parts)  // small batch: split over K into planes
                MVDB_TRY(launch_gemm_x3_splitk(w.ffn, L.w2_p, L.w2_is, w.planes, Tptr, Tmax, H, F, ffn2_parts, e->device, s));
            else
                MVDB_TRY(launch_gemm_x3<EPI_BIAS_RESIDUAL>(w.ffn, L.w2_p, L.w2_is, L.b2, w.x, w.y, Tptr, Tmax, H, F, e->device, s));
        } else {
            launch_gemm<EPI_BIAS_GELU>(w.x, L.w1, L.b1, nullptr, w.ffn, Tptr, Tmax, F, H, cus, s);
            launch_gemm<EPI_BIAS_RESIDUAL>(w.ffn, L.w2, L.b2, w.x, w.y, Tptr, Tmax, H, F, cus, s);
        }
#define LN2_CALL(V) launch_ln<V>(w.y, w.seq_start, B, L.ln2g, L.ln2b, c.ln_eps, H, w.x, xp, Tmax, s)
#define LN2P_CALL(V) launch_ln_partials<V>(w.planes, ffn2_parts, Tmax * H, L.b2, w.seq_start, B, L.ln2g, L.ln2b, c.ln_eps, H, w.x, xp, Tmax, s)
        if (!ln_fused && ffn2_parts) { MVDB_VPT_SWITCH(LN2P_CALL) }
        else if (!ln_fused) { MVDB_VPT_SWITCH(LN2_CALL) }
#undef LN2P_CALL
#undef LN2_CALL
    }
#undef MVDB_VPT_SWITCH
    // (w.y is free here: chunk partials of sentences longer than kPoolChunk tokens)
    hipLaunchKernelGGL(pool_norm_kernel, dim3(B, (S + kPoolChunk - 1) / kPoolChunk), dim3(256), 0, s, w.x, w.seq_start, H, c.pooling, out,
                       e->overflow_flag, w.y, w.pool_ctr);
    if (hidden)
        hipLaunchKernelGGL(unpack_hidden_kernel, dim3((unsigned)Tmax), dim3(256), 0, s, w.x, w.rank,
                           w.seq_start, S, H, hidden);
    MVDB_HIP(hipGetLastError());
    return 0;
}

// One forward: the whole batch on lane 0, or — exact mode, >= 64 sentences and >= 32768 token slots — two halves on
// two streams (fork / join by events, capturable into one hipGraph): the halves are independent, so one half's kernel
// tails and attention run beside the other's GEMMs.  Measured (B = 256): S = 512 63.3 vs 65.4 ms, ragged 40.6 vs
// 43.9 ms; at S = 32 (T = 8192) it gains nothing (4.09 vs 4.06 ms) and costs 6 % on a ragged batch: not used there.
int enqueue_forward(mvdb_encoder* e, const int32_t* ids, const int32_t* mask, int B, int S, int compute,
                    float* out, float* hidden, hipStream_t s) {
    static const int split_mode = []() {
        const char* v = getenv("MVDB_ENCODER_SPLIT");
        return v ? atoi(v) : 1;
    }();
    const bool split = split_mode && s && compute == 0 && B >= 64 && (int64_t)B * S >= 32768 && e->stream2;
    if (!split) return enqueue_lane(e, e->lane[0], ids, mask, B, S, compute, out, hidden, s, true);  // (clears the overflow word)
    MVDB_HIP(hipMemsetAsync(e->overflow_flag, 0, sizeof(unsigned int), s));  // (a memset node of the captured graph)
    const int b0 = B / 2, b1 = B - b0;
    const int H = e->cfg.hidden;
    MVDB_HIP(hipEventRecord(e->ev_fork, s));
    MVDB_HIP(hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
    MVDB_TRY(enqueue_lane(e, e->lane[0], ids, mask, b0, S, compute, out, hidden, s));
    MVDB_TRY(enqueue_lane(e, e->lane[1], ids + (int64_t)b0 * S, mask + (int64_t)b0 * S, b1, S, compute,
                          out + (int64_t)b0 * H, hidden ? hidden + (int64_t)b0 * S * H : nullptr, e->stream2));
    MVDB_HIP(hipEventRecord(e->ev_join, e->stream2));
    MVDB_HIP(hipStreamWaitEvent(s, e->ev_join, 0));
    return 0;
}

// walk_mode: 0 = per-op kernels whatever the shape; 1 = the walking launch where it applies, the CALLER waits for the stream
// and then releases the device's gate (when *walked comes back true); 2 = the same, the gate released by a host callback
// behind the launch (the device entry never waits).
int forward_core(mvdb_encoder* e, const int32_t* ids, const int32_t* mask, int B, int S, int compute,
                 float* out, float* hidden, hipStream_t s, int walk_mode = 0, bool* walked = nullptr) {
    if (walked) *walked = false;
    if (compute != 0 && compute != 2)
        return fail(MVDB_ERR_ARG, "unknown compute mode %d (0 = exact-fp32 MFMA, 2 = split-precision fp16 x 3; 1, the single-bf16-product "
                                  "mode of earlier builds, was removed: slower and less exact than 2)", compute);
    const mvdb_encoder_cfg& c = e->cfg;
    if (B <= 0 || S <= 0) return fail(MVDB_ERR_ARG, "B and S must be positive");
    if (S + (c.position_offset > 0 ? c.position_offset : 0) > c.max_positions)
        return fail(MVDB_ERR_ARG, "sequence length %d exceeds max_positions %d", S, c.max_positions);
    if (walk_mode && walk_eligible(e, B, S)) {
        // <= walk::kTmax token slots (one sentence per call is the reference's only shape): ONE launch walks the layers, exact
        // fp32 in both modes — unless the caller captures the stream (a captured launch can take no part in the ordering of
        // walking launches: per-op kernels, which are capturable), launches of this encoder were abandoned lately, or another
        // process holds the device's gate.
        hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
        const bool captured = s && hipStreamIsCapturing(s, &cap0) == hipSuccess && cap0 != hipStreamCaptureStatusNone;
        if (e->walk_aborts_host && *e->walk_aborts_host != e->walk_aborts_seen) {  // (mirrors of COMPLETED launches)
            e->walk_aborts_seen = *e->walk_aborts_host;
            e->walk_suspended = kWalkSuspendCalls;
        }
        if (e->walk_suspended > 0) --e->walk_suspended;
        else if (!captured && !gate_acquire(e->device)) e->walk_suspended = kGateBusyCalls;  // (do not pay kGateWaitUs on every call)
        else if (!captured) {
            const int pslot0 = prof_begin("encoder", s);
            WalkTurn turn;
            int rc0 = turn.begin(e->device, s);
            if (!rc0) rc0 = launch_walk(e, ids, mask, B, S, out, hidden, s);  // (clears the overflow flag itself: exact fp32)
            if (!rc0) rc0 = turn.launched(s);
            prof_end(pslot0, s);
            if (rc0 || walk_mode == 2) {
                if (rc0 || hipLaunchHostFunc(s, gate_release_cb, (void*)(intptr_t)e->device) != hipSuccess) {
                    (void)hipGetLastError();
                    if (!rc0) (void)hipStreamSynchronize(s);
                    gate_release(e->device);
                }
            } else if (walked) {
                *walked = true;
            } else {
                (void)hipStreamSynchronize(s);
                gate_release(e->device);
            }
            return rc0;
        }
    }
    if (compute == 2) {
        if (c.hidden % HBK || c.intermediate % HBK)
            return fail(MVDB_ERR_ARG, "the split-precision mode needs hidden and intermediate to be multiples of %d", HBK);
        MVDB_TRY(ensure_x3_weights(e, s));
    }
    const uint64_t gen_before = e->ws_gen;
    MVDB_TRY(ensure_ws(e, B, S));
    if (e->ws_gen != gen_before) e->drop_graphs();  // workspace moved: captured pointers are stale

    // ~90 short launches per forward: replay them as ONE hipGraph per (shape, buffers) instead of paying
    // the host launch path per kernel (the S = 32 forward is launch-bound otherwise)
    static const bool use_graph = []() {
        const char* v = getenv("MVDB_ENCODER_GRAPH");
        return !(v && *v == '0');
    }();
    // the caller is capturing `s` into a graph of its own (e.g. encoder -> search as ONE graph): plain launches join it
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool outer_capture = s && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    int pslot = outer_capture ? -1 : prof_begin("encoder", s);
    int rc = 0;
    if (use_graph && !hidden && s && !outer_capture) {
        const GraphKey key{B, S, compute, ids, mask, out};
        auto it = e->graphs.find(key);
        if (it == e->graphs.end()) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
                (void)hipGetLastError();  // stream not capturable: plain launches
                rc = enqueue_forward(e, ids, mask, B, S, compute, out, nullptr, s);
                prof_end(pslot, s);
                return rc;
            }
            rc = enqueue_forward(e, ids, mask, B, S, compute, out, nullptr, s);
            hipError_t ec = hipStreamEndCapture(s, &graph);
            if (!rc && ec != hipSuccess) rc = fail(MVDB_ERR_HIP, "graph capture failed: %s", hipGetErrorString(ec));
            if (!rc) {
                ec = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                if (ec != hipSuccess) rc = fail(MVDB_ERR_HIP, "graph instantiate failed: %s", hipGetErrorString(ec));
            }
            if (graph) (void)hipGraphDestroy(graph);
            if (rc) return rc;
            if (e->graphs.size() >= 64) e->drop_graphs();
            it = e->graphs.emplace(key, exec).first;
        }
        MVDB_HIP(hipGraphLaunch(it->second, s));
    } else {
        rc = enqueue_forward(e, ids, mask, B, S, compute, out, hidden, s);
    }
    prof_end(pslot, s);
    return rc;
}

}  // namespace

extern "C" {

#ifdef MVDB_X3_ABLATE
// ablation build only: copies the DBG == 5 timeline of the last traced GEMM launch (16 words per tile) and clears it
int mvdb_debug_x3_trace(unsigned long long* out, int nblocks) {
    if (!out || nblocks <= 0 || nblocks > kX3TraceBlocks) return fail(MVDB_ERR_ARG, "bad trace request");
    MVDB_HIP(hipDeviceSynchronize());
    MVDB_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x3_trace), (size_t)nblocks * kX3TraceWords * sizeof(unsigned long long)));
    void* sym = nullptr;
    MVDB_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_x3_trace)));
    MVDB_HIP(hipMemset(sym, 0, sizeof(unsigned long long) * kX3TraceWords * kX3TraceBlocks));
    return 0;
}

// ablation build only: the s_memrealtime stamps (100 MHz) of the last layer-walking launch, [workgroups][256]; returns the
// number of workgroups (or a negative error)
int mvdb_debug_walk_trace(mvdb_encoder* e, unsigned long long* out, int max_wg) {
    if (!e || !out || !e->walk_trace) return -1;
    (void)hipDeviceSynchronize();
    const int n = std::min(max_wg, e->walk_grid);
    if (hipMemcpy(out, e->walk_trace, sizeof(unsigned long long) * n * walk::kTraceSlots, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    return n;
}
#endif


int mvdb_encoder_weight_count(const mvdb_encoder_cfg* cfg) {
    if (check_cfg(cfg)) return -1;
    return kFixedWeights + kPerLayer * cfg->layers;
}

const char* mvdb_encoder_weight_name(const mvdb_encoder_cfg* cfg, int i) {
    static thread_local char buf[160];
    if (check_cfg(cfg)) return nullptr;
    if (i < 0 || i >= kFixedWeights + kPerLayer * cfg->layers) return nullptr;
    if (i < kFixedWeights) return kFixedNames[i];
    const int l = (i - kFixedWeights) / kPerLayer, j = (i - kFixedWeights) % kPerLayer;
    snprintf(buf, sizeof(buf), "encoder.layer.%d.%s", l, kLayerNames[j]);
    return buf;
}

int mvdb_encoder_create(const mvdb_encoder_cfg* cfg, const void* const* w, int device, mvdb_encoder** out) {
    if (!out) return fail(MVDB_ERR_ARG, "out is NULL");
    *out = nullptr;
    MVDB_TRY(check_cfg(cfg));
    if (!w) return fail(MVDB_ERR_ARG, "weight table is NULL");
    const int nw = kFixedWeights + kPerLayer * cfg->layers;
    for (int i = 0; i < nw; ++i)
        if (!w[i]) return fail(MVDB_ERR_ARG, "weight %d (%s) is NULL", i, mvdb_encoder_weight_name(cfg, i));
    MVDB_TRY(ensure_device(device));
    DeviceGuard dg(device);
    mvdb_encoder* e = new mvdb_encoder();
    e->cfg = *cfg;
    e->device = device;
    {
        const char* v = getenv("MVDB_GEMM_LN_FUSED");
        e->opt_ln_fused = v && *v ? atoi(v) : 1;
        v = getenv("MVDB_ATTENTION_IMG");
        e->opt_img_attn = !(v && *v == '0');
        v = getenv("MVDB_ENCODER_WALK");
        e->opt_walk = !(v && *v == '0');
        v = getenv("MVDB_WALK_ROLES");
        e->opt_walk_roles = !(v && *v == '0');
        v = getenv("MVDB_WALK_PINNED");
        e->opt_walk_pinned = !(v && *v == '0');
#ifdef MVDB_X3_ABLATE
        v = getenv("MVDB_LN_SKIP_X");
        const int skip = v && *v == '1';
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ln_skip_x), &skip, sizeof(int));
#endif
    }
    e->word = (const float*)w[0];
    e->pos = (const float*)w[1];
    e->type = (const float*)w[2];
    e->embg = (const float*)w[3];
    e->embb = (const float*)w[4];
    const int64_t H = cfg->hidden;
    int rc = 0;
    for (int l = 0; l < cfg->layers && !rc; ++l) {
        const void* const* p = w + kFixedWeights + (int64_t)l * kPerLayer;
        float *wqkv = nullptr, *bqkv = nullptr;
        if (hipMalloc((void**)&wqkv, 3 * H * H * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&bqkv, 3 * H * sizeof(float)) != hipSuccess) {
            rc = fail(MVDB_ERR_OOM, "device allocation for fused QKV weights failed");
            if (wqkv) (void)hipFree(wqkv);
            break;
        }
        e->owned.push_back(wqkv);
        e->owned.push_back(bqkv);
        hipLaunchKernelGGL(concat3_kernel, dim3((unsigned)((H * H + 255) / 256)), dim3(256), 0, nullptr,
                           (const float*)p[0], (const float*)p[2], (const float*)p[4], H * H, wqkv);
        hipLaunchKernelGGL(concat3_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, nullptr,
                           (const float*)p[1], (const float*)p[3], (const float*)p[5], H, bqkv);
        LayerW L;
        L.wqkv = wqkv;
        L.bqkv = bqkv;
        L.wo = (const float*)p[6];
        L.bo = (const float*)p[7];
        L.ln1g = (const float*)p[8];
        L.ln1b = (const float*)p[9];
        L.w1 = (const float*)p[10];
        L.b1 = (const float*)p[11];
        L.w2 = (const float*)p[12];
        L.b2 = (const float*)p[13];
        L.ln2g = (const float*)p[14];
        L.ln2b = (const float*)p[15];
        e->layers.push_back(L);
    }
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(MVDB_ERR_HIP, "weight fusion failed");
    if (!rc && (hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess))
        rc = fail(MVDB_ERR_HIP, "stream / event creation for the split forward failed");
    if (!rc && hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(MVDB_ERR_HIP, "stream creation failed");
    if (!rc && (hipMalloc((void**)&e->overflow_flag, sizeof(unsigned int)) != hipSuccess ||
                hipMemset(e->overflow_flag, 0, sizeof(unsigned int)) != hipSuccess))
        rc = fail(MVDB_ERR_OOM, "device allocation for the overflow flag failed");
    if (rc) {
        mvdb_encoder_free(e);
        return rc;
    }
    *out = e;
    return 0;
}

const unsigned int* mvdb_encoder_overflow_flag(const mvdb_encoder* e) { return e ? e->overflow_flag : nullptr; }

int mvdb_encoder_walk_stats(const mvdb_encoder* e, unsigned long long* aborts, unsigned long long* fallbacks, int* suspended_calls) {
    if (!e) return fail(MVDB_ERR_ARG, "encoder is NULL");
    if (aborts) *aborts = e->walk_aborts_host ? *e->walk_aborts_host : 0u;
    if (fallbacks) *fallbacks = e->walk_fallbacks;
    if (suspended_calls) *suspended_calls = e->walk_suspended;
    return 0;
}

int mvdb_encoder_walks(const mvdb_encoder* e, int B, int S) {
    return e && B > 0 && S > 0 && walk_eligible(e, B, S) ? 1 : 0;
}

int mvdb_encoder_splitk_planes(int64_t tokens, int n, int k, int compute_units) {
    if (tokens <= 0 || n <= 0 || k <= 0 || compute_units <= 0) return 0;
    return x3_splitk_parts(tokens, n, k, compute_units);
}

int mvdb_encoder_gemm_tile_form(int64_t tokens, int n, int compute_units) {
    if (tokens <= 0 || n <= 0 || compute_units <= 0) return 0;
    const int bn = n % 256 == 0 ? 256 : n % 192 == 0 ? 192 : 0;
    return bn != 0 && x3_big_form(tokens, n, bn, compute_units) ? bn : 0;
}

int mvdb_encoder_free(mvdb_encoder* e) {
    if (!e) return 0;
    {
        DeviceGuard dg(e->device);
        (void)hipDeviceSynchronize();
        e->drop_graphs();
        e->free_ws();
        for (float* p : e->owned) (void)hipFree(p);
        for (void* p : e->owned_h) (void)hipFree(p);
        if (e->ids_stage) (void)hipFree(e->ids_stage);
        if (e->mask_stage) (void)hipFree(e->mask_stage);
        if (e->out_stage) (void)hipFree(e->out_stage);
        void* walk_bufs[] = {e->walk_layers, e->walk_x, e->walk_x1, e->walk_qkv, e->walk_pl, e->walk_h, e->walk_bar, e->walk_trace, e->overflow_flag};
        for (void* p : walk_bufs)
            if (p) (void)hipFree(p);
        if (e->stream) (void)hipStreamDestroy(e->stream);
        if (e->stream2) (void)hipStreamDestroy(e->stream2);
        if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
        if (e->ev_join) (void)hipEventDestroy(e->ev_join);
        if (e->walk_done) (void)hipEventDestroy(e->walk_done);
        if (e->walk_aborts) (void)hipFree(e->walk_aborts);
        if (e->walk_aborts_host) (void)hipHostFree(e->walk_aborts_host);
        e->walk_pin.release();
    }
    delete e;
    return 0;
}

static int ensure_stage(mvdb_encoder* e, int64_t tokens, int64_t outn) {
    if (tokens > e->stage_cap) {
        e->drop_graphs();
        if (e->ids_stage) (void)hipFree(e->ids_stage);
        if (e->mask_stage) (void)hipFree(e->mask_stage);
        e->ids_stage = e->mask_stage = nullptr;
        e->stage_cap = 0;
        MVDB_HIP(hipMalloc((void**)&e->ids_stage, tokens * sizeof(int32_t)));
        MVDB_HIP(hipMalloc((void**)&e->mask_stage, tokens * sizeof(int32_t)));
        e->stage_cap = tokens;
    }
    if (outn > e->out_cap) {
        e->drop_graphs();
        if (e->out_stage) (void)hipFree(e->out_stage);
        e->out_stage = nullptr;
        e->out_cap = 0;
        MVDB_HIP(hipMalloc((void**)&e->out_stage, outn * sizeof(float)));
        e->out_cap = outn;
    }
    return 0;
}

int mvdb_encoder_forward_device(mvdb_encoder* e, const int32_t* ids_dev, const int32_t* mask_dev, int B,
                                int S, int compute, float* out_dev, float* hidden_dev, void* stream) {
    if (!e) return fail(MVDB_ERR_ARG, "encoder is NULL");
    if (!ids_dev || !mask_dev || !out_dev) return fail(MVDB_ERR_ARG, "NULL buffer");
    if (B <= 0 || S <= 0) return fail(MVDB_ERR_ARG, "B and S must be positive");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard dg(e->device);
    hipStream_t s = (hipStream_t)stream;
    if (hidden_dev || !s)  // diagnostic output, or the legacy NULL stream (not capturable): plain launches
        return forward_core(e, ids_dev, mask_dev, B, S, compute, out_dev, hidden_dev, s, 2);
    // Route through the encoder's own staging buffers so that the captured graph (keyed by buffer
    // addresses) is reused whatever tensors the caller passes: three small device-to-device copies.
    const int64_t tokens = (int64_t)B * S, outn = (int64_t)B * e->cfg.hidden;
    MVDB_TRY(ensure_stage(e, tokens, outn));
    // One forward at a time per encoder ON THE DEVICE too, whatever streams the callers use (the host lock above only orders
    // the enqueues): an event behind every forward, waited for by the next.  Not while the caller captures the stream.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (!capturing) {
        if (!e->walk_done) MVDB_HIP(hipEventCreateWithFlags(&e->walk_done, hipEventDisableTiming));
        else MVDB_HIP(hipStreamWaitEvent(s, e->walk_done, 0));
    }
    MVDB_HIP(hipMemcpyAsync(e->ids_stage, ids_dev, tokens * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    MVDB_HIP(hipMemcpyAsync(e->mask_stage, mask_dev, tokens * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    MVDB_TRY(forward_core(e, e->ids_stage, e->mask_stage, B, S, compute, e->out_stage, nullptr, s, 2));
    MVDB_HIP(hipMemcpyAsync(out_dev, e->out_stage, outn * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (!capturing) MVDB_HIP(hipEventRecord(e->walk_done, s));
    return 0;
}

int mvdb_encoder_forward(mvdb_encoder* e, const int32_t* ids_host, const int32_t* mask_host, int B, int S,
                         int compute, float* out_host) {
    if (!e) return fail(MVDB_ERR_ARG, "encoder is NULL");
    if (!ids_host || !mask_host || !out_host) return fail(MVDB_ERR_ARG, "NULL buffer");
    if (B <= 0 || S <= 0) return fail(MVDB_ERR_ARG, "B and S must be positive");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard dg(e->device);
    const int64_t tokens = (int64_t)B * S;
    const int64_t outn = (int64_t)B * e->cfg.hidden;
    for (int64_t i = 0; i < tokens; ++i)
        if (mask_host[i] && (ids_host[i] < 0 || ids_host[i] >= e->cfg.vocab_size))
            return fail(MVDB_ERR_ARG, "token id %d at %lld outside the vocabulary [0,%d)", ids_host[i],
                        (long long)i, e->cfg.vocab_size);
    // A walking launch that was abandoned (bounded waits, encoder_walk.hpp) has counted itself in *walk_aborts_host by the
    // time the stream is idle: the forward runs again, on the per-op kernels, within this call.
    auto abandoned = [&](unsigned int before) {
        if (!e->walk_aborts_host || *e->walk_aborts_host == before) return false;
        ++e->walk_fallbacks;
        return true;
    };
    const unsigned int aborts_before = e->walk_aborts_host ? *e->walk_aborts_host : 0u;
    bool walked = false;
    if (tokens <= kPinSlots && e->opt_walk_pinned) {
        // ONE sentence per call (extract_embeddings truncates at 512 tokens, embedding_model.py:64,77): the kernels read the ids
        // and the mask from host-mapped memory and write the embedding there — no copy engine work at all around the forward
        // (two H2D copies, a memset and a D2H copy before: ~20 us of a 0.27 ms call).  The walking launch up to its 128 token
        // slots, the per-op kernels beyond.
        const size_t in_bytes = 2 * (size_t)kPinSlots * sizeof(int32_t);
        MVDB_TRY(e->walk_pin.reserve(in_bytes + (size_t)kPinSlots * e->cfg.hidden * sizeof(float)));
        int32_t* pin_ids = reinterpret_cast<int32_t*>(e->walk_pin.p);
        int32_t* pin_mask = pin_ids + kPinSlots;
        float* pin_out = reinterpret_cast<float*>(reinterpret_cast<char*>(e->walk_pin.p) + in_bytes);
        memcpy(pin_ids, ids_host, tokens * sizeof(int32_t));
        memcpy(pin_mask, mask_host, tokens * sizeof(int32_t));
        if (e->walk_done) MVDB_HIP(hipStreamWaitEvent(e->stream, e->walk_done, 0));
        int rc = forward_core(e, pin_ids, pin_mask, B, S, compute, pin_out, nullptr, e->stream, 1, &walked);
        if (!rc && hipStreamSynchronize(e->stream) != hipSuccess) rc = fail(MVDB_ERR_HIP, "encoder forward failed: %s", hipGetErrorString(hipGetLastError()));
        if (walked) gate_release(e->device);
        if (rc) return rc;
        if (walked && abandoned(aborts_before)) {  // (the per-op kernels read the same host-mapped buffers)
            MVDB_TRY(forward_core(e, pin_ids, pin_mask, B, S, compute, pin_out, nullptr, e->stream, 0));
            MVDB_HIP(hipStreamSynchronize(e->stream));
        }
        memcpy(out_host, pin_out, outn * sizeof(float));
        return 0;
    }
    MVDB_TRY(ensure_stage(e, tokens, outn));
    if (e->walk_done) MVDB_HIP(hipStreamWaitEvent(e->stream, e->walk_done, 0));  // a device-entry forward may still be running
    MVDB_HIP(hipMemcpyAsync(e->ids_stage, ids_host, tokens * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    MVDB_HIP(hipMemcpyAsync(e->mask_stage, mask_host, tokens * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc = forward_core(e, e->ids_stage, e->mask_stage, B, S, compute, e->out_stage, nullptr, e->stream, attempt == 0 ? 1 : 0, &walked);
        if (!rc && hipMemcpyAsync(out_host, e->out_stage, outn * sizeof(float), hipMemcpyDeviceToHost, e->stream) != hipSuccess)
            rc = fail(MVDB_ERR_HIP, "copy of the embeddings failed: %s", hipGetErrorString(hipGetLastError()));
        if (!rc && hipStreamSynchronize(e->stream) != hipSuccess) rc = fail(MVDB_ERR_HIP, "encoder forward failed: %s", hipGetErrorString(hipGetLastError()));
        if (walked) gate_release(e->device);
        if (rc) return rc;
        if (!(walked && abandoned(aborts_before))) break;
    }
    return 0;
}

}  // extern "C"
