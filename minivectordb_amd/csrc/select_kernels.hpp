// select_kernels.hpp — exact top-k for k > 64 from a materialised score vector.
//
// The reference clamps search_k to min(k, rows) (minivectordb/vector_database.py:489-492) and its
// tests go to k = 999 / 825 (tests/test_mongolike_operators.py:43,
// tests/test_sharded_multithreaded_operations.py:65-68), so large k must work; it is not the
// bandwidth path (the score vector is 4 B per row against ld*4 B per row of corpus).
//
//   1. flat_scan_kernel<MODE = kModeScores> writes scores[n]
//   2. 8 x (radix_hist_kernel, radix_pick_kernel): MSB-first 8-bit radix select of the k-th
//      largest 64-bit key (score image << 32 | ~row) — keys are unique, so the answer is exact
//      and ties resolve to the lower row number
//   3. radix_compact_kernel gathers the exactly-k keys >= the pivot
//   4. bitonic sort (descending) of the k keys, emit (D, I)
#pragma once
#include "topk_device.hpp"

namespace mvdb {

struct SelectState {
    uint64_t prefix;     // digits decided so far (high bits)
    uint64_t mask;       // which bits of prefix are decided
    uint64_t k_rem;      // rank still to resolve inside the prefix bucket
    uint32_t hist[256];  // current pass histogram
    uint32_t out_count;  // compaction cursor
    uint32_t pad;
};

__global__ void select_init_kernel(SelectState* st, uint64_t k) {
    if (threadIdx.x == 0) {
        st->prefix = 0;
        st->mask = 0;
        st->k_rem = k;
        st->out_count = 0;
    }
    if (threadIdx.x < 256) st->hist[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void radix_hist_kernel(const float* __restrict__ scores,
                                                         int64_t n, int shift, SelectState* st) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t prefix = st->prefix, mask = st->mask;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = make_key(scores[i], (uint32_t)i);
        if ((key & mask) == prefix) atomicAdd(&h[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}

// one block of 256: pick the digit holding the k_rem-th largest key of the bucket
__global__ __launch_bounds__(256) void radix_pick_kernel(int shift, SelectState* st) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = st->hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t k_rem = st->k_rem, above = 0;
        int dsel = 0;
        for (int dgt = 255; dgt >= 0; --dgt) {
            if (above + h[dgt] >= k_rem) {
                dsel = dgt;
                break;
            }
            above += h[dgt];
        }
        st->k_rem = k_rem - above;
        st->prefix |= (uint64_t)dsel << shift;
        st->mask |= (uint64_t)255 << shift;
    }
    __syncthreads();
    st->hist[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void radix_compact_kernel(const float* __restrict__ scores,
                                                            int64_t n, SelectState* st,
                                                            uint64_t* __restrict__ out) {
    const uint64_t pivot = st->prefix;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = make_key(scores[i], (uint32_t)i);
        if (key >= pivot) out[atomicAdd(&st->out_count, 1u)] = key;
    }
}

// keys[P] (P power of two, tail zero padded) sorted descending.
// LDS version: one block, P <= 4096.
__global__ __launch_bounds__(1024) void bitonic_sort_lds_kernel(uint64_t* keys, int P) {
    __shared__ uint64_t s[4096];
    for (int i = threadIdx.x; i < P; i += blockDim.x) s[i] = keys[i];
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int tix = threadIdx.x; tix < P / 2; tix += blockDim.x) {
                const int lo = 2 * tix - (tix & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = s[lo], b = s[hi];
                if ((a < b) == desc) {
                    s[lo] = b;
                    s[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < P; i += blockDim.x) keys[i] = s[i];
}

// Global-memory version: one launch per (size, stride).
__global__ __launch_bounds__(256) void bitonic_step_kernel(uint64_t* keys, int64_t P, int64_t size,
                                                           int64_t stride) {
    const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tix >= P / 2) return;
    const int64_t lo = 2 * tix - (tix & (stride - 1));
    const int64_t hi = lo + stride;
    const bool desc = (lo & size) == 0;
    const uint64_t a = keys[lo], b = keys[hi];
    if ((a < b) == desc) {
        keys[lo] = b;
        keys[hi] = a;
    }
}

__global__ void zero_tail_kernel(uint64_t* keys, int64_t from, int64_t to) {
    const int64_t i = from + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < to) keys[i] = 0;
}

// drop_unselected: bitmap-selected search — rows outside the mask were written with score -inf and are not results
__global__ void emit_sorted_kernel(const uint64_t* __restrict__ keys, int k, int metric,
                                   int64_t label_offset, float* __restrict__ D,
                                   int64_t* __restrict__ I, int drop_unselected) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const uint64_t key = keys[i];
    if (key && !(drop_unselected && key_score(key) == -INFINITY)) {
        const float s = key_score(key);
        D[i] = metric == 0 ? s : -s;
        I[i] = label_offset + (int64_t)key_row(key);
    } else {
        D[i] = metric == 0 ? -3.402823466e+38f : 3.402823466e+38f;
        I[i] = -1;
    }
}

}  // namespace mvdb
