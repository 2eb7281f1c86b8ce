// load_issue_cost.hip — what a wave pays to ISSUE one 1-KiB vector load, by kind:
//   global_load_lds_dwordx4 (LDS-DMA, 16 B per lane straight into LDS)  vs  global_load_dwordx4 (into four VGPRs).
// Every wave issues N loads back to back from a buffer that fits the L2, stamps s_memtime before the first and after the
// last ISSUE (no s_waitcnt in between), then waits for all of them.  waves per SIMD 1 or 2; optional MFMAs between loads.
// build: hipcc --offload-arch=gfx950 -O2 load_issue_cost.hip -o load_issue_cost ; run on a GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

template <int KIND, int N, int MF>
__global__ void probe(const float* __restrict__ src, unsigned long long* __restrict__ out, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = reinterpret_cast<const char*>(src) + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * (size_t)N * 1024 + lane * 16;
    f4 r[KIND == 1 ? N : 1];
    f16v acc = {0};
    h8 a = {(_Float16)1.f, 0, 0, 0, 0, 0, 0, 0}, b = a;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (KIND == 0)
            __builtin_amdgcn_global_load_lds((gbl_ptr)(base + i * 1024), (lds_ptr)(lds + wave * N * 1024 + i * 1024), 16, 0, 0);
        else
            r[i] = *reinterpret_cast<const f4*>(base + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MF; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    float s = acc[0];
    if (KIND == 1)
#pragma unroll
        for (int i = 0; i < N; ++i) s += r[i][0];
    else
        s += reinterpret_cast<float*>(lds)[threadIdx.x];
    if (s == 1.2345e-30f) sink[0] = s;
    if (lane == 0) {
        out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2] = t1 - t0;
        out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = t2 - t0;
    }
}

template <int KIND, int N, int MF>
void run(const char* name, int waves_per_block, const float* src, unsigned long long* out, float* sink) {
    const int blocks = 256;
    std::vector<unsigned long long> h(blocks * waves_per_block * 2);
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL((probe<KIND, N, MF>), dim3(blocks), dim3(waves_per_block * 64), waves_per_block * N * 1024, 0, src, out, sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> iss, tot;
    for (size_t i = 0; i < h.size() / 2; ++i) {
        iss.push_back((double)h[2 * i] / N);
        tot.push_back((double)h[2 * i + 1]);
    }
    std::sort(iss.begin(), iss.end());
    std::sort(tot.begin(), tot.end());
    printf("%-28s waves/block %d  MFMAs between loads %d: issue %.1f cycles per load (median; p90 %.1f), all landed after %.0f cycles\n", name,
           waves_per_block, MF, iss[iss.size() / 2], iss[iss.size() * 9 / 10], tot[tot.size() / 2]);
}

int main() {
    float *src, *sink;
    unsigned long long* out;
    const size_t bytes = (size_t)256 * 8 * 16 * 1024;
    hipMalloc(&src, bytes);
    hipMemset(src, 0, bytes);
    hipMalloc(&sink, 64);
    hipMalloc(&out, 256 * 8 * 2 * 8);
    for (int w : {4, 8}) {
        if (w == 4) {
            run<0, 8, 0>("global_load_lds_dwordx4", 4, src, out, sink);
            run<1, 8, 0>("global_load_dwordx4", 4, src, out, sink);
            run<0, 8, 2>("global_load_lds_dwordx4", 4, src, out, sink);
            run<1, 8, 2>("global_load_dwordx4", 4, src, out, sink);
        } else {
            run<0, 8, 0>("global_load_lds_dwordx4", 8, src, out, sink);
            run<1, 8, 0>("global_load_dwordx4", 8, src, out, sink);
            run<0, 8, 2>("global_load_lds_dwordx4", 8, src, out, sink);
            run<1, 8, 2>("global_load_dwordx4", 8, src, out, sink);
        }
    }
    return 0;
}
