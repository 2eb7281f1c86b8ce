// stream_read.hip — ceiling probe: how fast can ANY kernel read 20.48 GB from HBM on this part?
// A bare grid-stride float4 reduction (non-temporal loads, nothing else) swept over unroll depth and
// resident blocks per CU.  The flat-scan kernel's 7.2 TB/s is judged against the best of these.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_read stream_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ x, size_t n4, float* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += x[i];
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;  // keep the loads alive
}

template <int U, bool NT>
double run(const f32x4* x, size_t n4, float* out, int blocks, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((read_kernel<U, NT>), dim3(blocks), dim3(256), 0, 0, x, n4, out);
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((read_kernel<U, NT>), dim3(blocks), dim3(256), 0, 0, x, n4, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return (double)n4 * 16 * reps / (ms * 1e-3) / 1e9;
}

int main() {
    const size_t bytes = 20480000000ull;
    const size_t n4 = bytes / 16;
    f32x4* x;
    float* out;
    if (hipMalloc(&x, bytes) != hipSuccess) return 1;
    hipMalloc(&out, 4);
    hipMemset(x, 1, bytes);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    double best = 0;
    for (int bpc : {1, 2, 3, 4, 6, 8}) {
        const int blocks = cus * bpc;
        double r[6] = {run<1, true>(x, n4, out, blocks, 10), run<2, true>(x, n4, out, blocks, 10),
                       run<4, true>(x, n4, out, blocks, 10), run<8, true>(x, n4, out, blocks, 10),
                       run<4, false>(x, n4, out, blocks, 10), run<8, false>(x, n4, out, blocks, 10)};
        printf("blocks/CU=%d  nt U=1:%7.0f  U=2:%7.0f  U=4:%7.0f  U=8:%7.0f | plain U=4:%7.0f  U=8:%7.0f GB/s\n", bpc,
               r[0], r[1], r[2], r[3], r[4], r[5]);
        for (double v : r) best = v > best ? v : best;
    }
    printf("best streaming read: %.0f GB/s (%.3f of 8000)\n", best, best / 8000.0);
    return 0;
}
