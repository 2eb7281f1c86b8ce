// Does v_mfma_f32_32x32x16_f16 read fp16 SUBNORMAL inputs or flush them to zero?  (half_eps in half_scan.hip bounds both
// behaviours; this probe says which one gfx950 has.)  build: hipcc --offload-arch=gfx950 -O2 mfma_f16_subnormal.hip -o mfma_f16_subnormal
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float a, float b, float* out) {
    h8 A, B;
    for (int i = 0; i < 8; ++i) {
        A[i] = (_Float16)a;
        B[i] = (_Float16)b;
    }
    f16v acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* d;
    hipMalloc(&d, 4);
    const float as[] = {1.0f, 6.103515625e-05f /* 2^-14: smallest normal */, 3.0517578125e-05f /* 2^-15 */, 9.5367431640625e-07f /* 2^-20 */,
                        5.9604644775390625e-08f /* 2^-24: smallest subnormal */};
    for (float a : as) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 1024.0f, d);
        float h;
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a = %.10e (x 1024 x 16 terms): mfma = %.10e expected %.10e -> %s\n", a, h, a * 1024.0 * 16, h == a * 1024.0f * 16 ? "kept" : (h == 0 ? "FLUSHED" : "other"));
    }
    return 0;
}
