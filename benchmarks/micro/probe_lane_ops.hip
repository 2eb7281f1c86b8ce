// probe_lane_ops.hip — prints the observed lane mapping of the gfx950 cross-lane primitives the encoder kernels rely
// on, so that a wrong assumption shows up as a readable table instead of a failed parity test:
//   1. ds_read_b64_tr_b16: which LDS element lands in which (lane, element) slot;
//   2. v_permlane16_swap_b32 with both operands equal: the row pairing of the "all-reduce over a 32-lane half";
//   3. DPP row_ror all-reduce over 16 lanes.
// build: hipcc --offload-arch=gfx950 -O2 probe_lane_ops.hip -o probe_lane_ops ; run on a GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__global__ void tr_probe(uint16_t* out /* [64][4] */) {
    __shared__ __attribute__((aligned(16))) uint16_t m[64 * 64];  // [row][col], row pitch 128 bytes; value = row * 64 + col
    for (int i = threadIdx.x; i < 64 * 64; i += 64) m[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    // group g reads the block with first row 8 g, first column 16 (g & 1): lane (q, p) points at row q, columns 4 p ..
    const unsigned addr = (unsigned)(size_t)(&m[(8 * g + q) * 64 + 16 * (g & 1) + 4 * p]);
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = (uint16_t)(v[0] & 0xffff);
    out[lane * 4 + 1] = (uint16_t)(v[0] >> 16);
    out[lane * 4 + 2] = (uint16_t)(v[1] & 0xffff);
    out[lane * 4 + 3] = (uint16_t)(v[1] >> 16);
}

__global__ void swap16_probe(float* out /* [3][64] */) {
    const int lane = threadIdx.x;
    const float x = (float)(1 << (lane >> 4)) * 1000.f + (float)lane;  // row id in the thousands, lane in the units
    const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    out[lane] = __uint_as_float(r[0]);
    out[64 + lane] = __uint_as_float(r[1]);
    // DPP rotation all-reduce over a row of 16 lanes
    float s = (float)(lane & 15);
    s += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(s), 0x128, 0xf, 0xf, false));  // row_ror:8
    s += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(s), 0x124, 0xf, 0xf, false));  // row_ror:4
    s += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(s), 0x122, 0xf, 0xf, false));  // row_ror:2
    s += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(s), 0x121, 0xf, 0xf, false));  // row_ror:1
    out[128 + lane] = s;  // expected 120 everywhere
}

int main() {
    uint16_t* d_tr;
    float* d_sw;
    hipMalloc(&d_tr, 64 * 4 * sizeof(uint16_t));
    hipMalloc(&d_sw, 3 * 64 * sizeof(float));
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d_tr);
    hipLaunchKernelGGL(swap16_probe, dim3(1), dim3(64), 0, 0, d_sw);
    std::vector<uint16_t> tr(256);
    std::vector<float> sw(192);
    hipMemcpy(tr.data(), d_tr, 512, hipMemcpyDeviceToHost);
    hipMemcpy(sw.data(), d_sw, 768, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16 (value = row * 64 + col of a [64][64] u16 matrix; group g points at rows 8g.., cols 16(g&1)..):\n");
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, li = lane & 15;
        printf("  lane %2d:", lane);
        for (int e = 0; e < 4; ++e) {
            const int v = tr[lane * 4 + e];
            printf(" (r%2d,c%2d)", v / 64, v % 64);
            // expectation: element e = row 8 g + e, column 16 (g & 1) + li
            if (v != (8 * g + e) * 64 + 16 * (g & 1) + li) ++bad;
        }
        printf("\n");
    }
    printf("tr_b16 expectation (element e of lane li in group g = row 8g+e, col 16(g&1)+li): %s (%d mismatches)\n",
           bad ? "MISMATCH" : "ok", bad);
    printf("permlane16_swap(x, x): lane -> (new vdst, new src)\n");
    int bad2 = 0;
    for (int lane = 0; lane < 64; ++lane) {
        printf("  lane %2d: %7.0f %7.0f\n", lane, sw[lane], sw[64 + lane]);
        // expectation: vdst rows (d0, s0, d2, s2), src rows (d1, s1, d3, s3) with d = s = x: the sum of the two is the
        // sum over the row pair, i.e. x(lane) + x(lane ^ 16)
        const int partner = lane ^ 16;
        const float xl = (float)(1 << (lane >> 4)) * 1000.f + (float)lane, xp = (float)(1 << (partner >> 4)) * 1000.f + (float)partner;
        if (sw[lane] + sw[64 + lane] != xl + xp) ++bad2;
    }
    printf("permlane16_swap all-reduce expectation (r0 + r1 == x[lane] + x[lane ^ 16]): %s (%d mismatches)\n",
           bad2 ? "MISMATCH" : "ok", bad2);
    int bad3 = 0;
    for (int lane = 0; lane < 64; ++lane)
        if (sw[128 + lane] != 120.f) ++bad3;
    printf("row_ror all-reduce over 16 lanes: %s (%d mismatches; lane 0 = %g)\n", bad3 ? "MISMATCH" : "ok", bad3, sw[128]);
    return (bad || bad2 || bad3) ? 1 : 0;
}
