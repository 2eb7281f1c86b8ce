// split128.hpp — interface of split128.hip (the 33..128-query split-precision main launch) to mvdb.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvdb {

struct Split128Args {
    const float* X;
    int64_t n;
    int64_t ld;
    const __bf16* qh;   // K-step-major bf16 images of split_queries_kernel: [K / 32][128][32], rows >= nq zero
    const __bf16* ql;
    int nq;             // <= 128
    uint64_t* cand;     // [nq, gridDim.x, 16] nominee keys per block
    int64_t tile0;      // 32-row tiles [tile0, tile1)
    int64_t tile1;
    const float* thr0;  // [nq] admission floors of the seed launch, or NULL
    unsigned int* stats; // NULL, or [2]: list inserts, wave-tiles that reached the slow path (diagnostics)
};

// true when flat_scan_split128_kernel has an instantiation for dimension d
bool split128_supported(int d);
// launches the kernel on `stream`; *nblocks_out = number of nominee lists per query it writes
int launch_split128(int d, const Split128Args& a, int device, hipStream_t stream, int* nblocks_out);

}  // namespace mvdb
