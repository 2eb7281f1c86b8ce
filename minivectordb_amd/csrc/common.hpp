// common.hpp — host-side plumbing shared by the translation units of libmvdb.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mvdb.h"

namespace mvdb {

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define MVDB_HIP(expr)                                                                       \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return ::mvdb::fail(e__ == hipErrorOutOfMemory ? MVDB_ERR_OOM : MVDB_ERR_HIP,    \
                                "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),      \
                                __FILE__, __LINE__);                                         \
    } while (0)

#define MVDB_TRY(expr)           \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) return rc__;   \
    } while (0)

// RAII device switch: every entry point runs on its object's device and restores the caller's.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int ensure_device(int device);  // validates ordinal, MVDB_ERR_NODEVICE otherwise
int device_cus(int device);

// ---- profiling hooks ---------------------------------------------------------------------------
bool prof_enabled();
// returns an opaque slot (or -1 when disabled) after recording the start event on `stream`
int prof_begin(const char* name, hipStream_t stream);
void prof_end(int slot, hipStream_t stream);

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    int reserve(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 64;
        MVDB_HIP(hipMalloc((void**)&p, want * sizeof(T)));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;  // bytes
    int reserve(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 256;
        MVDB_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace mvdb
