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
// while profiling is on: remember which kernel instantiation a label's launches run (mvdb_prof_symbol; bench.py checks
// the committed PMC profile against it)
void prof_symbol(const char* label, const char* fmt, ...);

// Every MVDB_* tuning / A-B hook of the SEARCH path.  The environment is read ONCE per index — at mvdb_index_create, and
// again only when the caller asks (mvdb_index_reload_env: A/B runs and tests that flip a hook inside one process) — never on
// the search path.  Defaults are the measured best (docs/DESIGN_NOTES.md section 7).
struct Knobs {
    int scan_blocks_per_cu = 0;      // MVDB_SCAN_BLOCKS_PER_CU (0: the measured default per shape)
    int mfma_blocks_per_cu = 0;      // MVDB_MFMA_BLOCKS_PER_CU (0: 4 for the fragment-load kernel, 2 for the staged ones)
    int mfma_stage = 16;             // MVDB_MFMA_STAGE (8 | 16)
    int mfma_ng2 = -1;               // MVDB_MFMA_NG2 (-1: on for the staged kernel, off otherwise)
    int gemm_scan_min_nq = 104;      // MVDB_GEMM_SCAN_MIN_NQ
    int gemm_scan_blocks_per_cu = 2; // MVDB_GEMM_SCAN_BLOCKS_PER_CU
    int split_scan_min_nq = -1;      // MVDB_SPLIT_SCAN_MIN_NQ: fewest queries of a call for the certified pass (-1: by corpus size, mvdb.hip half_min_nq)
    bool disable_rescue = false;       // MVDB_DISABLE_RESCUE: refused queries go straight to the exact passes (A/B)
    bool disable_tile_skip = false;    // MVDB_DISABLE_TILE_SKIP: the rescue launches scan every tile of the shadow (A/B)
    int tile_flags_mode = -1;          // MVDB_TILE_FLAGS: -1 kept while the index refuses certificates (mvdb.hip: tile_flags_wanted), 1 always, 0 never
    int tile_flag_min_tiles = 12288;   // MVDB_TILE_FLAG_MIN_TILES: rows / 32 from which the certified pass keeps tile flags for the rescue pass
                                       // (393k rows.  Clustered corpus, with / without flags: 1M rows 0.43 / 0.51 ms at 32 per call, 0.91 / 0.96 at
                                       // 256; 500k rows 0.31 / 0.34, 0.645 / 0.639; 200k rows 0.166 / 0.166, 0.336 / 0.321)
    bool disable_rerun_floor = false;  // MVDB_DISABLE_RERUN_FLOOR: the exact re-run of refused queries starts every list from -inf (A/B)
    int half_phase_growth = 0;       // MVDB_HALF_PHASE_GROWTH (0: by the pass width — 16 / 6 up to 128 queries per pass, 6 / 4 at 256)
    int half_last_growth = 0;        // MVDB_HALF_LAST_GROWTH
    bool split_stats = false;        // MVDB_SPLIT_STATS
    bool disable_mfma_scan = false, disable_l2_mfma = false, disable_gemm_scan = false, disable_split_scan = false,
         disable_half_scan = false, disable_masked_batch = false,
         disable_l2_cert = false,    // MVDB_DISABLE_* (L2_CERT: L2 batches back on the exact fp32 kernels)
         disable_half_shadow = false;  // MVDB_DISABLE_HALF_SHADOW (index option half_shadow = 0): no fp16 shadow, batches on the exact fp32 passes
    bool shadow_single_query = false;  // MVDB_SHADOW_SINGLE_QUERY (1: single queries through the certified fp16-shadow pass too)
    long long compact_bytes = 512ll << 20;  // MVDB_COMPACT_BYTES: staging buffer of a row compaction (mvdb_index_remove_rows)
    bool compact_inplace = true;            // MVDB_COMPACT_INPLACE (0: a few deleted rows take the staging path too)
};
Knobs read_knobs();

// hipOccupancyMaxActiveBlocksPerMultiprocessor, asked once per (kernel, dynamic LDS) and remembered
int cached_occupancy(const void* kern, int threads, size_t lds, int dflt);

// A search captured into a hipGraph bakes its workspace pointers into the graph.  While a retire list is installed
// (RetireScope: searches on a workspace that has been captured once), a buffer that must grow is NOT freed — it goes on
// the list and lives as long as the workspace, so replaying the earlier graph still reads and writes valid memory.
extern thread_local std::vector<void*>* tls_retire;
struct RetireScope {
    std::vector<void*>* prev;
    explicit RetireScope(std::vector<void*>* list) : prev(tls_retire) { tls_retire = list; }
    ~RetireScope() { tls_retire = prev; }
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    int reserve(size_t n) {
        if (n <= cap) return 0;
        if (p) {
            if (tls_retire) tls_retire->push_back((void*)p);
            else (void)hipFree(p);
        }
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
