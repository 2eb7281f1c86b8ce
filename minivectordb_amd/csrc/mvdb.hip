// mvdb.hip — C-ABI (include/mvdb.h) of the flat index: host orchestration + kernel launches.
//
// HBM layout of an index: ONE row-major fp32 matrix X[cap, ld], ld = d rounded up to a multiple
// of 4 floats so that every row starts on a 16-byte boundary and is read as whole dwordx4 chunks;
// the padding is zero.  Rows [0, n) are live.  Nothing else of the corpus lives on the device
// (ids / metadata stay in the Python host layer, as in the reference).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <map>
#include <tuple>
#include <vector>
#include <shared_mutex>

#include "common.hpp"
#include "scan_kernels.hpp"
#include "scan_mfma_kernels.hpp"
#include "select_kernels.hpp"
#include "half_scan.hpp"
#include "util_kernels.hpp"

using namespace mvdb;

// ================================================================================================
// errors / device helpers / profiling
// ================================================================================================
namespace mvdb {

int launch_merge_di_sort(int metric, int nlists, int nq, int k, const float* D, int64_t strideD, const int64_t* I,
                         int64_t strideI, float* Dout, int64_t* Iout, int device, hipStream_t stream);  // collective.hip

static thread_local char g_err[512] = "";
thread_local std::vector<void*>* tls_retire = nullptr;

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
Knobs read_knobs() {
    Knobs k;
    k.scan_blocks_per_cu = env_int("MVDB_SCAN_BLOCKS_PER_CU", 0);
    k.mfma_blocks_per_cu = env_int("MVDB_MFMA_BLOCKS_PER_CU", 0);
    k.mfma_stage = env_int("MVDB_MFMA_STAGE", 16);
    k.mfma_ng2 = env_int("MVDB_MFMA_NG2", -1);
    k.gemm_scan_min_nq = env_int("MVDB_GEMM_SCAN_MIN_NQ", 104);
    k.gemm_scan_blocks_per_cu = env_int("MVDB_GEMM_SCAN_BLOCKS_PER_CU", 2);
    k.split_scan_min_nq = env_int("MVDB_SPLIT_SCAN_MIN_NQ", -1);
    k.half_phase_growth = env_int("MVDB_HALF_PHASE_GROWTH", 0);   // 0: by the pass width (launch_half_pass)
    k.half_last_growth = env_int("MVDB_HALF_LAST_GROWTH", 0);
    k.split_stats = env_int("MVDB_SPLIT_STATS", 0) != 0;
    k.disable_mfma_scan = env_int("MVDB_DISABLE_MFMA_SCAN", 0) != 0;
    k.disable_l2_mfma = env_int("MVDB_DISABLE_L2_MFMA", 0) != 0;
    k.disable_gemm_scan = env_int("MVDB_DISABLE_GEMM_SCAN", 0) != 0;
    k.disable_split_scan = env_int("MVDB_DISABLE_SPLIT_SCAN", 0) != 0;
    k.disable_half_scan = env_int("MVDB_DISABLE_HALF_SCAN", 0) != 0;
    k.disable_rerun_floor = env_int("MVDB_DISABLE_RERUN_FLOOR", 0) != 0;
    k.disable_rescue = env_int("MVDB_DISABLE_RESCUE", 0) != 0;
    k.disable_tile_skip = env_int("MVDB_DISABLE_TILE_SKIP", 0) != 0;
    k.tile_flag_min_tiles = env_int("MVDB_TILE_FLAG_MIN_TILES", 12288);
    k.tile_flags_mode = env_int("MVDB_TILE_FLAGS", -1);
    k.disable_masked_batch = env_int("MVDB_DISABLE_MASKED_BATCH", 0) != 0;
    k.disable_l2_cert = env_int("MVDB_DISABLE_L2_CERT", 0) != 0;
    k.disable_half_shadow = env_int("MVDB_DISABLE_HALF_SHADOW", 0) != 0;
    k.shadow_single_query = env_int("MVDB_SHADOW_SINGLE_QUERY", 0) != 0;
    if (const char* v = getenv("MVDB_COMPACT_BYTES"))
        if (*v) k.compact_bytes = std::max(1ll, atoll(v));
    k.compact_inplace = env_int("MVDB_COMPACT_INPLACE", 1) != 0;
    return k;
}

int cached_occupancy(const void* kern, int threads, size_t lds, int dflt) {
    static std::mutex mu;
    static std::map<std::tuple<const void*, size_t, int, int>, int> cache;  // (kernel, dynamic LDS, threads, current device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_tuple(kern, lds, threads, dev);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds) != hipSuccess || nb <= 0) nb = dflt;
    cache[key] = nb;
    return nb;
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int ensure_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MVDB_ERR_NODEVICE, "no HIP device available (%s); libmvdb has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail(MVDB_ERR_ARG, "device ordinal %d out of range [0,%d)", device, n);
    return 0;
}

int device_cus(int device) {
    static std::mutex mu;
    static std::map<int, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(device);
    if (it != cache.end()) return it->second;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
        cus <= 0)
        cus = 256;
    cache[device] = cus;
    return cus;
}

struct ProfPair {
    std::string name;
    hipEvent_t a, b;
    bool closed;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static int g_prof_next = 0;
static std::map<int, ProfPair> g_prof;  // keyed by a ticket that stays valid across mvdb_prof_read

static std::map<std::string, std::string> g_prof_sym;  // label -> the kernel instantiation last launched under it
void prof_symbol(const char* label, const char* fmt, ...) {
    if (!g_prof_on) return;
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_sym[label] = buf;
}
bool prof_enabled() { return g_prof_on; }
int prof_begin(const char* name, hipStream_t stream) {
    if (!g_prof_on) return -1;
    ProfPair p;
    p.name = name;
    p.closed = false;
    if (hipEventCreate(&p.a) != hipSuccess) return -1;
    if (hipEventCreate(&p.b) != hipSuccess) {
        (void)hipEventDestroy(p.a);
        return -1;
    }
    (void)hipEventRecord(p.a, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const int ticket = g_prof_next++;
    g_prof[ticket] = p;
    return ticket;
}
void prof_end(int ticket, hipStream_t stream) {
    if (ticket < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto it = g_prof.find(ticket);
    if (it != g_prof.end()) {
        (void)hipEventRecord(it->second.b, stream);
        it->second.closed = true;
    }
}

}  // namespace mvdb

// ================================================================================================
// index object
// ================================================================================================
namespace {

struct Workspace {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevBuf<float> q;
    DevBuf<float> qn;  // normalised copy of the queries for the multi-query pass
    DevBuf<uint64_t> cand;
    DevBuf<int64_t> out;  // packed results of the host API: [I: total int64 | D: total fp32] -> ONE D2H copy
    PinnedBuf pin_out;
    DevBuf<float> scores;
    DevBuf<uint64_t> selkeys;
    DevBuf<int64_t> rows;
    DevBuf<__bf16> qsplit;  // the fp16 image of one chunk of queries (certified pass; 2-byte elements)
    DevBuf<float> qnorm;
    DevBuf<int> flags;      // per-chunk count of uncertified queries
    DevBuf<int> qfail;      // per query: 1 = failed certification; behind them the compact list of those queries
    DevBuf<float> requery;  // the failed queries, gathered, and their exact results [D | I]
    DevBuf<int64_t> relabel;
    DevBuf<int> nfail;      // number of uncertified queries of the call (device-side gate of the exact re-run)
    DevBuf<int> need;       // one word per 32-query exact re-run pass: raised by the rescue pass where a query stays unanswered
    DevBuf<float> qfloor;   // per query: the admission floor of its exact re-run (half_certify_kernel), behind them the compact copy
    DevBuf<uint32_t> tflags;  // certified pass: one bit per (query of the call, 32-row tile) that may matter to the rescue pass
    DevBuf<int> tlist;        // the rescue launches' tile lists
    SelectState* st = nullptr;
    PinnedBuf pin;
    std::mutex use_mu;  // stream workspaces are shared by every host thread that names the stream: one search at a time
    bool captured = false;         // a search on this workspace has been captured into a hipGraph: its buffers are never freed
    std::vector<void*> retired;    // ... outgrown ones wait here until the workspace goes (common.hpp, RetireScope)

    int init(int dev, hipStream_t s) {
        device = dev;
        if (s) {
            stream = s;
            own_stream = false;
        } else {
            MVDB_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            own_stream = true;
        }
        MVDB_HIP(hipMalloc((void**)&st, sizeof(SelectState)));
        return 0;
    }
    void destroy() {
        q.release();
        qn.release();
        cand.release();
        out.release();
        pin_out.release();
        scores.release();
        selkeys.release();
        rows.release();
        qsplit.release();
        qnorm.release();
        flags.release();
        qfail.release();
        requery.release();
        relabel.release();
        nfail.release();
        pin.release();
        for (void* old : retired) (void)hipFree(old);
        retired.clear();
        if (st) (void)hipFree(st);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace

struct mvdb_index {
    int d = 0, d4 = 0, metric = 0, device = 0;
    int64_t ld = 0;
    float* X = nullptr;
    int64_t n = 0, cap = 0;
    float row_norm_bound = 0.f;  // upper bound of |row| over the stored rows (INFINITY: unknown, raw adds)
    float norm2_lo = INFINITY, norm2_hi = 0.f;  // L2 metric only: bounds of |row|^2 over the stored rows (true values inside)
    uint64_t renumbered = 0;     // bumped whenever stored rows change their numbers (remove_rows, reset): resident row sets
                                 // built before are stale
    Knobs kn;                    // the MVDB_* hooks as read at creation (mvdb_index_reload_env re-reads)
    hipStream_t mut = nullptr;   // the mutators' own non-blocking stream: add / remove_rows never touch the legacy stream,
                                 // so work other libraries have in flight on the device (an encoder forward) is not stalled
    unsigned int* normmax = nullptr;  // 4-byte scratch of note_row_norms (raw adds)
    float* ctmp = nullptr;            // bounded staging buffer of mvdb_index_remove_rows (kept once a delete has run)
    size_t ctmp_bytes = 0;
    // fp16 shadow of the rows (half_scan.hip: flat_scan_h16_kernel): built by the first batch search that can use it, extended
    // by add, emptied (allocation kept) by whatever renumbers rows or changes the scale, freed with the matrix.  Searches (shared lock) build / read it under
    // shadow_mu; mutators (exclusive lock, searches quiesced) edit it directly.
    mutable std::mutex shadow_mu;
    mutable _Float16* Xh = nullptr;
    mutable int64_t xh_cap = 0, xh_rows = 0;   // rows allocated (+ kRowSlack behind them) / rows converted
    mutable float xh_scale = 0.f;
    mutable std::atomic<bool> xh_failed{false};  // allocation failed: not retried until the index changes (read by routing code
                                                 // that holds only the shared lock, written under shadow_mu: atomic)
    mutable float* Hn = nullptr;               // L2 over rows of mixed norms: |x_r|^2 / 2 of rows [0, hn_rows) (+ zeroed slack), dies with Xh
    mutable int64_t hn_rows = 0;
    // The opt-in single-query route over the shadow costs MORE than the exact scan when its certificate is refused (the
    // nomination pass, then the exact scan anyway: 0.59 against 0.31 ms per query on a clustered 1M x 512 corpus,
    // BENCH certified_passes_on_unfriendly_data).  The device keeps a running count of refused single-query certificates and
    // mirrors it into a host-mapped word; the routing looks at windows of 32 such calls and, when half of a window was
    // refused, sends the next 512 single queries straight to the exact scan before it probes again.  No synchronisation: the
    // word lags by the calls in flight, which a heuristic can afford.
    mutable unsigned int* sq_fail_dev = nullptr;
    mutable PinnedBuf sq_fail_host;
    mutable std::atomic<unsigned int> sq_calls{0}, sq_window_calls{0}, sq_window_fail{0};
    mutable std::atomic<int> sq_suspend_left{0};
    // adaptive tile flags (search_core): a running count of refused certificates of ANY certified call, mirrored into a
    // host-mapped word; the certified pass keeps tile flags only while that count has been moving
    mutable unsigned int* rf_dev = nullptr;
    mutable PinnedBuf rf_host;
    mutable std::atomic<unsigned int> rf_seen{0};
    mutable std::atomic<int> rf_calls_left{0};
    mutable std::atomic<unsigned long long> sq_suspensions{0};
    mutable std::shared_mutex mu;  // search: shared; add/reset/remove/free: exclusive
    mutable std::mutex ws_mu;
    mutable std::vector<Workspace*> free_ws;           // synchronous searches
    mutable std::map<void*, Workspace*> stream_ws;     // async searches, one per caller stream
    mutable Workspace* default_stream_ws = nullptr;

    Workspace* acquire() const {
        {
            std::lock_guard<std::mutex> lk(ws_mu);
            if (!free_ws.empty()) {
                Workspace* w = free_ws.back();
                free_ws.pop_back();
                return w;
            }
        }
        Workspace* w = new Workspace();
        if (w->init(device, nullptr)) {
            w->destroy();
            delete w;
            return nullptr;
        }
        return w;
    }
    void release(Workspace* w) const {
        std::lock_guard<std::mutex> lk(ws_mu);
        free_ws.push_back(w);
    }
    Workspace* for_stream(hipStream_t s) const {
        std::lock_guard<std::mutex> lk(ws_mu);
        if (!s) {
            if (!default_stream_ws) {
                // legacy default stream: work is enqueued on stream 0 itself
                Workspace* w = new Workspace();
                w->device = device;
                w->stream = nullptr;
                w->own_stream = false;
                if (hipMalloc((void**)&w->st, sizeof(SelectState)) != hipSuccess) {
                    delete w;
                    return nullptr;
                }
                default_stream_ws = w;
            }
            return default_stream_ws;
        }
        auto it = stream_ws.find((void*)s);
        if (it != stream_ws.end()) return it->second;
        Workspace* w = new Workspace();
        if (w->init(device, s)) {
            w->destroy();
            delete w;
            return nullptr;
        }
        stream_ws[(void*)s] = w;
        return w;
    }
};

namespace {

// The launchers below read the hooks of the index being searched through a thread-local pointer (search_core installs it).
const Knobs g_default_knobs;
thread_local const Knobs* tls_kn = &g_default_knobs;
struct KnobScope {
    const Knobs* prev;
    explicit KnobScope(const Knobs* k) : prev(tls_kn) { tls_kn = k; }
    ~KnobScope() { tls_kn = prev; }
};
inline const Knobs& kn() { return *tls_kn; }

// ---- shape selection: G lanes per row, C chunks per lane, U rows in flight ----------------------
struct Shape {
    int G, C;
};
Shape choose_shape(int d4) {
    // one chunk per lane, 16 / 32 / 64 lanes per row (rows of up to 64 / 128 / 256 floats; the 1-, 2-, 4- and 8-lane
    // shapes of rounds 1 - 4 served rows of <= 4, 8 and 32 floats only: 84 instantiations for toy widths)
    if (d4 <= 64) return {d4 <= 16 ? 16 : d4 <= 32 ? 32 : 64, 1};  // (rows of <= 16 floats ride the 16-lane shape, lanes masked)
    if (d4 % 64 == 0 && d4 / 64 <= 8) return {64, d4 / 64};
    if (d4 == 96) return {32, 3};  // d = 384 (e5-small, config 5): three exact chunks on 32 lanes, two rows per wave-instruction
    // more than eight chunks per lane (d > 2048): ONE shape, sixteen chunks with the lanes beyond the row masked off — the
    // seven shapes in between (nine to fifteen chunks: 224 kernel instantiations, 30 % of this code object) went in round 5;
    // a masked chunk issues no load, the scan stays bound by the bytes of the row
    const int c = (d4 + 63) / 64;
    return {64, c > 8 ? 16 : c};
}
constexpr int kMaxC = 16;  // d <= 4096

template <int G, int C, int U, int METRIC, int MODE, bool NT, int SEL, bool MASKED>
int launch_scan_kern(const ScanArgs& a, int nq, int device, hipStream_t stream, int* nblocks_out) {
    void (*kern)(ScanArgs) = flat_scan_kernel<G, C, U, METRIC, MODE, NT, SEL, MASKED>;
    bool gated = false;
    // device-gated exact re-run of an L2 batch's queries: over the whole index (SEL 0) and under a bitmap (SEL 2 — round-4
    // advisor finding: without the gated instantiation every query of a certified masked L2 batch paid a full masked scan)
    if constexpr (METRIC == 1 && MODE == kModeTopK && (SEL == 0 || SEL == 2)) {
        if (a.gate) {
            kern = flat_scan_kernel<G, C, U, METRIC, MODE, NT, SEL, MASKED, true>;
            gated = true;
        }
    }
    if (a.gate && !gated) return fail(MVDB_ERR_ARG, "internal: a device-gated scan was requested for a kernel form that has no gated instantiation");
    const int occ_hw = cached_occupancy((const void*)kern, kScanThreads, 0, 4);  // blocks per CU this instantiation sustains
    int occ;
    {
        const int nb = occ_hw;
        // measured on MI355X (round-1 sweep, profiles/r01_sweep_scan_variants.txt, 10M x 512): with ~4
        // independent 16-B loads per lane, 2 resident blocks (8 waves) per CU reach 7.21-7.24 TB/s; more
        // waves or more loads in flight per lane are 2-4 % slower, 1 block per CU is latency-starved
        // (gpurun_out/sweep_dims.log: the one- and three-chunk shapes — d = 64 / 256 / 384 — prefer U = 4 with
        //  3 resident blocks: 7.0-7.1 TB/s vs 6.5-6.9 at 2)
        occ = std::min(nb, (C == 1 || C == 3) ? 3 : 2);
        const int cap_env = kn().scan_blocks_per_cu;  // tuning hook
        if (cap_env > 0) occ = std::min(occ_hw, cap_env);
    }
    constexpr int RB = (kWave / G) * U;
    const int64_t nbatches = (a.n + RB - 1) / RB;
    int64_t want = (nbatches + kScanWaves - 1) / kScanWaves;
    int64_t cap = (int64_t)device_cus(device) * occ;
    int nblocks = (int)std::max<int64_t>(1, std::min(want, cap));
    if (nblocks_out) *nblocks_out = nblocks;
    const char* pname = MODE == kModeTopK ? "ip_scan" : "ip_scan_scores";
    prof_symbol(pname, "flat_scan_kernel<%d, %d, %d, %d, %d, %s, %d, %s, %s>", G, C, U, METRIC, MODE, NT ? "true" : "false", SEL,
                MASKED ? "true" : "false", gated ? "true" : "false");
    int slot = prof_begin(pname, stream);
    hipLaunchKernelGGL(kern, dim3(nblocks, nq), dim3(kScanThreads), 0, stream, a);
    prof_end(slot, stream);
    MVDB_HIP(hipGetLastError());
    return 0;
}

// runtime -> compile-time switches: subset indirection, lane masking
template <int G, int C, int U, int METRIC, int MODE, bool NT = true>
int launch_scan_inst(const ScanArgs& a, int nq, int device, hipStream_t s, int* nb) {
    // (32 lanes x 3 / 5 / 7 chunks are chosen for rows that fill them exactly: their lane-masked forms are never instantiated)
    constexpr bool kAlwaysFull = G == 32 && C > 1;
    const bool masked = !kAlwaysFull && a.d4 != G * C;
    if constexpr (kAlwaysFull) {
        if (a.mask) return launch_scan_kern<G, C, U, METRIC, MODE, NT, 2, false>(a, nq, device, s, nb);
        if (a.rows) return launch_scan_kern<G, C, U, METRIC, MODE, NT, 1, false>(a, nq, device, s, nb);
        return launch_scan_kern<G, C, U, METRIC, MODE, NT, 0, false>(a, nq, device, s, nb);
    }
    if (a.mask) {  // bitmap-selected rows (mvdb_index_search_masked)
        if (masked) return launch_scan_kern<G, C, U, METRIC, MODE, NT, 2, true>(a, nq, device, s, nb);
        return launch_scan_kern<G, C, U, METRIC, MODE, NT, 2, false>(a, nq, device, s, nb);
    }
    if (a.rows) {
        if (masked) return launch_scan_kern<G, C, U, METRIC, MODE, NT, 1, true>(a, nq, device, s, nb);
        return launch_scan_kern<G, C, U, METRIC, MODE, NT, 1, false>(a, nq, device, s, nb);
    }
    if (masked) return launch_scan_kern<G, C, U, METRIC, MODE, NT, 0, true>(a, nq, device, s, nb);
    return launch_scan_kern<G, C, U, METRIC, MODE, NT, 0, false>(a, nq, device, s, nb);
}

template <int G, int C, int U>
int launch_scan_gcu(int metric, int mode, const ScanArgs& a, int nq, int device, hipStream_t s,
                    int* nb) {
    if (metric == MVDB_METRIC_IP) {
        if (mode == kModeTopK) return launch_scan_inst<G, C, U, 0, kModeTopK>(a, nq, device, s, nb);
        return launch_scan_inst<G, C, U, 0, kModeScores>(a, nq, device, s, nb);
    }
    if (mode == kModeTopK) return launch_scan_inst<G, C, U, 1, kModeTopK>(a, nq, device, s, nb);
    return launch_scan_inst<G, C, U, 1, kModeScores>(a, nq, device, s, nb);
}

// the row-list (SEL 1) forms of a shape only: the gather variants of the two- and three-chunk shapes exist for nothing else
template <int G, int C, int U>
int launch_scan_rows(int metric, int mode, const ScanArgs& a, int nq, int device, hipStream_t s, int* nb) {
    const bool masked = a.d4 != G * C;
#define MVDB_ROWS_CASE(M, MD)                                                                              \
    (masked ? launch_scan_kern<G, C, U, M, MD, true, 1, true>(a, nq, device, s, nb) \
            : launch_scan_kern<G, C, U, M, MD, true, 1, false>(a, nq, device, s, nb))
    if (metric == MVDB_METRIC_IP) return mode == kModeTopK ? MVDB_ROWS_CASE(0, kModeTopK) : MVDB_ROWS_CASE(0, kModeScores);
    return mode == kModeTopK ? MVDB_ROWS_CASE(1, kModeTopK) : MVDB_ROWS_CASE(1, kModeScores);
#undef MVDB_ROWS_CASE
}

// Grid size the scan will use for (shape, n): needed up front to size the candidate buffer.
int scan_grid_upper_bound(int device) { return device_cus(device) * 8; }

// Raises a kernel's dynamic-LDS limit once per (kernel, device): the attribute is per device, and one process may
// hold indexes on several GPUs.
int ensure_dynamic_lds(const void* kern, size_t lds, int device) {
    if (lds <= 48 * 1024) return 0;
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> done;
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = done[{kern, device}];
    if (lds > have) {
        MVDB_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        have = lds;
    }
    return 0;
}

int launch_scan(int metric, int mode, const ScanArgs& a, int nq, int device, hipStream_t s,
                int* nblocks) {
    const Shape sh = choose_shape(a.d4);
    // (Rounds 1 - 2 carried run-time tuning hooks here — MVDB_SCAN_VARIANT, MVDB_SCAN_U: other rows-in-flight counts per shape,
    //  17 more shapes x 24 kernels each; the sweeps they served are in profiles/r01_sweep_scan_variants.txt.  Removed in round 3:
    //  they were 40 % of this code object, which a process loads whole before its first search.)
    // Row-list (gather) scans of two- and three-chunk rows keep FOUR rows in flight per wave (the streaming scan: two): a
    // gathered row is a fresh DRAM page, so more rows must be outstanding to cover its latency — 10M x 512 rows resident,
    // ids on the device: 10 % of the rows 5.74 -> 6.23 TB/s of rows touched, 50 % 6.37 -> 6.78, 99 % 6.54 -> 6.95.
    if (a.rows && sh.G == 64 && sh.C == 2) return launch_scan_rows<64, 2, 4>(metric, mode, a, nq, device, s, nblocks);
    if (a.rows && sh.G == 64 && sh.C == 3) return launch_scan_rows<64, 3, 4>(metric, mode, a, nq, device, s, nblocks);
#define MVDB_SCAN_CASE(G_, C_, U_) \
    if (sh.G == G_ && sh.C == C_) return launch_scan_gcu<G_, C_, U_>(metric, mode, a, nq, device, s, nblocks);
    MVDB_SCAN_CASE(16, 1, 4)
    MVDB_SCAN_CASE(32, 1, 4)
    MVDB_SCAN_CASE(32, 3, 4)
    MVDB_SCAN_CASE(64, 1, 4)
    MVDB_SCAN_CASE(64, 2, 2)
    MVDB_SCAN_CASE(64, 3, 2)
    MVDB_SCAN_CASE(64, 4, 1)
    MVDB_SCAN_CASE(64, 5, 1)
    MVDB_SCAN_CASE(64, 6, 1)
    MVDB_SCAN_CASE(64, 7, 1)
    MVDB_SCAN_CASE(64, 8, 1)
    MVDB_SCAN_CASE(64, 16, 1)
#undef MVDB_SCAN_CASE
    return fail(MVDB_ERR_ARG, "dimension with %d 16-byte chunks per row is not supported (d <= 4096)",
                a.d4);
}

__global__ void fill_missing_kernel(float* D, int64_t* I, int64_t total, int metric) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        D[i] = metric == 0 ? -3.402823466e+38f : 3.402823466e+38f;
        I[i] = -1;
    }
}

int64_t pow2ceil(int64_t v) {
    int64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

int normalize_range(const mvdb_index* idx, float* base, int64_t n, hipStream_t s);

// Mutators (add / remove_rows / reset / reserve) hold the index exclusively, which only excludes HOST calls: scans enqueued
// earlier by mvdb_index_search_device on caller streams may still be reading the matrix.  Wait for THIS index's searches —
// every stream that ever searched it has a workspace here (the synchronous API's own workspaces are idle once a call has
// returned) — and for nothing else: no hipDeviceSynchronize, so an encoder forward or another index on the same device
// keeps running.  The mutators then work on the index's own non-blocking stream (idx->mut) and wait for that stream only.
int quiesce(mvdb_index* idx) {
    std::lock_guard<std::mutex> lk(idx->ws_mu);
    // A caller may have destroyed a stream it once searched on (hipStreamDestroy completes the stream's work first), or be
    // capturing one: neither may wedge the index (round-4 advisor finding: the error made every later add / remove_rows /
    // reset / reserve fail).  A dead handle is forgotten with its workspace; any other failure falls back to the device-wide
    // wait the mutators used before round 4.
    bool device_wide = false;
    for (auto it = idx->stream_ws.begin(); it != idx->stream_ws.end();) {
        const hipError_t e = hipStreamSynchronize((hipStream_t)it->first);
        if (e == hipSuccess) {
            ++it;
            continue;
        }
        (void)hipGetLastError();
        if (e == hipErrorInvalidHandle || e == hipErrorInvalidResourceHandle || e == hipErrorContextIsDestroyed) {
            it->second->destroy();
            delete it->second;
            it = idx->stream_ws.erase(it);
        } else {
            device_wide = true;
            ++it;
        }
    }
    if (device_wide) MVDB_HIP(hipDeviceSynchronize());
    if (idx->default_stream_ws) MVDB_HIP(hipStreamSynchronize(nullptr));
    if (!idx->mut) MVDB_HIP(hipStreamCreateWithFlags(&idx->mut, hipStreamNonBlocking));
    return 0;
}

template <int KB, int NG>
int launch_mfma_inst(const MfmaScanArgs& a, int device, hipStream_t stream, int* nblocks_out) {
    auto kern = flat_scan_mfma_kernel<KB, NG>;
    const size_t lds = (size_t)NG * KB * 4 * 64 * 4 + (size_t)kScanWaves * NG * 16 * a.k * 8;
    MVDB_TRY(ensure_dynamic_lds((const void*)kern, lds, device));
    int nb = cached_occupancy((const void*)kern, kScanThreads, lds, 1);
    nb = std::min(nb, kn().mfma_blocks_per_cu > 0 ? kn().mfma_blocks_per_cu : 4);
    const int64_t ntiles = (a.n + 15) / 16;
    const int64_t want = (ntiles + kScanWaves - 1) / kScanWaves;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)device_cus(device) * nb));
    *nblocks_out = nblocks;
    int slot = prof_begin("ip_scan_mfma", stream);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(kScanThreads), lds, stream, a);
    prof_end(slot, stream);
    MVDB_HIP(hipGetLastError());
    return 0;
}

template <int KB, int NG, int SKB>
int launch_mfma2_gated_inst(const MfmaScanArgs& a, int device, hipStream_t stream, int* nblocks_out, const int* gate, int gate_lo,
                            const int* need, int npasses, int per_pass, int rtot, int64_t cand_stride) {
    auto kern = a.mask ? flat_scan_mfma2_gated_kernel<KB, NG, SKB, true> : flat_scan_mfma2_gated_kernel<KB, NG, SKB, false>;
    const size_t lds = (size_t)kScanWaves * mfma2_wave_lds_bytes(SKB) + (size_t)kScanWaves * NG * 16 * a.k * 8;
    MVDB_TRY(ensure_dynamic_lds((const void*)kern, lds, device));
    int nb = cached_occupancy((const void*)kern, kScanThreads, lds, 1);
    nb = std::min(nb, kn().mfma_blocks_per_cu > 0 ? kn().mfma_blocks_per_cu : 2);
    const int64_t ntiles = (a.n + 15) / 16;
    const int64_t want = (ntiles + kScanWaves - 1) / kScanWaves;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)device_cus(device) * nb));
    *nblocks_out = nblocks;
    prof_symbol("ip_scan_rerun", "flat_scan_mfma2_gated_kernel<%d, %d, %d, %s>", KB, NG, SKB, a.mask ? "true" : "false");
    int slot = prof_begin("ip_scan_rerun", stream);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(kScanThreads), lds, stream, a, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
    prof_end(slot, stream);
    MVDB_HIP(hipGetLastError());
    return 0;
}

template <int KB, int NG, int SKB>
int launch_mfma2_inst(const MfmaScanArgs& a, int device, hipStream_t stream, int* nblocks_out, int metric = MVDB_METRIC_IP) {
    // the L2 form exists where it fits the registers (mfma_path_ok / two_groups route L2 batches accordingly)
    constexpr bool kL2Form = KB <= 48 && !(KB == 32 && NG == 2);
    auto kern = flat_scan_mfma2_kernel<KB, NG, SKB, 0>;
    if (metric == MVDB_METRIC_L2) {
        if constexpr (kL2Form)
            kern = flat_scan_mfma2_kernel<KB, NG, SKB, 1>;
        else
            return fail(MVDB_ERR_ARG, "no L2 form of the staged multi-query kernel for d = %d with %d query group(s)", KB * 16, NG);
    }
    const size_t lds = (size_t)kScanWaves * mfma2_wave_lds_bytes(SKB) + (size_t)kScanWaves * NG * 16 * a.k * 8;
    MVDB_TRY(ensure_dynamic_lds((const void*)kern, lds, device));
    int nb = cached_occupancy((const void*)kern, kScanThreads, lds, 1);
    nb = std::min(nb, kn().mfma_blocks_per_cu > 0 ? kn().mfma_blocks_per_cu : 2);
    const int64_t ntiles = (a.n + 15) / 16;
    const int64_t want = (ntiles + kScanWaves - 1) / kScanWaves;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)device_cus(device) * nb));
    *nblocks_out = nblocks;
    prof_symbol("ip_scan_mfma", "flat_scan_mfma2_kernel<%d, %d, %d, %d>", KB, NG, SKB, metric == MVDB_METRIC_L2 ? 1 : 0);
    int slot = prof_begin("ip_scan_mfma", stream);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(kScanThreads), lds, stream, a);
    prof_end(slot, stream);
    MVDB_HIP(hipGetLastError());
    return 0;
}

template <int NG>
int launch_mfma2(int KB, const MfmaScanArgs& a, int device, hipStream_t s, int* nb, int metric = MVDB_METRIC_IP) {
    // 1-KiB-per-row stages (one contiguous KiB per DMA instruction) measured 2-5 % faster than 512-B
    // stages at 10M x 512; they need 32 KiB of LDS per wave, so fall back when the k-lists do not fit
    const size_t lds_deep = (size_t)kScanWaves * mfma2_wave_lds_bytes(16) + (size_t)kScanWaves * NG * 16 * a.k * 8;
    const bool deep = kn().mfma_stage == 16 && lds_deep <= 160 * 1024;
    switch (KB) {
        case 8: return launch_mfma2_inst<8, NG, 8>(a, device, s, nb, metric);
        case 16: return deep ? launch_mfma2_inst<16, NG, 16>(a, device, s, nb, metric)
                             : launch_mfma2_inst<16, NG, 8>(a, device, s, nb, metric);
        case 24: return launch_mfma2_inst<24, NG, 8>(a, device, s, nb, metric);
        case 32: return deep ? launch_mfma2_inst<32, NG, 16>(a, device, s, nb, metric)
                             : launch_mfma2_inst<32, NG, 8>(a, device, s, nb, metric);
        // d = 768 / 1024 (e5-large, bge-m3 widths): one query group only — the 192 / 256 registers of
        // query fragments leave one wave per SIMD, like two groups at d = 512
        case 40: if (NG == 1) return launch_mfma2_inst<40, 1, 8>(a, device, s, nb, metric);   // (d = 640 / 896: round 6 — the exact pass
                 break;                                                                      //  behind the rescue tier at those widths)
        case 48: if (NG == 1) return deep ? launch_mfma2_inst<48, 1, 16>(a, device, s, nb, metric)
                                          : launch_mfma2_inst<48, 1, 8>(a, device, s, nb, metric);
                 break;
        case 56: if (NG == 1) return launch_mfma2_inst<56, 1, 8>(a, device, s, nb, metric);
                 break;
        case 64: if (NG == 1) return deep ? launch_mfma2_inst<64, 1, 16>(a, device, s, nb, metric)
                                          : launch_mfma2_inst<64, 1, 8>(a, device, s, nb, metric);
                 break;
        default: break;
    }
    return fail(MVDB_ERR_ARG, "no staged multi-query kernel for d = %d with %d query group(s)", KB * 16, NG);
}

// The exact re-run of uncertified queries (search_core): the same pass, enabled on the device by `*gate > gate_lo`.
// Returns the queries one launch takes (32 at d <= 512, 16 at d = 640 .. 1024), 0 when there is no kernel for d.
int mfma_gated_queries(const mvdb_index* idx) {
    const int KB = idx->d / 16;
    if (idx->d % 128 != 0) return 0;  // the staged kernel only
    return KB <= 32 ? 32 : KB <= 64 ? 16 : 0;
}
int launch_mfma2_gated(int KB, const MfmaScanArgs& a, int device, hipStream_t s, int* nb, const int* gate, int gate_lo,
                       const int* need = nullptr, int npasses = 1, int per_pass = 0, int rtot = 0, int64_t cand_stride = 0) {
    if (per_pass <= 0) per_pass = a.nq;   // one pass of a.nq queries
    if (rtot <= 0) rtot = a.nq;
    if (a.nq <= 16) {
        // one query group: with two, a pass of <= 16 queries issues twice the MFMAs it needs and is bound by them (8 queries
        // under a bitmap at 10M x 512: 3.67 ms against 3.0 unfiltered)
        switch (KB) {
            case 8: return launch_mfma2_gated_inst<8, 1, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
            case 16: return launch_mfma2_gated_inst<16, 1, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
            case 24: return launch_mfma2_gated_inst<24, 1, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
            case 32: return launch_mfma2_gated_inst<32, 1, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
            default: break;
        }
    }
    switch (KB) {
        case 8: return launch_mfma2_gated_inst<8, 2, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
        case 16: return launch_mfma2_gated_inst<16, 2, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
        case 24: return launch_mfma2_gated_inst<24, 2, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
        case 32: return launch_mfma2_gated_inst<32, 2, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
        case 40: return launch_mfma2_gated_inst<40, 1, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
        case 48: return launch_mfma2_gated_inst<48, 1, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
        case 56: return launch_mfma2_gated_inst<56, 1, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
        case 64: return launch_mfma2_gated_inst<64, 1, 8>(a, device, s, nb, gate, gate_lo, need, npasses, per_pass, rtot, cand_stride);
        default: break;
    }
    return fail(MVDB_ERR_ARG, "no gated multi-query kernel for d = %d", KB * 16);
}

template <int NG>
int launch_mfma_ng(int KB, const MfmaScanArgs& a, int device, hipStream_t s, int* nb) {
    switch (KB) {
        case 4: return launch_mfma_inst<4, NG>(a, device, s, nb);   // (d = 64: the one width the staged kernel — d % 128 == 0 — does not serve)
        default: return fail(MVDB_ERR_ARG, "no multi-query kernel for d = %d", KB * 16);
    }
}

// nq >= 64, k <= 16: compute-bound tiled GEMM + top-k (scan_mfma_kernels.hpp, last section)
bool mfma_path_ok(const mvdb_index* idx, int nq, int k, const int64_t* rows_dev);
// Fewest queries of a call for the GEMM-tiled exact scan (128 queries per launch, compute-bound: one launch costs ~4.7 single
// scans at 10M x 512): 104 where the fp32-MFMA passes exist (32 queries per HBM-bound pass), 6 at the widths that have neither
// those nor the certified pass (d % 16 == 0 outside 64 / 128 / ... / 1024: the retired bf16-split generation served them).
int gemm_min_nq(const mvdb_index* idx, int k) {
    return mfma_path_ok(idx, 2, k, nullptr) ? idx->kn.gemm_scan_min_nq : std::min(idx->kn.gemm_scan_min_nq, 6);
}
bool gemm_path_ok(const mvdb_index* idx, int nq, int k, const int64_t* rows_dev) {
    if (idx->kn.disable_gemm_scan) return false;
    if (k > kGemmScanMaxK || rows_dev || idx->metric != MVDB_METRIC_IP || nq < gemm_min_nq(idx, k)) return false;
    return idx->d % 16 == 0 && idx->ld == idx->d;
}

int launch_gemm_scan(const mvdb_index* idx, const float* q, int nq, int k, int64_t n, uint64_t* cand,
                     hipStream_t stream, int* nblocks_out, const int* gate = nullptr, int gate_lo = 0) {
    GemmScanArgs a;
    a.X = idx->X;
    a.n = n;
    a.ld = idx->ld;
    a.K = idx->d;
    a.q = q;
    a.nq = nq;
    a.k = k;
    a.cand = cand;
    const size_t lds = (size_t)2 * 2 * 16 * 132 * 4 + (size_t)4 * 64 * k * 8;
    MVDB_TRY(ensure_dynamic_lds((const void*)flat_scan_gemm_kernel, lds, idx->device));
    int nb = cached_occupancy((const void*)flat_scan_gemm_kernel, 256, lds, 1);
    nb = std::min(nb, std::max(1, kn().gemm_scan_blocks_per_cu));
    const int qtiles = (nq + 127) / 128;
    const int64_t ntiles = (n + 127) / 128;
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, (int64_t)device_cus(idx->device) * nb / qtiles));
    *nblocks_out = gx;
    if (gate) {
        MVDB_TRY(ensure_dynamic_lds((const void*)flat_scan_gemm_gated_kernel, lds, idx->device));
        hipLaunchKernelGGL(flat_scan_gemm_gated_kernel, dim3(gx, qtiles), dim3(256), lds, stream, a, gate, gate_lo);
        MVDB_HIP(hipGetLastError());
        return 0;
    }
    int slot = prof_begin("ip_scan_gemm", stream);
    hipLaunchKernelGGL(flat_scan_gemm_kernel, dim3(gx, qtiles), dim3(256), lds, stream, a);
    prof_end(slot, stream);
    MVDB_HIP(hipGetLastError());
    return 0;
}

// Chunks that held an uncertified query, counted ON THE DEVICE (split_plan_kernel): the host never reads a certification
// flag on the search path.  One counter per device, read (with a device synchronise) by mvdb_split_rerun_count().
std::mutex g_rerun_mu;
std::map<int, unsigned long long*> g_rerun_ctr;
unsigned long long* rerun_counter(int device) {
    std::lock_guard<std::mutex> lk(g_rerun_mu);
    auto it = g_rerun_ctr.find(device);
    if (it != g_rerun_ctr.end()) return it->second;
    unsigned long long* p = nullptr;
    // [0] refused chunks, [1] tiles the rescue launches were handed, [2] tiles they would have scanned without the tile flags
    if (hipMalloc((void**)&p, 3 * sizeof(*p)) != hipSuccess || hipMemset(p, 0, 3 * sizeof(*p)) != hipSuccess) return nullptr;
    g_rerun_ctr[device] = p;
    return p;
}

// k the certified pass serves: it re-scores 64 nominees, so its certificate compares the k-th exact score with the ~64th
// approximate one: at k = 32 the gap is still ~6 eps on
// exchangeable data (10M x 512 random rows: 32nd to 64th score 6.1e-3, eps 1.04e-3; 1.6e-2 at k = 10)
constexpr int kHalfMaxK = 32;
bool half_path_ok(const mvdb_index* idx);
int mfma_gated_queries(const mvdb_index* idx);

// L2 metric on the certified passes: they nominate by inner product, which ranks like the distance when the stored rows have
// (nearly) one norm — every row of the drop-in classes is normalised.  The certificate (topk_device.hpp: l2_certified) is
// exact for ANY spread, but a wide one would make most queries fail it; beyond 2^-10 relative the exact kernels serve.
bool l2_cert_ok(const mvdb_index* idx) {
    return idx->metric == MVDB_METRIC_L2 && idx->norm2_hi > 0.f && std::isfinite(idx->norm2_hi) &&
           idx->norm2_hi - idx->norm2_lo <= idx->norm2_hi * (1.0f / 1024.0f) && !idx->kn.disable_l2_cert;
}
// ... and rows of ANY norms where the nomination pass runs over the fp16 shadow: it then nominates by q.x - |x|^2 / 2 with
// per-row offsets kept beside the shadow (launch_half_pass: l2off).  True when a batch of nq queries over n rows would take
// that pass.
bool half_path_ok(const mvdb_index* idx);
int half_min_nq(const mvdb_index* idx, int64_t n);
bool l2_offsets_ok(const mvdb_index* idx, int nq, int64_t n) {
    return idx->metric == MVDB_METRIC_L2 && !idx->kn.disable_l2_cert && half_shadow_dim(idx->d) && idx->ld == idx->d &&
           !idx->kn.disable_half_shadow && !idx->xh_failed && half_path_ok(idx) && nq >= half_min_nq(idx, n);
}

// Fewest queries of a call that go to the certified passes.  Where the fp16 nomination pass streams the SHADOW of the rows
// (d = 256 / 384 / 512), one pass over n rows costs about what 0.55 n rows cost the exact scans, whatever the number of
// queries up to 128 — it beats the fp32-MFMA pass from TWO queries on once the corpus is big enough to bury its fixed cost
// (seed launch, phase merges, certification: ~0.1 ms).  Measured, device time per call, default / certified
// (profiles/r04_small_batch_crossover.jsonl): 10M x 512: 2 queries 2.93 / 1.61 ms, 32 queries 3.06 / 1.63; 1M rows: 2 queries
// 0.33 / 0.26, 13 queries 0.44 / 0.27; 100k rows: 2 queries 0.068 / 0.114, 8 queries 0.123 / 0.122, 13 queries 0.170 / 0.126.
// MVDB_SPLIT_SCAN_MIN_NQ overrides (the name dates from the retired split-precision passes).
// Tile flags cost a call that refuses nothing ~0.8 % (one compare + ballot per tile and wave, the flags' memset), so they are kept
// adaptively: split_plan_kernel mirrors a running count of refused certificates into a host-mapped word; when the word has moved
// since this index last looked, the next kTileFlagCalls certified calls keep flags (the call that FIRST meets refusals pays a
// whole-shadow rescue launch, the following ones walk tile lists).  MVDB_TILE_FLAGS=1: always, 0 (or MVDB_DISABLE_TILE_SKIP=1): never.
constexpr int kTileFlagCalls = 1024;
bool tile_flags_wanted(const mvdb_index* idx) {
    if (idx->kn.disable_tile_skip || idx->kn.tile_flags_mode == 0) return false;
    if (idx->kn.tile_flags_mode == 1) return true;
    if (!idx->rf_dev || !idx->rf_host.p) return false;
    const unsigned int v = *reinterpret_cast<volatile unsigned int*>(idx->rf_host.p);
    if (v != idx->rf_seen.load(std::memory_order_relaxed)) {
        idx->rf_seen.store(v, std::memory_order_relaxed);
        idx->rf_calls_left.store(kTileFlagCalls, std::memory_order_relaxed);
    }
    if (idx->rf_calls_left.load(std::memory_order_relaxed) <= 0) return false;
    idx->rf_calls_left.fetch_sub(1, std::memory_order_relaxed);
    return true;
}

thread_local bool tls_single_suspended = false;  // decided once per search (search_core), read by every routing question of that call
constexpr int kSingleWindow = 32, kSingleSuspend = 512;
// One decision per single-query search that asks for the shadow route: true = this call takes the exact scan.
bool single_route_suspended(const mvdb_index* idx) {
    if (idx->sq_suspend_left.load(std::memory_order_relaxed) > 0) {
        idx->sq_suspend_left.fetch_sub(1, std::memory_order_relaxed);
        return true;
    }
    const unsigned int calls = idx->sq_calls.fetch_add(1, std::memory_order_relaxed) + 1;
    if (calls - idx->sq_window_calls.load(std::memory_order_relaxed) >= (unsigned int)kSingleWindow && idx->sq_fail_host.p) {
        const unsigned int fails = *reinterpret_cast<volatile unsigned int*>(idx->sq_fail_host.p);
        if ((fails - idx->sq_window_fail.load(std::memory_order_relaxed)) * 2 >= (unsigned int)kSingleWindow) {
            idx->sq_suspend_left.store(kSingleSuspend, std::memory_order_relaxed);
            idx->sq_suspensions.fetch_add(1, std::memory_order_relaxed);
        }
        idx->sq_window_calls.store(calls, std::memory_order_relaxed);
        idx->sq_window_fail.store(fails, std::memory_order_relaxed);
    }
    return false;
}

int half_min_nq(const mvdb_index* idx, int64_t n) {
    if (idx->kn.split_scan_min_nq >= 0) return idx->kn.split_scan_min_nq;
    if (half_shadow_dim(idx->d) && idx->ld == idx->d && !idx->kn.disable_half_shadow && !idx->xh_failed) {
        // (opt-in, MVDB_SHADOW_SINGLE_QUERY=1: ONE query too — 10M x 512: 2.86 -> ~1.6 ms per query, certified like any batch; off
        //  by default: the single-query scan is the headline's exact fp32 kernel and its roofline is defined on the fp32 bytes)
        if (n >= 500000) return idx->kn.shadow_single_query && !tls_single_suspended ? 1 : 2;
        if (n >= 100000) return 8;
    }
    return 33;
}

// Does a call of nq queries over n rows go through the CERTIFIED pass (fp16 nomination over the shadow, half_scan.hip)?
// (Until round 6 a second certified generation — bf16 (hi, lo) split products over the fp32 rows, scan_split_kernels.hpp —
//  served the widths without an fp16 kernel; it is retired: those widths take the exact passes below.)
bool split_path_ok(const mvdb_index* idx, int nq, int k, const int64_t* rows_dev, int64_t n) {
    if (idx->kn.disable_split_scan || !half_path_ok(idx)) return false;
    if (nq < half_min_nq(idx, n) || (nq == 1 && !(idx->kn.shadow_single_query && !tls_single_suspended))) return false;
    if (rows_dev || (idx->metric != MVDB_METRIC_IP && !l2_cert_ok(idx) && !l2_offsets_ok(idx, nq, n))) return false;
    // (k > 16 also needs the gated fp32-MFMA pass for the exact re-runs: the GEMM-tiled scan keeps 16 results per query)
    if (k > kHalfMaxK || (k > kGemmScanMaxK && mfma_gated_queries(idx) <= 0)) return false;
    // rows of known, sane norm only: the bound scales with max|x|
    if (!(idx->row_norm_bound > 0.f) || !(idx->row_norm_bound < 1.0e30f)) return false;
    return true;
}

// Between the launches of a certified pass: merge the per-block lists of a launch (the seed launch: one tile per block) — and
// the running nominees of the launches before it — into each query's 16
// best approximate keys, and publish the 16th score as the admission floor of the main launch — every row of the
// global approximate top-16 scores at least that much, so the main pass only has to insert the few rows above it
// (without the floor each wave re-learns its threshold from scratch: ~7,000 LDS list inserts per wave at 10M rows,
// more time than the MFMAs).
__global__ __launch_bounds__(1024) void phase_fold_kernel(const uint64_t* __restrict__ keys, int nlists,
                                                          const uint64_t* prev, uint64_t* seed, float* __restrict__ thr0) {
    // prev: NULL, or the [nq][16] running nominees of the earlier phases (may alias `seed`: read before written)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = blockIdx.x;
    const int64_t total = (int64_t)nlists * kHalfKeep;
    const uint64_t* src = keys + (int64_t)qi * total;
    WaveTopK tk;
    tk.init(kHalfKeep);
    for (int64_t base = (int64_t)wave * 64; base < total; base += 1024) {
        const int64_t i = base + lane;
        tk.offer(i < total ? src[i] : 0ull);
    }
    if (prev && wave == 0) tk.offer(lane < kHalfKeep ? prev[(int64_t)qi * kHalfKeep + lane] : 0ull);
    __shared__ uint64_t sh[15 * 64];
    block_merge_topk(tk, sh, 16);
    if (wave == 0) {
        if (lane < kHalfKeep) seed[(int64_t)qi * kHalfKeep + lane] = tk.key;
        const uint64_t last = readlane_u64(tk.key, kHalfKeep - 1);
        if (lane == 0) thr0[qi] = last ? key_score(last) : -INFINITY;
    }
}

// ---- fp16 single-product nomination pass (half_scan.hip): 33+ queries per corpus pass where a kernel exists ----------
// (the pass streams the fp16 SHADOW of the rows: an index that may not keep one — option half_shadow = 0, or the shadow's
//  allocation failed — answers its batches on the exact fp32 passes)
bool half_path_ok(const mvdb_index* idx) {
    if (idx->kn.disable_half_scan || idx->kn.disable_half_shadow || idx->xh_failed) return false;
    return half_max_queries(idx->d) > 0 && half_shadow_dim(idx->d) && idx->ld == idx->d && half_xscale(idx->row_norm_bound) > 0.f;
}

const _Float16* ensure_shadow(const mvdb_index* idx, hipStream_t s, float xscale);
const float* ensure_offsets(const mvdb_index* idx, hipStream_t s);

int launch_half_pass(const mvdb_index* idx, Workspace* ws, const float* q, int nq, int nqpad, int k, int64_t n,
                     int64_t label_offset, float* D, int64_t* I, int* flag, int* failed, const uint32_t* mask = nullptr,
                     float* floor_out = nullptr, uint32_t* tflags = nullptr, int twords = 0, bool* flags_written = nullptr) {
    hipStream_t stream = ws->stream;
    if (flags_written) *flags_written = false;
    _Float16* qf = reinterpret_cast<_Float16*>(ws->qsplit.p);
    float* qnorm = ws->qnorm.p;
    float* floors = qnorm + nqpad;
    float* qinv = qnorm + 2 * nqpad;
    const float xscale = half_xscale(idx->row_norm_bound);
    MVDB_TRY(launch_half_queries(q, idx->ld, idx->d, nq, nqpad, xscale, qf, qnorm, qinv, stream));
    HalfScanArgs a;
    a.X = idx->X;
    a.n = n;
    a.ld = idx->ld;
    a.qf = qf;
    a.qinv = qinv;
    a.xscale = xscale;
    a.nq = nq;
    a.mask = mask;
    a.Xh = ensure_shadow(idx, stream, xscale);
    a.stats = idx->kn.split_stats ? reinterpret_cast<unsigned int*>(ws->flags.p) + (ws->flags.cap - 32) : nullptr;
    if (a.stats) MVDB_HIP(hipMemsetAsync(a.stats, 0, 8, stream));
    const int64_t ntiles = (n + 31) / 32;
    const int cus = device_cus(idx->device);
    uint64_t* seed_keys = ws->cand.p;                           // [nqpad][16] running nominees
    uint64_t* cand = ws->cand.p + (size_t)nqpad * kHalfKeep;   // [nq][lists][16]
    a.cand = cand;
    // SEED launch: the first min(tiles, CUs) tiles, one per block, every score dumped (32 keys per block and query);
    // phase_fold_kernel keeps each query's 16 best and publishes the 16th score as the first admission floor.
    const int64_t seed_tiles = std::min<int64_t>(ntiles, cus);
    int gx = 0;
    a.tile0 = 0;
    a.tile1 = seed_tiles;
    a.thr0 = nullptr;
    // L2 metric over rows of MIXED norms (round 4; rows of one norm keep the cheaper inner-product nomination with the norm-range
    // certificate): rows are nominated by q.x - |x|^2 / 2 with the per-row offsets kept beside the shadow (section 4.3e); the
    // seed is then the shadow kernel itself without a floor — one tile per block, its 16 best per query — because the fp32
    // seed kernel knows no offsets.
    const float* hn = idx->metric == MVDB_METRIC_L2 && a.Xh && !l2_cert_ok(idx) && !idx->kn.disable_l2_cert ? ensure_offsets(idx, stream)
                                                                                                          : nullptr;
    const bool l2off = hn != nullptr;
    if (l2off) {
        a.hn = hn;
        MVDB_TRY(launch_half_scan(idx->d, nqpad, false, a, idx->kn, idx->device, stream, &gx));
        hipLaunchKernelGGL(phase_fold_kernel, dim3(nq), dim3(1024), 0, stream, cand, gx, (const uint64_t*)nullptr, seed_keys, floors);
    } else {
        MVDB_TRY(launch_half_scan(idx->d, nqpad, true, a, idx->kn, idx->device, stream, &gx));
        hipLaunchKernelGGL(phase_fold_kernel, dim3(nq), dim3(1024), 0, stream, cand, 2 * gx, (const uint64_t*)nullptr,
                           seed_keys, floors);
    }
    MVDB_HIP(hipGetLastError());
    a.thr0 = floors;
    // The rest of the corpus is scanned in PHASES of growing size; between phases phase_fold_kernel folds the
    // per-block lists into the running 16 best and raises the floors (a block alone sees n / CUs rows: without the
    // refreshed floors its lists take hundreds of serial LDS inserts per wave).  Planned backwards: the LAST phase
    // covers at most `last_growth` times the rows before it — that leaves ~16 x last_growth candidates above its
    // floor for the 64-nominee certificate —, the earlier ones up to `growth` times.
    // Growth 16 / 6 (three main launches at 10M rows) up to 128 queries per pass; 6 / 4 (four) at 256, where a wave's insert
    // path stalls eight waves at the tile barrier and the rounds it takes scale with the sum of the growth factors
    // (profiles/r05_h16_phase_sweep.txt: 256 queries +3 - 5 % at 10M rows, +1 % at 1M; 128 / 32 queries lose 0.5 - 8 % to the extra
    // launch and merge, and keep 16 / 6).
    // (k <= 16 only: the floor the last phase starts from is the 16th best of the rows before it, and the k-th result has to
    //  beat it — with a quarter of the corpus before the last phase and k = 32, 16 of the 32 best fall into that quarter once
    //  in ~500 queries and the query is refused and re-run; a sixth: once in 1e5)
    const bool wide_pass = nqpad == 256 && k <= kHalfKeep;
    const int growth = std::max(2, idx->kn.half_phase_growth ? idx->kn.half_phase_growth : wide_pass ? 6 : 16);
    const int last_growth = std::max(2, idx->kn.half_last_growth ? idx->kn.half_last_growth : wide_pass ? 4 : 6);
    std::vector<int64_t> ends;
    for (int64_t b = ntiles, g = last_growth; b > seed_tiles; g = growth) {
        ends.push_back(b);
        b = (b + g - 1) / g;
        if (b <= 2 * seed_tiles) break;
    }
    int last_lists = 0;
    int64_t covered = seed_tiles;
    // Tile flags for the rescue pass (inner product, k <= 16; search_core decides).  With B = max|x| and e = half_eps(d) B: a
    // refused query's rescue launch admits rows with a(x) >= F = fl - e |q|, fl = t - m |q| (one ulp down), t the k-th fp32
    // re-score among the 64 nominees R, m = floor_margin = 2 d 2^-24 B.  At ANY moment of ANY main launch the threshold `thr` a
    // wave holds for the query (the larger of the phase's floor — the 16th best a() of the rows before the phase — and the 16th
    // of its list) is reached by at least 16 >= k candidates that are folded into R's pool, so R's k-th best a() is >= thr; a
    // nominee's re-score is >= a(x) - (e + m / 2) |q|; hence t >= thr - (e + m / 2) |q| and F >= thr - (2 e + 3 m / 2) |q| - ulp.
    // A tile whose best score stays below thr - flag_coef |q|, flag_coef = 2.25 e + 2 m, cannot hold an admitted row.
    if (tflags && !l2off && idx->metric == MVDB_METRIC_IP && k <= kHalfKeep) {
        a.tflags = tflags;
        a.twords = twords;
        a.flag_qn = qnorm;
        const double B = (double)idx->row_norm_bound;
        a.flag_coef = (float)((2.25 * half_eps(idx->d) + 4.0 * idx->d * std::ldexp(1.0, -24)) * B * (1.0 + 1e-6));
        if (flags_written) *flags_written = true;  // (the caller hands the rescue pass tile lists only if EVERY chunk's launches wrote flags)
    }
    for (size_t p = ends.size(); p-- > 0;) {
        a.tile0 = covered;
        a.tile1 = ends[p];
        MVDB_TRY(launch_half_scan(idx->d, nqpad, false, a, idx->kn, idx->device, stream, &gx));
        covered = ends[p];
        if (p > 0) {
            hipLaunchKernelGGL(phase_fold_kernel, dim3(nq), dim3(1024), 0, stream, cand, gx, seed_keys, seed_keys, floors);
            MVDB_HIP(hipGetLastError());
        } else {
            last_lists = gx;
        }
    }
    HalfCertifyArgs c;
    c.keys = cand;
    c.nlists = last_lists;
    c.base = seed_keys;
    c.thr0 = floors;
    c.X = idx->X;
    c.ld = idx->ld;
    c.d4 = idx->d4;
    c.q = q;
    c.qnorm = qnorm;
    c.eps = (float)(half_eps(idx->d) * (double)idx->row_norm_bound * (1.0 + 1e-6));  // rounded up
    c.k = k;
    c.label_offset = label_offset;
    c.D = D;
    c.I = I;
    c.uncertified = flag;
    c.failed = failed;
    c.l2 = l2off ? 2 : idx->metric == MVDB_METRIC_L2 ? 1 : 0;
    c.n2lo = idx->norm2_lo;
    c.floor_out = floor_out;
    c.floor_margin = (float)(2.0 * idx->d * std::ldexp(1.0, -24) * (double)idx->row_norm_bound * (1.0 + 1e-6));
    if (l2off) {
        // the stored |x|^2 / 2 (relative error < 2^-18, half_norms_kernel) and the fp32 subtraction a(x) - h (2^-23 of the larger)
        const double B = (double)idx->row_norm_bound;
        c.eps_h = (float)((std::ldexp(1.0, -18) + std::ldexp(1.0, -22)) * 0.5 * B * B * (1.0 + 1e-6));
        c.eps = (float)((double)c.eps + std::ldexp(1.0, -22) * B * (1.0 + 1e-6));
    }
    MVDB_TRY(launch_half_certify(c, nq, stream));
    if (a.stats) {
        unsigned int st[2] = {0, 0};
        MVDB_HIP(hipMemcpyAsync(st, a.stats, sizeof(st), hipMemcpyDeviceToHost, stream));
        MVDB_HIP(hipStreamSynchronize(stream));
        fprintf(stderr, "[mvdb half] list inserts %u, slow-path wave-rounds %u, %zu main launches\n", st[0], st[1], ends.size());
    }
    return 0;
}

bool mfma_path_ok(const mvdb_index* idx, int nq, int k, const int64_t* rows_dev) {
    if (idx->kn.disable_mfma_scan) return false;
    if (nq < 2 || k > kMaxFusedK || rows_dev) return false;
    if (idx->d % 16 || idx->ld != idx->d) return false;
    // squared L2 as |q|^2 + |x|^2 - 2 q.x: the staged kernel only (it sees whole rows go by)
    // (d <= 768: at d = 1024 the L2 form of the kernel spills; two 16-query groups only up to d = 384, same reason)
    if (idx->metric != MVDB_METRIC_IP && !(idx->d % 128 == 0 && idx->d <= 768 && !idx->kn.disable_l2_mfma)) return false;
    const int KB = idx->d / 16;
    return KB == 4 || KB == 8 || KB == 16 || KB == 24 || KB == 32 || KB == 40 || KB == 48 || KB == 56 || KB == 64;
}

// Core: queries already on the device (padded to ld), outputs on the device.  Enqueues on ws.stream.
// (Round 2 had an opt-in "dense subset" route here — score every row, pick the subset's scores out, radix-select over
// them — measured slower than the gather; round 3 replaced it by the bitmap-selected scan, mvdb_index_search_masked.)
// label of result = position p with rows = the ascending list of the mask's set bits: p = number of set bits below the row
__global__ __launch_bounds__(256) void mask_rank_kernel(int64_t* __restrict__ I, const uint64_t* __restrict__ mask) {
    __shared__ int part[256];
    const int64_t row = I[blockIdx.x];
    if (row < 0) return;  // block-uniform
    const int64_t words = row >> 6;
    int c = 0;
    for (int64_t w = threadIdx.x; w < words; w += 256) c += __popcll(mask[w]);
    if (threadIdx.x == 0 && (row & 63)) c += __popcll(mask[words] & ((1ull << (row & 63)) - 1ull));
    part[threadIdx.x] = c;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) part[threadIdx.x] += part[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) I[blockIdx.x] = part[0];
}

constexpr int kCoreTile = 1024;      // queries search_core answers per pass over its workspace
constexpr int64_t kShiftMaxRows = 8;           // most scattered deleted rows of a call the one-pass compaction takes (a run of any length qualifies)
constexpr size_t kShiftSideBytes = 64u << 20;   // ... and the most its side copies may occupy
constexpr size_t kTileFlagMaxBytes = 512u << 20;  // the most the certified pass's tile flags may occupy (queries of the call x tiles / 8)
constexpr int kGatedPassGroup = 8;   // exact re-run passes (of 16 / 32 compact queries) one gated launch walks: one list buffer of that many
int search_core(const mvdb_index* idx, Workspace* ws, const float* q_dev, int nq, int k,
                int normalize_q, const int64_t* rows_dev, int64_t m, int64_t label_offset,
                float* D_dev, int64_t* I_dev, bool allow_split = true, const uint64_t* mask_dev = nullptr) {
    KnobScope knobs(&idx->kn);
    hipStream_t s = ws->stream;
    // Workspace O(1) in nq: a call of more queries is answered kCoreTile at a time on the same stream-ordered workspace (the
    // compact re-run buffers, per-query floors and flags below are sized by the queries of ONE tile; a tile is a whole number of
    // every pass's chunk: 32 / 128 / 256).  [round-5 advisor: the re-run workspace grew with the batch — 4 GB at 16k queries]
    if (nq > kCoreTile) {
        for (int t0 = 0; t0 < nq; t0 += kCoreTile)
            MVDB_TRY(search_core(idx, ws, q_dev + (int64_t)t0 * idx->ld, std::min(kCoreTile, nq - t0), k, normalize_q, rows_dev, m, label_offset,
                                 D_dev + (int64_t)t0 * k, I_dev + (int64_t)t0 * k, allow_split, mask_dev));
        return 0;
    }
    if (allow_split) tls_single_suspended = nq == 1 && idx->kn.shadow_single_query && single_route_suspended(idx);
    // row list: its m entries; bitmap: the first m rows when the caller says how many rows the bitmap covers (a resident
    // row set built before later appends), else every row
    const int64_t n = rows_dev ? m : (mask_dev && m > 0 ? std::min<int64_t>(m, idx->n) : idx->n);
    // Bitmap-selected rows, several queries: the passes that take a bitmap are the fp16 nomination pass (33+ queries; the
    // bit is looked at where a row is about to be nominated) and the staged fp32-MFMA pass (2..32 queries, and the exact
    // re-runs of uncertified queries); everything else answers a bitmap one query at a time.
    const uint32_t* mask32 = reinterpret_cast<const uint32_t*>(mask_dev);
    const bool masked_batch = mask_dev && nq >= 2 && k <= kMaxFusedK && idx->metric == MVDB_METRIC_IP && idx->ld == idx->d &&
                              mfma_gated_queries(idx) > 0 && !idx->kn.disable_masked_batch;
    // L2 under a bitmap (round 4): only through the fp16 nomination pass (its gate looks the bit up; the uncertified queries'
    // re-run is the single-query scan, which takes the bitmap too); smaller batches answer one query at a time
    const bool l2_masked_half = mask_dev && nq >= 2 && k <= kMaxFusedK && idx->metric == MVDB_METRIC_L2 && idx->ld == idx->d &&
                                !idx->kn.disable_masked_batch && (l2_cert_ok(idx) || l2_offsets_ok(idx, nq, n));
    if (mask_dev) {
        rows_dev = nullptr;
        if (!(masked_batch || l2_masked_half) || !half_path_ok(idx) || nq < half_min_nq(idx, n)) allow_split = false;
    }
    // the other multi-query passes take neither a row list nor a bitmap
    const int64_t* restricted = mask_dev ? reinterpret_cast<const int64_t*>(mask_dev) : rows_dev;
    if (n == 0) {
        const int64_t total = (int64_t)nq * k;
        hipLaunchKernelGGL(fill_missing_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           s, D_dev, I_dev, total, idx->metric);
        MVDB_HIP(hipGetLastError());
        return 0;
    }
    ScanArgs a;
    a.X = idx->X;
    a.n = n;
    a.ld = idx->ld;
    a.d4 = idx->d4;
    a.q = q_dev;
    a.normalize_q = normalize_q;
    a.k = k;
    a.rows = rows_dev;
    a.mask = mask_dev;
    a.cand = nullptr;
    a.scores = nullptr;

    // Batches (2+ queries where the corpus buries the pass's fixed cost, half_min_nq), k <= 32, rows of known norm, a width the
    // fp16 kernels serve: ONE fp16 product over the shadow nominates, fp32 re-scores decide, a worst-case bound certifies
    // (half_scan.hip); queries whose certificate is refused are re-run below.
    if (allow_split && split_path_ok(idx, nq, k, rows_dev, n) && ensure_shadow(idx, s, half_xscale(idx->row_norm_bound))) {
        const float* qsrc = q_dev;
        if (normalize_q) {
            MVDB_TRY(ws->qn.reserve((size_t)nq * idx->ld));
            MVDB_HIP(hipMemcpyAsync(ws->qn.p, q_dev, (size_t)nq * idx->ld * sizeof(float),
                                    hipMemcpyDeviceToDevice, s));
            MVDB_TRY(normalize_range(idx, ws->qn.p, nq, s));
            qsrc = ws->qn.p;
        }
        // chunk plan: 128 / 256 queries per pass while at least min_nq remain; what is left takes the exact passes
        const int min_nq = half_min_nq(idx, n);
        const int chunk = half_max_queries(idx->d);
        std::vector<std::pair<int, int>> plan;  // (first query, count)
        int q0 = 0;
        // (L2 over rows of mixed norms: the pass nominates by q.x - |x|^2 / 2 with per-row offsets beside the shadow; rows of one
        //  norm by inner product, which ranks like the distance then)
        const bool ip_ranks = idx->metric == MVDB_METRIC_IP || l2_cert_ok(idx);
        while (nq - q0 >= min_nq && (ip_ranks || l2_offsets_ok(idx, nq - q0, n))) {
            plan.emplace_back(q0, std::min(nq - q0, chunk));
            q0 += plan.back().second;
        }
        const int nchunks = (int)plan.size();
        if (nchunks > 0) {
            MVDB_TRY(ws->qsplit.reserve((size_t)std::max(2 * 128, chunk) * idx->d));
            MVDB_TRY(ws->qnorm.reserve((size_t)std::max(256, 3 * chunk)));  // |q|, admission floors, 1 / scale
            MVDB_TRY(ws->flags.reserve((size_t)std::max(nchunks, 64) + 32));  // + diagnostics counters in the last 32 slots
            MVDB_TRY(ws->cand.reserve((size_t)std::max(128, chunk) * (scan_grid_upper_bound(idx->device) + 1) * kHalfKeep));
            MVDB_TRY(ws->qfail.reserve((size_t)q0));
            MVDB_TRY(ws->qfloor.reserve((size_t)2 * q0 + 128));  // [q0] per query | [q0 + 128] per compact slot
            MVDB_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ws->qfloor.p), 0xff800000u, (size_t)q0, s));  // -inf: no floor
            MVDB_HIP(hipMemsetAsync(ws->flags.p, 0, (size_t)nchunks * sizeof(int), s));
            MVDB_HIP(hipMemsetAsync(ws->qfail.p, 0, (size_t)q0 * sizeof(int), s));
            // Tile flags (round 6): the main launches note, per query, which 32-row tiles came near its running threshold; should the
            // query be refused, its rescue launch walks those tiles only (clustered 10M x 512, 256 per call: 2.6 % of the shadow).  Inner product, k <= 16 (the floors are 16th-best scores), from ~400k rows on (below, the rescue launch is
            // short and the flags' memset is not), at most 512 MiB of flags — and only while the index has been refusing
            // certificates (tile_flags_wanted): a corpus that certifies everything never pays for them.
            const int64_t ntiles_all = (n + 31) / 32;
            const int twords = (int)((ntiles_all + 31) / 32);
            const bool tile_flags = idx->metric == MVDB_METRIC_IP && k <= kHalfKeep && ntiles_all >= idx->kn.tile_flag_min_tiles && tile_flags_wanted(idx) &&
                                    !idx->kn.disable_rescue && !idx->kn.disable_rerun_floor && half_rescue_dim(idx->d) &&
                                    (size_t)q0 * twords * sizeof(uint32_t) <= kTileFlagMaxBytes;
            if (tile_flags) {
                MVDB_TRY(ws->tflags.reserve((size_t)q0 * twords));
                MVDB_HIP(hipMemsetAsync(ws->tflags.p, 0, (size_t)q0 * twords * sizeof(uint32_t), s));
            }
            bool flags_ok = tile_flags;  // ... and every chunk's main launches did write them (launch_half_pass decides per chunk)
            for (int c = 0; c < nchunks; ++c) {
                const int c0 = plan[c].first, take = plan[c].second;
                bool wrote = false;
                MVDB_TRY(launch_half_pass(idx, ws, qsrc + (int64_t)c0 * idx->ld, take, half_chunk_queries(idx->d, take), k, n, label_offset,
                                          D_dev + (int64_t)c0 * k, I_dev + (int64_t)c0 * k, ws->flags.p + c, ws->qfail.p + c0, mask32,
                                          ws->qfloor.p + c0, tile_flags ? ws->tflags.p + (size_t)c0 * twords : nullptr, twords, &wrote));
                flags_ok = flags_ok && wrote;
            }
            // ---- uncertified queries: re-run on the exact kernels WITHOUT the host ever learning which they were ----------
            // split_plan_kernel compacts the failed queries (ascending) and publishes their number nb; the exact passes below
            // are launched unconditionally over the compact batch and enabled on the device: a launch whose query range
            // starts at or beyond nb returns at once (a few microseconds each when every query certified — the usual case).
            //   compact queries [0, 64): two 32-query fp32-MFMA passes (one corpus pass each: a handful of duplicate-heavy
            //                            neighbourhoods in a batch cost about one pass);
            //   the rest, in 128s:       the GEMM-tiled exact scan (a whole chunk failing: duplicate-heavy data).
            // No stream synchronise, no device-to-host copy: the call can be captured into a hipGraph (after one eager call
            // has sized the workspace) and an encoder -> search chain stays one enqueue.
            unsigned long long* ctr = rerun_counter(idx->device);
            if (!ctr) return fail(MVDB_ERR_OOM, "device allocation for the re-run counter failed");
            const int R = q0;  // most queries that can fail
            MVDB_TRY(ws->nfail.reserve(1));
            MVDB_TRY(ws->relabel.reserve((size_t)R * (1 + k)));
            MVDB_TRY(ws->requery.reserve((size_t)(R + 128) * idx->ld + (size_t)R * k));  // + one query tile of slack for the scans
            int64_t* map = ws->relabel.p;
            int64_t* It = map + R;
            float* qc = ws->requery.p;
            float* Dt = qc + (size_t)(R + 128) * idx->ld;
            unsigned int *sq_dev = nullptr, *sq_host = nullptr;
            if (nq == 1 && idx->kn.shadow_single_query) {  // the opt-in single-query route reports its refusals (single_route_suspended)
                std::lock_guard<std::mutex> lk(idx->shadow_mu);
                if (!idx->sq_fail_dev && hipMalloc((void**)&idx->sq_fail_dev, sizeof(unsigned int)) == hipSuccess) {
                    if (hipMemsetAsync(idx->sq_fail_dev, 0, sizeof(unsigned int), s) != hipSuccess || idx->sq_fail_host.reserve(64) != 0) {
                        (void)hipFree(idx->sq_fail_dev);
                        idx->sq_fail_dev = nullptr;
                    } else {
                        *reinterpret_cast<volatile unsigned int*>(idx->sq_fail_host.p) = 0u;
                    }
                }
                (void)hipGetLastError();
                if (idx->sq_fail_dev && idx->sq_fail_host.p) {
                    sq_dev = idx->sq_fail_dev;
                    sq_host = reinterpret_cast<unsigned int*>(idx->sq_fail_host.p);
                }
            }
            unsigned int* rf_dev = idx->rf_dev && idx->rf_host.p ? idx->rf_dev : nullptr;
            hipLaunchKernelGGL(split_plan_kernel, dim3(1), dim3(64), 0, s, (const int*)ws->flags.p, nchunks, (const int*)ws->qfail.p, q0,
                               map, ws->nfail.p, ctr, sq_dev, (volatile unsigned int*)sq_host, rf_dev,
                               (volatile unsigned int*)(rf_dev ? idx->rf_host.p : nullptr));
            {
                const int64_t rows = R + 128, gtotal = rows * (idx->ld / 4);
                const int ggrid = (int)std::min<int64_t>((gtotal + 255) / 256, (int64_t)device_cus(idx->device) * 8);
                hipLaunchKernelGGL(gather_failed_kernel, dim3(ggrid), dim3(256), 0, s, qc, qsrc, (const int64_t*)map,
                                   (const int*)ws->nfail.p, rows, idx->ld, (const float*)ws->qfloor.p, ws->qfloor.p + q0);
            }
            MVDB_HIP(hipGetLastError());
            const int per_pass = mfma_gated_queries(idx);
            const int KB = idx->d / 16;
            int off = 0;
            // ---- the RESCUE pass (half_scan.hip): refused queries once more over the shadow, 128 at a time, every row above
            // the query's floor — what the k-th exact score (L2: distance) of its nominees admits, less the nomination error —
            // kept and re-scored in fp32; what it answers the exact passes skip.  need_per = compact queries per `need` word:
            // the queries of one exact pass (inner product: a gated fp32-MFMA pass of 32 / 16), or 1 (L2: one gated scan each).
            auto rescue = [&](int need_per, const int** need_out) -> int {
                *need_out = nullptr;
                const float xs = half_xscale(idx->row_norm_bound);
                if (need_per <= 0 || kRescueQueries % need_per != 0 || !half_rescue_dim(idx->d) || idx->ld != idx->d || k > kRescueKeep ||
                    idx->kn.disable_rescue || idx->kn.disable_rerun_floor || !(xs > 0.f))
                    return 0;
                const _Float16* Xh = ensure_shadow(idx, s, xs);
                if (!Xh) return 0;
                // (L2: the same nomination form the certified pass used — per-row offsets where the rows' norms differ)
                const bool l2 = idx->metric == MVDB_METRIC_L2;
                const float* hn = l2 && !l2_cert_ok(idx) && !idx->kn.disable_l2_cert ? ensure_offsets(idx, s) : nullptr;
                if (l2 && !l2_cert_ok(idx) && !hn) return 0;
                const int grid_ub = device_cus(idx->device) * kRescueBlocksPerCu;  // the rescue launch: one or two workgroups per CU
                const int slots = (R + kRescueQueries - 1) / kRescueQueries;
                const size_t nneed = (size_t)(R + kRescueQueries) / need_per + 8;
                const size_t nwords = nneed + slots;  // ... and the tile lists' lengths behind the need words (one memset)
                MVDB_TRY(ws->need.reserve(nwords));
                MVDB_TRY(ws->tlist.reserve((size_t)slots * ntiles_all));
                MVDB_TRY(ws->cand.reserve((size_t)kRescueQueries * (grid_ub + 1) * kRescueKeep));
                MVDB_TRY(ws->qsplit.reserve((size_t)2 * kRescueQueries * idx->d));
                MVDB_TRY(ws->qnorm.reserve((size_t)3 * kRescueQueries));
                MVDB_HIP(hipMemsetAsync(ws->need.p, 0, nwords * sizeof(int), s));
                _Float16* qf = reinterpret_cast<_Float16*>(ws->qsplit.p);
                float* qn2 = ws->qnorm.p;
                float* qinv = qn2 + 2 * kRescueQueries;
                float eps = (float)(half_eps(idx->d) * (double)idx->row_norm_bound * (1.0 + 1e-6));
                if (hn) eps = (float)((double)eps + std::ldexp(1.0, -22) * (double)idx->row_norm_bound * (1.0 + 1e-6));  // (launch_half_pass: the subtraction)
                {   // the launches' tile lists: what the refused queries' flags name (no flags: every tile)
                    RescueTilesArgs ta;
                    ta.tflags = flags_ok ? ws->tflags.p : nullptr;
                    ta.twords = twords;
                    ta.map = map;
                    ta.nfail = ws->nfail.p;
                    ta.seed_tiles = std::min<int64_t>(ntiles_all, device_cus(idx->device));  // (launch_half_pass: the seed launch's tiles)
                    ta.ntiles = ntiles_all;
                    ta.lists = ws->tlist.p;
                    ta.counts = ws->need.p + nneed;
                    ta.stats = ctr + 1;
                    MVDB_TRY(launch_rescue_tiles(ta, slots, s));
                }
                for (int off2 = 0; off2 < R; off2 += kRescueQueries) {
                    MVDB_TRY(launch_half_queries(qc + (int64_t)off2 * idx->ld, idx->ld, idx->d, kRescueQueries, kRescueQueries, xs, qf, qn2,
                                                 qinv, s));
                    HalfScanArgs ra;
                    ra.X = idx->X;
                    ra.n = n;
                    ra.ld = idx->ld;
                    ra.qf = qf;
                    ra.qinv = qinv;
                    ra.xscale = xs;
                    ra.nq = std::min(kRescueQueries, R - off2);  // (slots past the call's queries can never be live)
                    ra.mask = mask32;
                    ra.Xh = Xh;
                    ra.hn = hn;
                    ra.stats = nullptr;
                    ra.cand = ws->cand.p;
                    ra.tile0 = 0;
                    ra.tile1 = (n + 31) / 32;
                    ra.thr0 = ws->qfloor.p + q0 + off2;
                    ra.thr_eps = eps;
                    ra.thr_qn = qn2;
                    ra.gate = ws->nfail.p;
                    ra.gate_lo = off2;
                    ra.tile_list = ws->tlist.p + (size_t)(off2 / kRescueQueries) * ntiles_all;
                    ra.tile_count = ws->need.p + nneed + off2 / kRescueQueries;
                    int gx = 0;
                    MVDB_TRY(launch_half_rescue_scan(idx->d, ra, idx->device, s, &gx));
                    HalfRescueArgs rc;
                    rc.keys = ws->cand.p;
                    rc.nlists = gx;
                    rc.X = idx->X;
                    rc.ld = idx->ld;
                    rc.d4 = idx->d4;
                    rc.q = qc + (int64_t)off2 * idx->ld;
                    rc.k = k;
                    rc.label_offset = label_offset;
                    rc.D = Dt + (int64_t)off2 * k;
                    rc.I = It + (int64_t)off2 * k;
                    rc.gate = ws->nfail.p;
                    rc.gate_lo = off2;
                    rc.need = ws->need.p + off2 / need_per;
                    rc.per_pass = need_per;
                    rc.l2 = l2 ? 1 : 0;
                    MVDB_TRY(launch_half_rescue_certify(rc, s));
                }
                *need_out = ws->need.p;
                return 0;
            };
            if (idx->metric == MVDB_METRIC_L2) {
                // L2: the rescue pass, then — for what it could not hold — the exact single-query scan (sum (q - x)^2 directly),
                // 32 compact queries per launch, each query's blocks enabled on the device (refused count and `need` word)
                const int* need = nullptr;
                MVDB_TRY(rescue(1, &need));
                const int per = 32;
                MVDB_TRY(ws->cand.reserve((size_t)per * scan_grid_upper_bound(idx->device) * k));
                for (; off < R; off += per) {
                    const int take = std::min(per, R - off);
                    ScanArgs ga = a;
                    ga.q = qc + (int64_t)off * idx->ld;
                    ga.normalize_q = 0;
                    ga.cand = ws->cand.p;
                    ga.gate = ws->nfail.p;
                    ga.gate_lo = off;
                    ga.need = need ? need + off : nullptr;
                    int nblocks = 0;
                    MVDB_TRY(launch_scan(idx->metric, kModeTopK, ga, take, idx->device, s, &nblocks));
                    MergeArgs mg;
                    mg.keys = ws->cand.p;
                    mg.nlists = nblocks;
                    mg.k = k;
                    mg.metric = idx->metric;
                    mg.label_offset = label_offset;
                    mg.D = Dt + (int64_t)off * k;
                    mg.I = It + (int64_t)off * k;
                    mg.gate = ws->nfail.p;
                    mg.gate_lo = off;
                    if (need) {   // one word per query: "pass" = query, its lists nblocks * k keys after the query before
                        mg.need = need + off;
                        mg.per_pass = 1;
                        mg.pass_stride = (int64_t)nblocks * k;
                    }
                    hipLaunchKernelGGL(merge_keys_kernel, dim3(take), dim3(kMergeThreads), 0, s, mg);
                    MVDB_HIP(hipGetLastError());
                }
                off = R;
            } else if (per_pass > 0 && !idx->kn.disable_mfma_scan) {
                const int* need = nullptr;
                MVDB_TRY(rescue(per_pass, &need));
                MVDB_TRY(ws->cand.reserve((size_t)32 * scan_grid_upper_bound(idx->device) * k));
                // (the GEMM scan keeps k <= 16 and takes no bitmap: those re-runs are all fp32-MFMA passes)
                // every refused query through the gated fp32-MFMA pass, 32 at a time (launches beyond the refused count return at
                // once).  Until round 5 the batch went to the 128-query GEMM-tiled scan after two such passes — 20-25 ms per 128
                // queries at 10M x 512 where four MFMA passes take 14.
                const int max_passes = (R + per_pass - 1) / per_pass;
                {
                    // ONE launch walks up to kGatedPassGroup passes on the device (pass p: compact queries [p per_pass, ...); each
                    // enabled by the refused count and its `need` word), ONE merge launch covers the group's compact queries; the
                    // groups share one list buffer in stream order (so the lists are O(1) in the batch: 8 passes x 5 MB at k = 10)
                    const int64_t cand_stride = (int64_t)per_pass * scan_grid_upper_bound(idx->device) * k;
                    MVDB_TRY(ws->cand.reserve((size_t)std::min(max_passes, kGatedPassGroup) * cand_stride));
                    for (int p0 = 0; p0 < max_passes; p0 += kGatedPassGroup) {
                        const int np = std::min(kGatedPassGroup, max_passes - p0), qoff = p0 * per_pass;
                        MfmaScanArgs ma;
                        ma.X = idx->X;
                        ma.n = n;
                        ma.ld = idx->ld;
                        ma.q = qc + (int64_t)qoff * idx->ld;
                        ma.nq = std::min(per_pass, R - qoff);
                        ma.k = k;
                        ma.cand = ws->cand.p;
                        ma.mask = mask32;
                        ma.thr0 = idx->kn.disable_rerun_floor ? nullptr : ws->qfloor.p + q0 + qoff;
                        int nblocks = 0;
                        MVDB_TRY(launch_mfma2_gated(KB, ma, idx->device, s, &nblocks, ws->nfail.p, qoff, need ? need + p0 : nullptr, np, per_pass,
                                                    R - qoff, cand_stride));
                        MergeArgs mg;
                        mg.keys = ws->cand.p;
                        mg.nlists = nblocks;
                        mg.k = k;
                        mg.metric = idx->metric;
                        mg.label_offset = label_offset;
                        mg.D = Dt + (int64_t)qoff * k;
                        mg.I = It + (int64_t)qoff * k;
                        mg.gate = ws->nfail.p;
                        mg.gate_lo = qoff;
                        mg.need = need ? need + p0 : nullptr;
                        mg.per_pass = per_pass;
                        mg.pass_stride = cand_stride;
                        hipLaunchKernelGGL(merge_keys_kernel, dim3(std::min(np * per_pass, R - qoff)), dim3(kMergeThreads), 0, s, mg);
                        MVDB_HIP(hipGetLastError());
                    }
                    off = R;
                }
            }
            if (off < R && mask_dev) return fail(MVDB_ERR_ARG, "internal: bitmap re-run left to the GEMM scan");
            if (off < R) MVDB_TRY(ws->cand.reserve((size_t)128 * scan_grid_upper_bound(idx->device) * k));
            while (off < R) {
                const int take = std::min(128, R - off);
                int nblocks = 0;
                MVDB_TRY(launch_gemm_scan(idx, qc + (int64_t)off * idx->ld, take, k, n, ws->cand.p, s, &nblocks, ws->nfail.p, off));
                MergeArgs mg;
                mg.keys = ws->cand.p;
                mg.nlists = nblocks;
                mg.k = k;
                mg.metric = idx->metric;
                mg.label_offset = label_offset;
                mg.D = Dt + (int64_t)off * k;
                mg.I = It + (int64_t)off * k;
                mg.gate = ws->nfail.p;
                mg.gate_lo = off;
                hipLaunchKernelGGL(merge_keys_kernel, dim3(take), dim3(kMergeThreads), 0, s, mg);
                MVDB_HIP(hipGetLastError());
                off += take;
            }
            hipLaunchKernelGGL(scatter_failed_kernel, dim3((unsigned)(((int64_t)R * k + 255) / 256)), dim3(256), 0, s, (const float*)Dt,
                               (const int64_t*)It, (const int64_t*)map, (const int*)ws->nfail.p, (int64_t)R, k, D_dev, I_dev);
            MVDB_HIP(hipGetLastError());
        }
        if (q0 == nq) return 0;
        return search_core(idx, ws, qsrc + (int64_t)q0 * idx->ld, nq - q0, k, 0, rows_dev, m, label_offset,
                           D_dev + (int64_t)q0 * k, I_dev + (int64_t)q0 * k, false, mask_dev);
    }

    if (masked_batch) {
        // ---- bitmap-selected rows, 2+ queries: staged fp32-MFMA passes of up to 32 (d <= 512) / 16 queries, exact ----------
        const float* qsrc = q_dev;
        if (normalize_q) {
            MVDB_TRY(ws->qn.reserve((size_t)nq * idx->ld));
            MVDB_HIP(hipMemcpyAsync(ws->qn.p, q_dev, (size_t)nq * idx->ld * sizeof(float), hipMemcpyDeviceToDevice, s));
            MVDB_TRY(normalize_range(idx, ws->qn.p, nq, s));
            qsrc = ws->qn.p;
        }
        const int per_pass = mfma_gated_queries(idx);
        MVDB_TRY(ws->cand.reserve((size_t)32 * scan_grid_upper_bound(idx->device) * k));
        for (int q0 = 0; q0 < nq; q0 += per_pass) {
            const int take = std::min(per_pass, nq - q0);
            MfmaScanArgs ma;
            ma.X = idx->X;
            ma.n = n;
            ma.ld = idx->ld;
            ma.q = qsrc + (int64_t)q0 * idx->ld;
            ma.nq = take;
            ma.k = k;
            ma.cand = ws->cand.p;
            ma.mask = mask32;
            int nblocks = 0;
            int slot = prof_begin("ip_scan_mfma_masked", s);
            MVDB_TRY(launch_mfma2_gated(idx->d / 16, ma, idx->device, s, &nblocks, nullptr, 0));
            prof_end(slot, s);
            MergeArgs mg;
            mg.keys = ws->cand.p;
            mg.nlists = nblocks;
            mg.k = k;
            mg.metric = idx->metric;
            mg.label_offset = label_offset;
            mg.D = D_dev + (int64_t)q0 * k;
            mg.I = I_dev + (int64_t)q0 * k;
            hipLaunchKernelGGL(merge_keys_kernel, dim3(take), dim3(kMergeThreads), 0, s, mg);
            MVDB_HIP(hipGetLastError());
        }
        return 0;
    }

    // Large batches are cut into chunks: >= 104 queries left -> one 128-query GEMM-tiled launch (compute-
    // bound, 13.7 ms at 10M x 512), fewer -> 32-query MFMA passes (4.0 ms each); measured crossover ~100.
    if (gemm_path_ok(idx, nq, k, restricted)) {
        const float* qsrc = q_dev;
        if (normalize_q) {
            MVDB_TRY(ws->qn.reserve((size_t)nq * idx->ld));
            MVDB_HIP(hipMemcpyAsync(ws->qn.p, q_dev, (size_t)nq * idx->ld * sizeof(float),
                                    hipMemcpyDeviceToDevice, s));
            MVDB_TRY(normalize_range(idx, ws->qn.p, nq, s));
            qsrc = ws->qn.p;
        }
        MVDB_TRY(ws->cand.reserve((size_t)128 * scan_grid_upper_bound(idx->device) * k));
        const int min_nq = gemm_min_nq(idx, k);
        int q0 = 0;
        while (nq - q0 >= min_nq) {
            const int take = std::min(nq - q0, 128);
            int nblocks = 0;
            MVDB_TRY(launch_gemm_scan(idx, qsrc + (int64_t)q0 * idx->ld, take, k, n, ws->cand.p, s, &nblocks));
            MergeArgs mg;
            mg.keys = ws->cand.p;
            mg.nlists = nblocks;
            mg.k = k;
            mg.metric = idx->metric;
            mg.label_offset = label_offset;
            mg.D = D_dev + (int64_t)q0 * k;
            mg.I = I_dev + (int64_t)q0 * k;
            hipLaunchKernelGGL(merge_keys_kernel, dim3(take), dim3(kMergeThreads), 0, s, mg);
            MVDB_HIP(hipGetLastError());
            q0 += take;
        }
        if (q0 == nq) return 0;
        // remainder: already-normalised queries through the paths below (a re-run of uncertified queries stays exact)
        return search_core(idx, ws, qsrc + (int64_t)q0 * idx->ld, nq - q0, k, 0, rows_dev, m, label_offset,
                           D_dev + (int64_t)q0 * k, I_dev + (int64_t)q0 * k, allow_split);
    }

    if (mfma_path_ok(idx, nq, k, restricted)) {
        // ---- several queries per corpus pass on the fp32 matrix cores ---------------------------
        const float* qsrc = q_dev;
        if (normalize_q) {
            MVDB_TRY(ws->qn.reserve((size_t)nq * idx->ld));
            MVDB_HIP(hipMemcpyAsync(ws->qn.p, q_dev, (size_t)nq * idx->ld * sizeof(float),
                                    hipMemcpyDeviceToDevice, s));
            MVDB_TRY(normalize_range(idx, ws->qn.p, nq, s));
            qsrc = ws->qn.p;
        }
        const int grid_ub = scan_grid_upper_bound(idx->device);
        MVDB_TRY(ws->cand.reserve((size_t)32 * grid_ub * k));
        for (int q0 = 0; q0 < nq;) {
            const int left = nq - q0;
            // staged kernel: 32 queries per pass (two query groups share each B fragment); the v1
            // kernel with two groups runs at one wave per SIMD and loses to two 16-query passes
            const bool staged = idx->d % 128 == 0;
            const bool two_groups = left > 16 && idx->d <= (idx->metric == MVDB_METRIC_IP ? 512 : 384) && (idx->kn.mfma_ng2 >= 0 ? idx->kn.mfma_ng2 != 0 : staged);
            const int take = two_groups ? std::min(left, 32) : std::min(left, 16);
            MfmaScanArgs ma;
            ma.X = idx->X;
            ma.n = n;
            ma.ld = idx->ld;
            ma.q = qsrc + (int64_t)q0 * idx->ld;
            ma.nq = take;
            ma.k = k;
            ma.cand = ws->cand.p;
            int nblocks = 0;
            const int KB = idx->d / 16;
            if (take > 16 && staged)
                MVDB_TRY(launch_mfma2<2>(KB, ma, idx->device, s, &nblocks, idx->metric));
            else if (take > 16)
                MVDB_TRY(launch_mfma_ng<2>(KB, ma, idx->device, s, &nblocks));
            else if (staged)
                MVDB_TRY(launch_mfma2<1>(KB, ma, idx->device, s, &nblocks, idx->metric));  // LDS-DMA staged, coalesced
            else
                MVDB_TRY(launch_mfma_ng<1>(KB, ma, idx->device, s, &nblocks));
            MergeArgs mg;
            mg.keys = ws->cand.p;
            mg.nlists = nblocks;
            mg.k = k;
            mg.metric = idx->metric;
            mg.label_offset = label_offset;
            mg.D = D_dev + (int64_t)q0 * k;
            mg.I = I_dev + (int64_t)q0 * k;
            hipLaunchKernelGGL(merge_keys_kernel, dim3(take), dim3(kMergeThreads), 0, s, mg);
            MVDB_HIP(hipGetLastError());
            q0 += take;
        }
        return 0;
    }

    if (k <= kMaxFusedK) {
        MVDB_TRY(ws->cand.reserve((size_t)nq * scan_grid_upper_bound(idx->device) * k));
        a.cand = ws->cand.p;
        int nblocks = 0;
        MVDB_TRY(launch_scan(idx->metric, kModeTopK, a, nq, idx->device, s, &nblocks));
        MergeArgs ma;
        ma.keys = ws->cand.p;
        ma.nlists = nblocks;
        ma.k = k;
        ma.metric = idx->metric;
        ma.label_offset = label_offset;
        ma.D = D_dev;
        ma.I = I_dev;
        hipLaunchKernelGGL(merge_keys_kernel, dim3(nq), dim3(kMergeThreads), 0, s, ma);
        MVDB_HIP(hipGetLastError());
        return 0;
    }

    // ---- large k: scores -> radix select -> sort ------------------------------------------------------
    int nblocks = 0;
    MVDB_TRY(ws->scores.reserve((size_t)nq * n));
    a.scores = ws->scores.p;
    MVDB_TRY(launch_scan(idx->metric, kModeScores, a, nq, idx->device, s, &nblocks));
    const float* sc_base = ws->scores.p;
    const int64_t k_eff = std::min<int64_t>(k, n);
    const int64_t P = pow2ceil(std::max<int64_t>(k, 2));
    MVDB_TRY(ws->selkeys.reserve((size_t)P));
    const int cus = device_cus(idx->device);
    const int sel_grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, cus * 8));
    for (int qi = 0; qi < nq; ++qi) {
        const float* sc = sc_base + (int64_t)qi * n;
        hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, s, ws->st, (uint64_t)k_eff);
        for (int shift = 56; shift >= 0; shift -= 8) {
            hipLaunchKernelGGL(radix_hist_kernel, dim3(sel_grid), dim3(256), 0, s, sc, n, shift,
                               ws->st);
            hipLaunchKernelGGL(radix_pick_kernel, dim3(1), dim3(256), 0, s, shift, ws->st);
        }
        hipLaunchKernelGGL(radix_compact_kernel, dim3(sel_grid), dim3(256), 0, s, sc, n, ws->st,
                           ws->selkeys.p);
        if (P > k_eff)
            hipLaunchKernelGGL(zero_tail_kernel, dim3((unsigned)((P - k_eff + 255) / 256)), dim3(256),
                               0, s, ws->selkeys.p, k_eff, P);
        if (P <= 4096) {
            hipLaunchKernelGGL(bitonic_sort_lds_kernel, dim3(1), dim3(1024), 0, s, ws->selkeys.p,
                               (int)P);
        } else {
            for (int64_t size = 2; size <= P; size <<= 1)
                for (int64_t stride = size >> 1; stride > 0; stride >>= 1)
                    hipLaunchKernelGGL(bitonic_step_kernel, dim3((unsigned)((P / 2 + 255) / 256)),
                                       dim3(256), 0, s, ws->selkeys.p, P, size, stride);
        }
        hipLaunchKernelGGL(emit_sorted_kernel, dim3((k + 255) / 256), dim3(256), 0, s,
                           ws->selkeys.p, k, idx->metric, label_offset, D_dev + (int64_t)qi * k,
                           I_dev + (int64_t)qi * k, mask_dev ? 1 : 0);
        MVDB_HIP(hipGetLastError());
    }
    return 0;
}

// Stage host queries [nq,d] into ws->q (padded to ld) on ws->stream.
int stage_queries(const mvdb_index* idx, Workspace* ws, const float* q_host, int nq) {
    const size_t elems = (size_t)nq * idx->ld;
    MVDB_TRY(ws->q.reserve(elems));
    MVDB_TRY(ws->pin.reserve(elems * sizeof(float)));
    float* st = (float*)ws->pin.p;
    if (idx->ld == idx->d) {
        memcpy(st, q_host, elems * sizeof(float));
    } else {
        memset(st, 0, elems * sizeof(float));
        for (int i = 0; i < nq; ++i)
            memcpy(st + (size_t)i * idx->ld, q_host + (size_t)i * idx->d, idx->d * sizeof(float));
    }
    MVDB_HIP(hipMemcpyAsync(ws->q.p, st, elems * sizeof(float), hipMemcpyHostToDevice, ws->stream));
    return 0;
}

// Host API epilogue: results sit packed in ws->out ([I | D]); one async D2H into pinned staging, sync, unpack.
int fetch_results(Workspace* ws, size_t total, float* D_host, int64_t* I_host) {
    const size_t bytes = total * (sizeof(int64_t) + sizeof(float));
    MVDB_TRY(ws->pin_out.reserve(bytes));
    hipError_t e = hipMemcpyAsync(ws->pin_out.p, ws->out.p, bytes, hipMemcpyDeviceToHost, ws->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ws->stream);
    if (e != hipSuccess) return fail(MVDB_ERR_HIP, "search failed: %s", hipGetErrorString(e));
    memcpy(I_host, ws->pin_out.p, total * sizeof(int64_t));
    memcpy(D_host, (const char*)ws->pin_out.p + total * sizeof(int64_t), total * sizeof(float));
    return 0;
}

// A search being captured into a hipGraph: from now on this workspace keeps every buffer it outgrows (RetireScope), so a
// later, larger eager call on the same stream cannot free memory the graph still names.
void note_capture(Workspace* ws) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (ws->stream && !ws->captured && hipStreamIsCapturing(ws->stream, &st) == hipSuccess &&
        st == hipStreamCaptureStatusActive)
        ws->captured = true;
}

int check_search_args(const mvdb_index* idx, const void* q, int nq, int k, const void* D,
                      const void* I) {
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    if (!q || !D || !I) return fail(MVDB_ERR_ARG, "NULL buffer passed to search");
    if (nq <= 0) return fail(MVDB_ERR_ARG, "nq must be positive (got %d)", nq);
    if (k <= 0) return fail(MVDB_ERR_ARG, "k must be positive (got %d)", k);
    return 0;
}

// Rows of readable slack allocated behind the matrix: the tiled batch kernels fetch whole 32-row tiles (rows past n - 1
// are read, scored and never nominated) so that their DMA issue needs no per-lane bounds handling.
constexpr int64_t kRowSlack = 32;

void drop_shadow(const mvdb_index* idx) {
    if (idx->Xh) (void)hipFree(idx->Xh);
    if (idx->Hn) (void)hipFree(idx->Hn);
    idx->Hn = nullptr;
    idx->hn_rows = 0;
    idx->Xh = nullptr;
    idx->xh_cap = idx->xh_rows = 0;
    idx->xh_scale = 0.f;
    idx->xh_failed = false;
}

// The rows changed their numbers or their scale but the allocation still fits: the shadow is emptied, not freed — hipFree waits
// for the whole device and the next batch search would allocate the same bytes again — and rebuilt in place on demand.
void invalidate_shadow(const mvdb_index* idx) {
    idx->xh_rows = 0;
    idx->hn_rows = 0;
    idx->xh_failed = false;
}

// The shadow for a batch search on stream s (caller holds the index shared): the existing one if it covers the rows at
// this scale, else built here — once: 10M x 512 convert in ~5 ms — and published after a wait for s, so that searches on
// other streams find complete data.  NULL (the fp32 path serves): no kernel for d, switched off, allocation failed, or the
// stream is being captured (a build allocates and synchronises).
const _Float16* ensure_shadow(const mvdb_index* idx, hipStream_t s, float xscale) {
    if (!half_shadow_dim(idx->d) || idx->ld != idx->d || idx->kn.disable_half_shadow || !(xscale > 0.f)) return nullptr;
    std::lock_guard<std::mutex> lk(idx->shadow_mu);
    if (idx->Xh && idx->xh_scale == xscale && idx->xh_rows == idx->n) return idx->Xh;
    if (idx->xh_failed) return nullptr;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive) return nullptr;
    // (a stale shadow can only be one whose rows are a prefix at another scale or count: mutators drop or extend it while no
    //  search is in flight, so nobody else is reading what is rebuilt here)
    if (idx->Xh && idx->xh_cap < idx->n) drop_shadow(idx);
    if (idx->Xh && idx->xh_scale != xscale) {   // another scale: every row is converted again, into the same allocation
        invalidate_shadow(idx);
        idx->xh_scale = xscale;
    }
    if (!idx->Xh) {
        _Float16* p = nullptr;
        const int64_t cap = std::max<int64_t>(idx->cap, idx->n);
        if (hipMalloc((void**)&p, (size_t)(cap + kRowSlack) * idx->d * sizeof(_Float16)) != hipSuccess) {
            (void)hipGetLastError();
            idx->xh_failed = true;
            return nullptr;
        }
        idx->Xh = p;
        idx->xh_cap = cap;
        idx->xh_rows = 0;
        idx->xh_scale = xscale;
        // (with the shadow — never inside a capture, once per index —: the refusal count and its host-mapped mirror, tile_flags_wanted)
        if (!idx->rf_dev && hipMalloc((void**)&idx->rf_dev, sizeof(unsigned int)) == hipSuccess) {
            if (hipMemset(idx->rf_dev, 0, sizeof(unsigned int)) != hipSuccess || idx->rf_host.reserve(64) != 0) {
                (void)hipFree(idx->rf_dev);
                idx->rf_dev = nullptr;
            } else {
                *reinterpret_cast<volatile unsigned int*>(idx->rf_host.p) = 0u;
            }
        }
        (void)hipGetLastError();
    }
    if (launch_half_shadow(idx->X + idx->xh_rows * idx->ld, idx->ld, idx->d, idx->n - idx->xh_rows, xscale,
                           idx->Xh + idx->xh_rows * idx->d, idx->device, s) != 0 ||
        hipStreamSynchronize(s) != hipSuccess) {
        drop_shadow(idx);
        idx->xh_failed = true;
        return nullptr;
    }
    idx->xh_rows = idx->n;
    return idx->Xh;
}

// L2 over rows of mixed norms (launch_half_pass: l2off): |x_r|^2 / 2 of the shadow's rows, 4 bytes per row beside it (+ zeroed
// slack: the kernel fetches a whole tile's offsets), built on first use and caught up after appends here; dropped with the
// shadow.  NULL: no complete shadow, allocation failed, or the stream is being captured before the array exists (the pass
// then nominates by inner product and the norm-range certificate sends what it cannot certify to the exact kernels).
const float* ensure_offsets(const mvdb_index* idx, hipStream_t s) {
    std::lock_guard<std::mutex> lk(idx->shadow_mu);
    if (!idx->Xh || idx->xh_rows != idx->n) return nullptr;
    if (idx->Hn && idx->hn_rows == idx->n) return idx->Hn;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive) return nullptr;
    if (!idx->Hn) {
        float* h = nullptr;
        const size_t bytes = (size_t)(idx->xh_cap + kRowSlack) * sizeof(float);
        if (hipMalloc((void**)&h, bytes) != hipSuccess || hipMemsetAsync(h, 0, bytes, s) != hipSuccess) {
            (void)hipGetLastError();
            if (h) (void)hipFree(h);
            return nullptr;
        }
        idx->Hn = h;
        idx->hn_rows = 0;
    }
    if (launch_half_norms(idx->X + idx->hn_rows * idx->ld, idx->ld, idx->d, idx->n - idx->hn_rows, idx->Hn + idx->hn_rows, idx->device,
                          s) != 0 ||
        hipStreamSynchronize(s) != hipSuccess) {
        (void)hipFree(idx->Hn);
        idx->Hn = nullptr;
        idx->hn_rows = 0;
        return nullptr;
    }
    idx->hn_rows = idx->n;
    return idx->Hn;
}

int grow(mvdb_index* idx, int64_t need) {
    if (need <= idx->cap) return 0;
    int64_t cap = std::max<int64_t>(need, idx->cap + idx->cap / 2);
    cap = std::max<int64_t>(cap, 1024);
    // the fp16 shadow (50 % of the matrix's bytes) goes first: it is sized for the old capacity anyway (rebuilt by the next
    // batch search), and an add on a nearly full device must not fail for a copy that is about to be dropped
    drop_shadow(idx);
    float* nx = nullptr;
    MVDB_HIP(hipMalloc((void**)&nx, (size_t)(cap + kRowSlack) * idx->ld * sizeof(float)));
    if (idx->n > 0) {
        hipError_t e = hipMemcpyAsync(nx, idx->X, (size_t)idx->n * idx->ld * sizeof(float),
                                      hipMemcpyDeviceToDevice, idx->mut);
        if (e == hipSuccess) e = hipStreamSynchronize(idx->mut);
        if (e != hipSuccess) {
            (void)hipFree(nx);
            return fail(MVDB_ERR_HIP, "device copy while growing index failed: %s",
                        hipGetErrorString(e));
        }
    }
    if (idx->X) (void)hipFree(idx->X);
    idx->X = nx;
    idx->cap = cap;
    return 0;
}

int normalize_range(const mvdb_index* idx, float* base, int64_t n, hipStream_t s) {
    if (n <= 0) return 0;
    const Shape sh = choose_shape(idx->d4);
    const int rpi = 64 / sh.G;
    const int64_t waves = (n + rpi - 1) / rpi;
    const int cus = device_cus(idx->device);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((waves + 3) / 4, (int64_t)cus * 8));
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(grid), dim3(256), 0, s, base, n, idx->ld, idx->d4,
                       sh.G);
    MVDB_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* mvdb_last_error(void) { return g_err; }
int mvdb_abi_version(void) { return 1; }

int mvdb_device_count(int* count) {
    if (!count) return fail(MVDB_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        *count = 0;
        return fail(MVDB_ERR_NODEVICE, "no HIP device available (%s)",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    *count = n;
    return 0;
}

int mvdb_index_create(int d, int metric, int device, mvdb_index** out) {
    if (!out) return fail(MVDB_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (d <= 0) return fail(MVDB_ERR_ARG, "dimension must be positive (got %d)", d);
    if (d > kMaxC * 64 * 4) return fail(MVDB_ERR_ARG, "dimension %d > 4096 is not supported", d);
    if (metric != MVDB_METRIC_IP && metric != MVDB_METRIC_L2)
        return fail(MVDB_ERR_ARG, "unknown metric %d", metric);
    MVDB_TRY(ensure_device(device));
    mvdb_index* idx = new mvdb_index();
    idx->d = d;
    idx->ld = ((int64_t)d + 3) / 4 * 4;
    idx->d4 = (int)(idx->ld / 4);
    idx->metric = metric;
    idx->device = device;
    idx->kn = read_knobs();
    *out = idx;
    return 0;
}

int mvdb_index_reload_env(mvdb_index* idx) {
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    idx->kn = read_knobs();
    return 0;
}

int mvdb_index_set_option(mvdb_index* idx, const char* name, long long value) {
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    if (!name) return fail(MVDB_ERR_ARG, "option name is NULL");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    const std::string n(name);
    if (n == "shadow_single_query") {
        idx->kn.shadow_single_query = value != 0;
        idx->sq_suspend_left.store(0);   // a fresh start: the route probes again from its next call
        idx->sq_window_calls.store(idx->sq_calls.load());
        if (idx->sq_fail_host.p) idx->sq_window_fail.store(*reinterpret_cast<volatile unsigned int*>(idx->sq_fail_host.p));
    } else if (n == "half_shadow")
        idx->kn.disable_half_shadow = value == 0;
    else if (n == "compact_bytes") {
        if (value <= 0) return fail(MVDB_ERR_ARG, "compact_bytes must be positive");
        idx->kn.compact_bytes = value;
    } else
        return fail(MVDB_ERR_ARG, "unknown index option '%s' (shadow_single_query, half_shadow, compact_bytes)", name);
    return 0;
}

long long mvdb_index_single_route_suspensions(const mvdb_index* idx) {
    return idx ? (long long)idx->sq_suspensions.load() : -1;
}

int mvdb_index_free(mvdb_index* idx) {
    if (!idx) return 0;
    {
        std::unique_lock<std::shared_mutex> lk(idx->mu);
        DeviceGuard dg(idx->device);
        (void)quiesce(idx);
        if (idx->mut) (void)hipStreamDestroy(idx->mut);
        if (idx->normmax) (void)hipFree(idx->normmax);
        if (idx->ctmp) (void)hipFree(idx->ctmp);
        if (idx->sq_fail_dev) (void)hipFree(idx->sq_fail_dev);
        idx->sq_fail_host.release();
        if (idx->rf_dev) (void)hipFree(idx->rf_dev);
        idx->rf_host.release();
        drop_shadow(idx);
        for (Workspace* w : idx->free_ws) {
            w->destroy();
            delete w;
        }
        for (auto& kv : idx->stream_ws) {
            kv.second->destroy();
            delete kv.second;
        }
        if (idx->default_stream_ws) {
            idx->default_stream_ws->destroy();
            delete idx->default_stream_ws;
        }
        if (idx->X) (void)hipFree(idx->X);
    }
    delete idx;
    return 0;
}

int mvdb_index_reset(mvdb_index* idx) {
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    DeviceGuard dg(idx->device);
    MVDB_TRY(quiesce(idx));
    idx->n = 0;
    drop_shadow(idx);
    idx->row_norm_bound = 0.f;
    idx->norm2_lo = INFINITY;
    idx->norm2_hi = 0.f;
    ++idx->renumbered;
    return 0;
}

int64_t mvdb_index_ntotal(const mvdb_index* idx) { return idx ? idx->n : -1; }
int64_t mvdb_index_shadow_rows(const mvdb_index* idx) {
    if (!idx) return -1;
    std::lock_guard<std::mutex> lk(idx->shadow_mu);
    return idx->Xh ? idx->xh_rows : 0;
}
int mvdb_index_dim(const mvdb_index* idx) { return idx ? idx->d : -1; }
int mvdb_index_device(const mvdb_index* idx) { return idx ? idx->device : -1; }

int mvdb_index_reserve(mvdb_index* idx, int64_t n) {
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    if (n < 0) return fail(MVDB_ERR_ARG, "negative row count");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    DeviceGuard dg(idx->device);
    if (n <= idx->cap) return 0;
    MVDB_TRY(quiesce(idx));
    // exact-size growth (no 1.5x slack): the caller knows the final size
    drop_shadow(idx);  // before the new matrix is allocated (see grow)
    float* nx = nullptr;
    MVDB_HIP(hipMalloc((void**)&nx, (size_t)(n + kRowSlack) * idx->ld * sizeof(float)));
    if (idx->n > 0) {
        hipError_t e = hipMemcpyAsync(nx, idx->X, (size_t)idx->n * idx->ld * sizeof(float), hipMemcpyDeviceToDevice, idx->mut);
        if (e == hipSuccess) e = hipStreamSynchronize(idx->mut);
        if (e != hipSuccess) {
            (void)hipFree(nx);
            return fail(MVDB_ERR_HIP, "device copy while reserving failed: %s", hipGetErrorString(e));
        }
    }
    if (idx->X) (void)hipFree(idx->X);
    idx->X = nx;
    idx->cap = n;
    drop_shadow(idx);
    return 0;
}

// Keeps idx->row_norm_bound >= |row| for every stored row: 1 for rows normalised on the device, one
// extra read of the new rows otherwise (non-finite rows leave the bound non-finite).
static int note_row_norms(mvdb_index* idx, const float* dst, int64_t n, int normalize) {
    const bool l2 = idx->metric == MVDB_METRIC_L2;
    if (normalize && !l2) {
        // |row| after normalize_rows_kernel: |x| / sqrt(fl(|x|^2)) with an fp32 sum of depth <= 4 * 16 + 6 (d <= 4096),
        // one sqrt, one divide, one multiply: <= 1 + (70 / 2 + 3) * 2^-24 = 1 + 2.3e-6
        idx->row_norm_bound = std::max(idx->row_norm_bound, 1.000004f);
        return 0;
    }
    // raw rows — and every row of an L2 index, whose certified batch passes need BOTH ends of the norm range (a zero row
    // that normalisation left alone has norm 0): one extra read of the new rows
    if (!idx->normmax) MVDB_HIP(hipMalloc((void**)&idx->normmax, 2 * sizeof(unsigned int)));
    unsigned int* dmax = idx->normmax;
    unsigned int bits[2] = {0u, 0x7F800000u};  // max = 0, min = +inf
    hipError_t e = hipMemcpyAsync(dmax, bits, sizeof(bits), hipMemcpyHostToDevice, idx->mut);
    if (e == hipSuccess) e = hipStreamSynchronize(idx->mut);  // (bits is a stack buffer)
    if (e == hipSuccess) {
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)device_cus(idx->device) * 16));
        if (l2)
            hipLaunchKernelGGL(row_norm2_range_kernel, dim3(grid), dim3(256), 0, idx->mut, dst, n, idx->ld, idx->d4, dmax);
        else
            hipLaunchKernelGGL(max_row_norm2_kernel, dim3(grid), dim3(256), 0, idx->mut, dst, n, idx->ld, idx->d4, dmax);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(bits, dmax, sizeof(bits), hipMemcpyDeviceToHost, idx->mut);
    if (e == hipSuccess) e = hipStreamSynchronize(idx->mut);
    MVDB_HIP(e);
    float n2, lo2;
    memcpy(&n2, &bits[0], sizeof(n2));
    memcpy(&lo2, &bits[1], sizeof(lo2));
    const float bound = std::isfinite(n2) ? std::sqrt(n2) * 1.000001f : INFINITY;
    idx->row_norm_bound = std::max(idx->row_norm_bound, normalize ? std::max(bound, 1.000004f) : bound);
    if (l2) {
        // the kernel's fp32 sums carry <= (depth + 1) 2^-24 < 5e-6 relative error: widen to bounds of the TRUE squared norms
        if (std::isfinite(n2) && std::isfinite(lo2)) {
            idx->norm2_hi = std::max(idx->norm2_hi, n2 * 1.00001f);
            idx->norm2_lo = std::min(idx->norm2_lo, lo2 * 0.99999f);
        } else {
            idx->norm2_hi = INFINITY;
            idx->norm2_lo = 0.f;
        }
    }
    return 0;
}

// add: the shadow follows when it is there, still fits and the scale the new norm bound asks for is the one it was built with
static int extend_shadow(mvdb_index* idx, int64_t n_new) {
    if (!idx->Xh) {
        idx->xh_failed = false;
        return 0;
    }
    const float xscale = half_xscale(idx->row_norm_bound);
    if (idx->xh_cap < idx->n + n_new) {
        drop_shadow(idx);
        return 0;
    }
    if (idx->xh_rows != idx->n || xscale != idx->xh_scale) {   // (the next batch search converts every row again, in place)
        invalidate_shadow(idx);
        return 0;
    }
    MVDB_TRY(launch_half_shadow(idx->X + idx->n * idx->ld, idx->ld, idx->d, n_new, xscale, idx->Xh + idx->n * idx->d, idx->device,
                                idx->mut));
    MVDB_HIP(hipStreamSynchronize(idx->mut));
    idx->xh_rows = idx->n + n_new;
    return 0;
}

int mvdb_index_add(mvdb_index* idx, const float* x_host, int64_t n, int normalize) {
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    if (n < 0) return fail(MVDB_ERR_ARG, "negative row count");
    if (n == 0) return 0;
    if (!x_host) return fail(MVDB_ERR_ARG, "x is NULL");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    DeviceGuard dg(idx->device);
    if (idx->n + n > 0xFFFFFFFFll)
        return fail(MVDB_ERR_ARG, "more than 2^32-1 rows per device index are not supported");
    MVDB_TRY(quiesce(idx));
    MVDB_TRY(grow(idx, idx->n + n));
    float* dst = idx->X + idx->n * idx->ld;
    // (the rows are not visible to searches until idx->n moves, below; everything runs on the index's own stream)
    if (idx->ld == idx->d) {
        MVDB_HIP(hipMemcpyAsync(dst, x_host, (size_t)n * idx->d * sizeof(float), hipMemcpyHostToDevice, idx->mut));
    } else {
        MVDB_HIP(hipMemsetAsync(dst, 0, (size_t)n * idx->ld * sizeof(float), idx->mut));
        MVDB_HIP(hipMemcpy2DAsync(dst, idx->ld * sizeof(float), x_host, idx->d * sizeof(float),
                                  idx->d * sizeof(float), (size_t)n, hipMemcpyHostToDevice, idx->mut));
    }
    if (normalize) MVDB_TRY(normalize_range(idx, dst, n, idx->mut));
    MVDB_HIP(hipStreamSynchronize(idx->mut));  // the caller's buffer is free again, the rows are in place
    MVDB_TRY(note_row_norms(idx, dst, n, normalize));
    MVDB_TRY(extend_shadow(idx, n));
    idx->n += n;
    return 0;
}

int mvdb_index_add_device(mvdb_index* idx, const float* x_dev, int64_t n, int normalize) {
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    if (n < 0) return fail(MVDB_ERR_ARG, "negative row count");
    if (n == 0) return 0;
    if (!x_dev) return fail(MVDB_ERR_ARG, "x is NULL");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    DeviceGuard dg(idx->device);
    if (idx->n + n > 0xFFFFFFFFll)
        return fail(MVDB_ERR_ARG, "more than 2^32-1 rows per device index are not supported");
    MVDB_TRY(quiesce(idx));
    MVDB_TRY(grow(idx, idx->n + n));
    float* dst = idx->X + idx->n * idx->ld;
    // x_dev is the caller's: whatever produced it must be complete (or ordered before the legacy stream) — as before, the
    // copy below waits for the legacy stream only when it is issued there, so it is issued on the mutators' stream after a
    // legacy-stream join
    MVDB_HIP(hipStreamSynchronize(nullptr));
    if (idx->ld == idx->d) {
        MVDB_HIP(hipMemcpyAsync(dst, x_dev, (size_t)n * idx->d * sizeof(float), hipMemcpyDeviceToDevice, idx->mut));
    } else {
        const int cus = device_cus(idx->device);
        const int64_t total = n * idx->ld;
        const int grid = (int)std::min<int64_t>((total + 255) / 256, (int64_t)cus * 16);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, idx->mut, dst, x_dev, n, idx->d,
                           idx->ld);
        MVDB_HIP(hipGetLastError());
    }
    if (normalize) MVDB_TRY(normalize_range(idx, dst, n, idx->mut));
    MVDB_HIP(hipStreamSynchronize(idx->mut));
    MVDB_TRY(note_row_norms(idx, dst, n, normalize));
    MVDB_TRY(extend_shadow(idx, n));
    idx->n += n;
    return 0;
}

int mvdb_index_add_synthetic(mvdb_index* idx, int64_t n, uint64_t seed, int64_t first_row,
                             int normalize) {
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    if (n < 0 || first_row < 0) return fail(MVDB_ERR_ARG, "negative row count / offset");
    if (n == 0) return 0;
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    DeviceGuard dg(idx->device);
    if (idx->n + n > 0xFFFFFFFFll)
        return fail(MVDB_ERR_ARG, "more than 2^32-1 rows per device index are not supported");
    MVDB_TRY(quiesce(idx));
    MVDB_TRY(grow(idx, idx->n + n));
    float* dst = idx->X + idx->n * idx->ld;
    const int cus = device_cus(idx->device);
    const int64_t total = n * idx->d4;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, (int64_t)cus * 16));
    hipLaunchKernelGGL(synth_fill_kernel, dim3(grid), dim3(256), 0, idx->mut, dst, n, idx->ld, idx->d,
                       seed, first_row);
    MVDB_HIP(hipGetLastError());
    if (normalize) MVDB_TRY(normalize_range(idx, dst, n, idx->mut));
    MVDB_HIP(hipStreamSynchronize(idx->mut));
    MVDB_TRY(note_row_norms(idx, dst, n, normalize));
    MVDB_TRY(extend_shadow(idx, n));
    idx->n += n;
    return 0;
}

int mvdb_index_get_rows(const mvdb_index* idx, int64_t row0, int64_t n, float* out_host) {
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    if (n == 0) return 0;
    if (!out_host) return fail(MVDB_ERR_ARG, "out is NULL");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    if (row0 < 0 || n < 0 || row0 + n > idx->n)
        return fail(MVDB_ERR_ARG, "rows [%lld,%lld) out of range [0,%lld)", (long long)row0,
                    (long long)(row0 + n), (long long)idx->n);
    DeviceGuard dg(idx->device);
    const float* src = idx->X + row0 * idx->ld;
    if (idx->ld == idx->d)
        MVDB_HIP(hipMemcpy(out_host, src, (size_t)n * idx->d * sizeof(float), hipMemcpyDeviceToHost));
    else
        MVDB_HIP(hipMemcpy2D(out_host, idx->d * sizeof(float), src, idx->ld * sizeof(float),
                             idx->d * sizeof(float), (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

int mvdb_index_remove_rows(mvdb_index* idx, const int64_t* rows_host, int64_t m) {
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    if (m < 0) return fail(MVDB_ERR_ARG, "negative row count");
    if (m == 0) return 0;
    if (!rows_host) return fail(MVDB_ERR_ARG, "rows is NULL");
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    std::vector<int64_t> del(rows_host, rows_host + m);
    std::sort(del.begin(), del.end());
    for (int64_t i = 0; i < m; ++i) {
        if (del[i] < 0 || del[i] >= idx->n)
            return fail(MVDB_ERR_ARG, "row %lld out of range [0,%lld)", (long long)del[i],
                        (long long)idx->n);
        if (i && del[i] == del[i - 1])
            return fail(MVDB_ERR_ARG, "row %lld listed twice", (long long)del[i]);
    }
    DeviceGuard dg(idx->device);
    MVDB_TRY(quiesce(idx));
    ++idx->renumbered;
    invalidate_shadow(idx);  // rows are renumbered: the next batch search rebuilds the shadow in the allocation it already has
    const int64_t n_new = idx->n - m;
    if (n_new == 0) {
        idx->n = 0;
        return 0;
    }
    // rows before the first deleted one do not move: compact only the tail [first, n).  Every kept row moves UP, so the
    // tail is compacted in place, in ascending chunks through a bounded staging buffer (512 MiB, MVDB_COMPACT_BYTES; kept by the index):
    // chunk c gathers the sources of the new rows [r0, r0 + rows) — all at or above r0 + 1 — into the buffer, then copies
    // the buffer over [r0, r0 + rows); stream order makes the next chunk's sources (all >= r0 + rows) untouched by that
    // copy.  (Until round 4 the whole tail went through a temporary of its own size: one early delete in a 164 GB index
    // needed another 164 GB.)  The source row of a new row is found by binary search in the sorted list of deleted rows.
    const int64_t first = del[0];
    const int64_t tail_new = n_new - first;
    // A handful of rows (the reference deletes ONE per call, vector_database.py:119): the tail is shifted in place in one pass —
    // every byte read once and written once — with a side copy of m rows per workgroup boundary (util_kernels.hpp:
    // shift_rows_kernel); delete of an early row of 10M x 512: 14.5 -> 8.2 ms (profiles/r06_delete_probe.jsonl).  MVDB_COMPACT_INPLACE=0: always the staging path.
    if (tail_new > 0 && idx->kn.compact_inplace && (m <= kShiftMaxRows || del[m - 1] - del[0] == m - 1)) {
        const bool run = del[m - 1] - del[0] == m - 1;
        const int64_t rowunits = idx->ld / 4;  // 16-byte units per row (ld is a multiple of 4 floats)
        const int64_t new_units = tail_new * rowunits, old_units = (idx->n - first) * rowunits;
        const int64_t slice = 256 * kShiftUnits;
        const int64_t side_units = m * rowunits;
        const int64_t cus = device_cus(idx->device);
        // eight workgroups per CU; fewer (down to two) where the side copies of that many boundaries would outgrow their bound
        int64_t groups = std::min<int64_t>(cus * 8, (new_units + slice - 1) / slice);
        const int64_t fit = (int64_t)(kShiftSideBytes / ((size_t)side_units * 16)) + 1;
        const bool fits = fit >= std::min<int64_t>(groups, cus * 2);
        groups = std::min(groups, fit);
        const int64_t range = ((new_units + groups - 1) / groups + slice - 1) / slice * slice;
        groups = (new_units + range - 1) / range;
        const size_t side_bytes = (size_t)std::max<int64_t>(groups - 1, 1) * side_units * 16;
        if (fits && side_bytes <= kShiftSideBytes) {
            if (idx->ctmp_bytes < side_bytes) {
                if (idx->ctmp) (void)hipFree(idx->ctmp);
                idx->ctmp = nullptr;
                idx->ctmp_bytes = 0;
                if (hipMalloc((void**)&idx->ctmp, side_bytes) != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(MVDB_ERR_OOM, "device allocation for row compaction failed (%zu bytes)", side_bytes);
                }
                idx->ctmp_bytes = side_bytes;
            }
            int64_t* del_dev = nullptr;
            hipError_t e = hipSuccess;
            if (!run) {
                MVDB_HIP(hipMallocAsync((void**)&del_dev, (size_t)m * sizeof(int64_t), idx->mut));
                for (auto& v : del) v -= first;  // positions relative to the tail
                e = hipMemcpyAsync(del_dev, del.data(), (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice, idx->mut);
            }
            f32x4u* tail = reinterpret_cast<f32x4u*>(idx->X + first * idx->ld);
            f32x4u* side = reinterpret_cast<f32x4u*>(idx->ctmp);
            if (groups > 1) {
                const unsigned gx = (unsigned)std::min<int64_t>((side_units + 255) / 256, 64);
                hipLaunchKernelGGL(shift_save_kernel, dim3(gx, (unsigned)(groups - 1)), dim3(256), 0, idx->mut, side, tail, range, side_units,
                                   old_units);
            }
            if (run)
                hipLaunchKernelGGL(shift_rows_kernel<true>, dim3((unsigned)groups), dim3(256), 0, idx->mut, tail, side, del_dev, m, new_units,
                                   range, rowunits, side_units);
            else
                hipLaunchKernelGGL(shift_rows_kernel<false>, dim3((unsigned)groups), dim3(256), 0, idx->mut, tail, side, del_dev, m, new_units,
                                   range, rowunits, side_units);
            if (del_dev) (void)hipFreeAsync(del_dev, idx->mut);
            if (e == hipSuccess) e = hipGetLastError();
            const hipError_t es = hipStreamSynchronize(idx->mut);
            if (e == hipSuccess) e = es;
            if (e != hipSuccess) return fail(MVDB_ERR_HIP, "row compaction failed: %s", hipGetErrorString(e));
            idx->n = n_new;
            return 0;
        }
    }
    if (tail_new > 0) {
        const size_t row_bytes = (size_t)idx->ld * sizeof(float);
        const size_t want = std::min<size_t>((size_t)tail_new * row_bytes, std::max<size_t>((size_t)idx->kn.compact_bytes, row_bytes));
        if (idx->ctmp_bytes < want) {
            if (idx->ctmp) (void)hipFree(idx->ctmp);
            idx->ctmp = nullptr;
            idx->ctmp_bytes = 0;
            if (hipMalloc((void**)&idx->ctmp, want) != hipSuccess) {
                (void)hipGetLastError();
                return fail(MVDB_ERR_OOM, "device allocation for row compaction failed (%zu bytes)", want);
            }
            idx->ctmp_bytes = want;
        }
        const int64_t chunk_rows = (int64_t)(idx->ctmp_bytes / row_bytes);
        int64_t* del_dev = nullptr;
        MVDB_HIP(hipMallocAsync((void**)&del_dev, (size_t)m * sizeof(int64_t), idx->mut));  // stream-ordered: hipFree would synchronise the device
        for (auto& v : del) v -= first;  // positions relative to the tail
        hipError_t e = hipMemcpyAsync(del_dev, del.data(), (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice, idx->mut);
        float* tail = idx->X + first * idx->ld;
        for (int64_t r0 = 0; r0 < tail_new && e == hipSuccess; r0 += chunk_rows) {
            const int64_t rows = std::min(chunk_rows, tail_new - r0);
            const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((rows + 3) / 4, (int64_t)device_cus(idx->device) * 16));
            hipLaunchKernelGGL(gather_kept_rows_kernel, dim3(grid), dim3(256), 0, idx->mut, idx->ctmp, tail, del_dev, m, r0, rows,
                               idx->ld);
            e = hipMemcpyAsync(tail + r0 * idx->ld, idx->ctmp, (size_t)rows * row_bytes, hipMemcpyDeviceToDevice, idx->mut);
        }
        (void)hipFreeAsync(del_dev, idx->mut);
        if (e == hipSuccess) e = hipGetLastError();
        const hipError_t es = hipStreamSynchronize(idx->mut);
        if (e == hipSuccess) e = es;
        if (e != hipSuccess) return fail(MVDB_ERR_HIP, "row compaction failed: %s", hipGetErrorString(e));
    }
    idx->n = n_new;
    return 0;
}

int mvdb_index_search(const mvdb_index* idx, const float* q_host, int nq, int k, int normalize_q,
                      float* D_host, int64_t* I_host) {
    MVDB_TRY(check_search_args(idx, q_host, nq, k, D_host, I_host));
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    DeviceGuard dg(idx->device);
    Workspace* ws = idx->acquire();
    if (!ws) return MVDB_ERR_HIP;
    int rc = 0;
    do {
        if ((rc = stage_queries(idx, ws, q_host, nq))) break;
        const size_t total = (size_t)nq * k;
        if ((rc = ws->out.reserve(total + (total + 1) / 2))) break;  // total int64 + total fp32
        float* D_dev = reinterpret_cast<float*>(ws->out.p + total);
        if ((rc = search_core(idx, ws, ws->q.p, nq, k, normalize_q, nullptr, 0, 0, D_dev, ws->out.p))) break;
        rc = fetch_results(ws, total, D_host, I_host);
    } while (0);
    idx->release(ws);
    return rc;
}

int mvdb_index_search_subset(const mvdb_index* idx, const float* q_host, int nq, int k,
                             int normalize_q, const int64_t* rows_host, int64_t m, float* D_host,
                             int64_t* I_host) {
    MVDB_TRY(check_search_args(idx, q_host, nq, k, D_host, I_host));
    if (m < 0) return fail(MVDB_ERR_ARG, "negative subset size");
    if (m > 0 && !rows_host) return fail(MVDB_ERR_ARG, "rows is NULL");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    {   // range check as a branch-free min/max reduction (vectorises; the list can hold millions of rows)
        int64_t lo = 0, hi = -1;
        if (m > 0) {
            lo = hi = rows_host[0];
            for (int64_t i = 1; i < m; ++i) {
                lo = rows_host[i] < lo ? rows_host[i] : lo;
                hi = rows_host[i] > hi ? rows_host[i] : hi;
            }
        }
        if (m > 0 && (lo < 0 || hi >= idx->n))
            return fail(MVDB_ERR_ARG, "subset row %lld out of range [0,%lld)", (long long)(lo < 0 ? lo : hi),
                        (long long)idx->n);
    }
    DeviceGuard dg(idx->device);
    Workspace* ws = idx->acquire();
    if (!ws) return MVDB_ERR_HIP;
    int rc = 0;
    do {
        if ((rc = stage_queries(idx, ws, q_host, nq))) break;
        const size_t total = (size_t)nq * k;
        if ((rc = ws->out.reserve(total + (total + 1) / 2))) break;
        float* D_dev = reinterpret_cast<float*>(ws->out.p + total);
        if ((rc = ws->rows.reserve((size_t)std::max<int64_t>(m, 1)))) break;
        if (m > 0) {
            hipError_t e = hipMemcpyAsync(ws->rows.p, rows_host, (size_t)m * sizeof(int64_t),
                                          hipMemcpyHostToDevice, ws->stream);
            if (e != hipSuccess) {
                rc = fail(MVDB_ERR_HIP, "subset upload failed: %s", hipGetErrorString(e));
                break;
            }
        }
        if ((rc = search_core(idx, ws, ws->q.p, nq, k, normalize_q, ws->rows.p, m, 0, D_dev, ws->out.p))) break;
        rc = fetch_results(ws, total, D_host, I_host);
    } while (0);
    idx->release(ws);
    return rc;
}

int mvdb_index_search_device(const mvdb_index* idx, const float* q_dev, int nq, int k,
                             int normalize_q, int64_t label_offset, float* D_dev, int64_t* I_dev,
                             void* stream) {
    MVDB_TRY(check_search_args(idx, q_dev, nq, k, D_dev, I_dev));
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    DeviceGuard dg(idx->device);
    Workspace* ws = idx->for_stream((hipStream_t)stream);
    if (!ws) return fail(MVDB_ERR_HIP, "workspace allocation failed");
    std::lock_guard<std::mutex> use(ws->use_mu);
    note_capture(ws);
    RetireScope keep(ws->captured ? &ws->retired : nullptr);
    const float* q = q_dev;
    if (idx->ld != idx->d) {  // pad the dense queries to the row stride
        MVDB_TRY(ws->q.reserve((size_t)nq * idx->ld));
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(((int64_t)nq * idx->ld + 255) / 256)),
                           dim3(256), 0, ws->stream, ws->q.p, q_dev, (int64_t)nq, idx->d, idx->ld);
        MVDB_HIP(hipGetLastError());
        q = ws->q.p;
    }
    // Nothing below synchronises the stream or reads from the device (certification failures of the batch passes are
    // re-run by device-gated launches): the call may be captured into a hipGraph once an eager call of the same shape
    // has sized the stream's workspace (allocation is not capturable).
    return search_core(idx, ws, q, nq, k, normalize_q, nullptr, 0, label_offset, D_dev, I_dev);
}

__global__ void map_subset_labels_kernel(int64_t* I, int64_t total, const int64_t* __restrict__ rows,
                                         int64_t label_offset) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const int64_t p = I[i];
        I[i] = p >= 0 ? rows[p] + label_offset : -1;
    }
}

int mvdb_index_search_subset_device(const mvdb_index* idx, const float* q_dev, int nq, int k, int normalize_q,
                                    const int64_t* rows_dev, int64_t m, int map_labels, int64_t label_offset,
                                    float* D_dev, int64_t* I_dev, void* stream) {
    MVDB_TRY(check_search_args(idx, q_dev, nq, k, D_dev, I_dev));
    if (m < 0) return fail(MVDB_ERR_ARG, "negative subset size");
    if (m > 0 && !rows_dev) return fail(MVDB_ERR_ARG, "rows is NULL");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    DeviceGuard dg(idx->device);
    Workspace* ws = idx->for_stream((hipStream_t)stream);
    if (!ws) return fail(MVDB_ERR_HIP, "workspace allocation failed");
    std::lock_guard<std::mutex> use(ws->use_mu);
    note_capture(ws);
    RetireScope keep(ws->captured ? &ws->retired : nullptr);
    const float* q = q_dev;
    if (idx->ld != idx->d) {
        MVDB_TRY(ws->q.reserve((size_t)nq * idx->ld));
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(((int64_t)nq * idx->ld + 255) / 256)),
                           dim3(256), 0, ws->stream, ws->q.p, q_dev, (int64_t)nq, idx->d, idx->ld);
        MVDB_HIP(hipGetLastError());
        q = ws->q.p;
    }
    // m == 0: search_core's empty-corpus branch needs a non-NULL row list to take the subset meaning
    const int64_t* rows = m > 0 ? rows_dev : reinterpret_cast<const int64_t*>(ws->st);
    MVDB_TRY(search_core(idx, ws, q, nq, k, normalize_q, rows, m, map_labels ? 0 : label_offset, D_dev, I_dev));
    if (map_labels && m > 0) {
        const int64_t total = (int64_t)nq * k;
        hipLaunchKernelGGL(map_subset_labels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ws->stream,
                           I_dev, total, rows_dev, label_offset);
        MVDB_HIP(hipGetLastError());
    }
    return 0;
}

static int finish_mask_labels(const mvdb_index* idx, Workspace* ws, int nq, int k, const uint64_t* mask_dev, int labels,
                              int64_t* I_dev) {
    if (labels == 0) {  // positions in the ascending list of the set rows (what mvdb_index_search_subset returns for that list)
        hipLaunchKernelGGL(mask_rank_kernel, dim3((unsigned)((int64_t)nq * k)), dim3(256), 0, ws->stream, I_dev, mask_dev);
        MVDB_HIP(hipGetLastError());
    }
    return 0;
}

int mvdb_index_search_masked_device(const mvdb_index* idx, const float* q_dev, int nq, int k, int normalize_q,
                                    const uint64_t* mask_dev, int labels, int64_t label_offset, float* D_dev,
                                    int64_t* I_dev, void* stream) {
    MVDB_TRY(check_search_args(idx, q_dev, nq, k, D_dev, I_dev));
    if (!mask_dev) return fail(MVDB_ERR_ARG, "mask is NULL");
    if (labels != 0 && labels != 1) return fail(MVDB_ERR_ARG, "labels must be 0 (positions) or 1 (row numbers)");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    DeviceGuard dg(idx->device);
    Workspace* ws = idx->for_stream((hipStream_t)stream);
    if (!ws) return fail(MVDB_ERR_HIP, "workspace allocation failed");
    std::lock_guard<std::mutex> use(ws->use_mu);
    note_capture(ws);
    RetireScope keep(ws->captured ? &ws->retired : nullptr);
    const float* q = q_dev;
    if (idx->ld != idx->d) {
        MVDB_TRY(ws->q.reserve((size_t)nq * idx->ld));
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(((int64_t)nq * idx->ld + 255) / 256)),
                           dim3(256), 0, ws->stream, ws->q.p, q_dev, (int64_t)nq, idx->d, idx->ld);
        MVDB_HIP(hipGetLastError());
        q = ws->q.p;
    }
    MVDB_TRY(search_core(idx, ws, q, nq, k, normalize_q, nullptr, 0, labels == 1 ? label_offset : 0, D_dev, I_dev, true, mask_dev));
    return finish_mask_labels(idx, ws, nq, k, mask_dev, labels, I_dev);
}

int mvdb_index_search_masked(const mvdb_index* idx, const float* q_host, int nq, int k, int normalize_q,
                             const uint64_t* mask_host, int labels, float* D_host, int64_t* I_host) {
    MVDB_TRY(check_search_args(idx, q_host, nq, k, D_host, I_host));
    if (!mask_host) return fail(MVDB_ERR_ARG, "mask is NULL");
    if (labels != 0 && labels != 1) return fail(MVDB_ERR_ARG, "labels must be 0 (positions) or 1 (row numbers)");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    DeviceGuard dg(idx->device);
    Workspace* ws = idx->acquire();
    if (!ws) return MVDB_ERR_HIP;
    int rc = 0;
    do {
        if ((rc = stage_queries(idx, ws, q_host, nq))) break;
        const size_t total = (size_t)nq * k;
        if ((rc = ws->out.reserve(total + (total + 1) / 2))) break;
        float* D_dev = reinterpret_cast<float*>(ws->out.p + total);
        const size_t words = (size_t)((idx->n + 63) / 64);
        if ((rc = ws->rows.reserve(std::max<size_t>(words, 1)))) break;  // 8-byte words: the row-list buffer serves
        uint64_t* mask_dev = reinterpret_cast<uint64_t*>(ws->rows.p);
        if (words) {
            hipError_t e = hipMemcpyAsync(mask_dev, mask_host, words * sizeof(uint64_t), hipMemcpyHostToDevice, ws->stream);
            if (e != hipSuccess) {
                rc = fail(MVDB_ERR_HIP, "mask upload failed: %s", hipGetErrorString(e));
                break;
            }
        }
        if ((rc = search_core(idx, ws, ws->q.p, nq, k, normalize_q, nullptr, 0, 0, D_dev, ws->out.p, true, mask_dev))) break;
        if ((rc = finish_mask_labels(idx, ws, nq, k, mask_dev, labels, ws->out.p))) break;
        rc = fetch_results(ws, total, D_host, I_host);
    } while (0);
    idx->release(ws);
    return rc;
}

// ---- resident row sets: a filter's rows uploaded ONCE, searched many times ---------------------------------------------
// Built on the host in one pass over the list (range check, sortedness, duplicates are the caller's business as in
// mvdb_index_search_subset): a SORTED list that keeps at least 90 % of the rows, and every "all rows but these" set,
// becomes a bitmap (n / 8 bytes up the wire instead of 8 per row, one full-rate pass per search); anything else stays a
// row list on the device.  Results carry ROW NUMBERS.
}  // extern "C"

// would search_core answer a bitmap-selected batch of nq queries on shared corpus passes? (its own conditions, restated)
static bool masked_batch_possible(const mvdb_index* idx, int nq, int k, int64_t n) {
    if (nq < 2 || k > kMaxFusedK || idx->ld != idx->d || idx->kn.disable_masked_batch) return false;
    if (idx->metric == MVDB_METRIC_IP) return mfma_gated_queries(idx) > 0;
    return (l2_cert_ok(idx) || l2_offsets_ok(idx, nq, n)) && half_path_ok(idx) && nq >= half_min_nq(idx, n);
}

struct mvdb_rowset {
    int device = 0;
    int64_t n_at_create = 0;  // the index's row count the set was built against
    uint64_t renumbered = 0;  // ... and its renumbering generation
    int64_t count = 0;        // rows selected
    int64_t* rows = nullptr;  // list form (device), in the caller's order
    uint64_t* mask = nullptr; // bitmap form (device)
    uint64_t* twin = nullptr; // list form, sorted, n >= 4096: the same rows as a bitmap too (n / 8 bytes) — what a BATCH of queries
                              // is searched under when sharing corpus passes beats one gathered scan per query (rowset_search_core)
};

extern "C" {

int mvdb_rowset_create(const mvdb_index* idx, const int64_t* rows_host, int64_t m, int excluded, mvdb_rowset** out) {
    if (!out) return fail(MVDB_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!idx) return fail(MVDB_ERR_ARG, "index is NULL");
    if (m < 0) return fail(MVDB_ERR_ARG, "negative row count");
    if (m > 0 && !rows_host) return fail(MVDB_ERR_ARG, "rows is NULL");
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    const int64_t n = idx->n;
    bool sorted = true;
    {
        int64_t lo = 0, hi = -1, prev = -1;
        for (int64_t i = 0; i < m; ++i) {
            const int64_t r = rows_host[i];
            lo = (i == 0 || r < lo) ? r : lo;
            hi = (i == 0 || r > hi) ? r : hi;
            sorted &= r > prev;
            prev = r;
        }
        if (m > 0 && (lo < 0 || hi >= n))
            return fail(MVDB_ERR_ARG, "row %lld out of range [0,%lld)", (long long)(lo < 0 ? lo : hi), (long long)n);
    }
    DeviceGuard dg(idx->device);
    mvdb_rowset* rs = new mvdb_rowset();
    rs->device = idx->device;
    rs->n_at_create = n;
    rs->renumbered = idx->renumbered;
    // bitmap: "all rows but these", and sorted lists that keep >= 90 % of the rows.  On the device a resident LIST is
    // gathered at 6.2-7.0 TB/s of rows touched (10M x 512: half the rows 1.50 ms, 90 % 2.65 ms, 99 % 2.89 ms) against
    // 2.87 ms for the bitmap's full pass at any density, so the bitmap only wins where the list is ~n long — there it
    // is 64x smaller in HBM and on the wire
    const bool as_mask = excluded || (sorted && m * 10 >= n * 9 && n >= 4096);
    hipError_t e = hipSuccess;
    if (as_mask) {
        const size_t words = (size_t)((n + 63) / 64);
        std::vector<uint64_t> w(std::max<size_t>(words, 1), excluded ? ~0ull : 0ull);
        if (excluded) {
            if (n == 0) w[0] = 0;                                   // an empty index: no row is selected
            else if (n & 63) w[words - 1] = (1ull << (n & 63)) - 1ull;  // bits at or beyond n stay clear
            int64_t removed = 0;
            for (int64_t i = 0; i < m; ++i) {
                uint64_t& word = w[rows_host[i] >> 6];
                const uint64_t bit = 1ull << (rows_host[i] & 63);
                removed += (word & bit) != 0;  // a row listed twice is removed once
                word &= ~bit;
            }
            rs->count = n - removed;
        } else {
            for (int64_t i = 0; i < m; ++i) w[rows_host[i] >> 6] |= 1ull << (rows_host[i] & 63);
            rs->count = m;  // sorted strictly ascending: no duplicates
        }
        e = hipMalloc((void**)&rs->mask, w.size() * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMemcpy(rs->mask, w.data(), w.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
    } else {
        rs->count = m;
        e = hipMalloc((void**)&rs->rows, (size_t)std::max<int64_t>(m, 1) * sizeof(int64_t));
        if (e == hipSuccess && m > 0) e = hipMemcpy(rs->rows, rows_host, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice);
        if (e == hipSuccess && sorted && n >= 4096 && m * 64 >= n) {   // (sparser than 1 row in 64: no batch is better off under a bitmap)
            const size_t words = (size_t)((n + 63) / 64);
            std::vector<uint64_t> w(words, 0ull);
            for (int64_t i = 0; i < m; ++i) w[rows_host[i] >> 6] |= 1ull << (rows_host[i] & 63);
            if (hipMalloc((void**)&rs->twin, words * sizeof(uint64_t)) != hipSuccess ||
                hipMemcpy(rs->twin, w.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipGetLastError();   // the twin is an optimisation: without it batches answer one query at a time
                if (rs->twin) (void)hipFree(rs->twin);
                rs->twin = nullptr;
            }
        }
    }
    if (e != hipSuccess) {
        if (rs->mask) (void)hipFree(rs->mask);
        if (rs->rows) (void)hipFree(rs->rows);
        if (rs->twin) (void)hipFree(rs->twin);
        delete rs;
        return fail(MVDB_ERR_HIP, "row set upload failed: %s", hipGetErrorString(e));
    }
    *out = rs;
    return 0;
}

int64_t mvdb_rowset_size(const mvdb_rowset* rs) { return rs ? rs->count : -1; }
int mvdb_rowset_is_bitmap(const mvdb_rowset* rs) { return rs && rs->mask ? 1 : 0; }

int mvdb_rowset_free(mvdb_rowset* rs) {
    if (!rs) return 0;
    {
        DeviceGuard dg(rs->device);
        if (rs->mask) (void)hipFree(rs->mask);
        if (rs->rows) (void)hipFree(rs->rows);
        if (rs->twin) (void)hipFree(rs->twin);
    }
    delete rs;
    return 0;
}

// the body shared by the host and the device entry points: queries on the device, results to device buffers
static int rowset_check(const mvdb_index* idx, const mvdb_rowset* rs) {
    if (!rs) return fail(MVDB_ERR_ARG, "row set is NULL");
    // rows appended since the set was built are simply not part of it (the filter was evaluated before they arrived);
    // a SHRUNK index has renumbered its rows: the set is stale
    if (rs->device != idx->device || rs->n_at_create > idx->n || rs->renumbered != idx->renumbered)
        return fail(MVDB_ERR_ARG, "the row set was built for another state of the index (%lld rows then, %lld now)",
                    (long long)rs->n_at_create, (long long)idx->n);
    return 0;
}
static int rowset_search_core(const mvdb_index* idx, Workspace* ws, const float* q, int nq, int k, int normalize_q,
                              const mvdb_rowset* rs, int64_t label_offset, float* D_dev, int64_t* I_dev) {
    tls_single_suspended = false;  // (the routing questions asked here are about batches; search_core decides for its own call)
    const int64_t total = (int64_t)nq * k;
    if (rs->count == 0 || (rs->mask && rs->n_at_create == 0)) {  // nothing selected (search_core reads m == 0 as "every row")
        hipLaunchKernelGGL(fill_missing_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ws->stream, D_dev, I_dev,
                           total, idx->metric);
        MVDB_HIP(hipGetLastError());
        return 0;
    }
    if (rs->mask)
        return search_core(idx, ws, q, nq, k, normalize_q, nullptr, rs->n_at_create, label_offset, D_dev, I_dev, true, rs->mask);
    // A batch under a LIST: one gathered scan per query costs nq x (fraction of the rows) full passes; under the list's bitmap
    // twin the queries share corpus passes (the certified pass over the fp16 shadow: ~0.6 of a full pass per 128 / 256
    // queries; else the fp32-MFMA pass: one per 16 / 32 queries).  Same labels, same tie order (the list is sorted).
    if (rs->twin && masked_batch_possible(idx, nq, k, rs->n_at_create)) {
        const int64_t n = rs->n_at_create;
        const bool half = half_path_ok(idx) && nq >= half_min_nq(idx, n);
        const int per = half ? std::max(1, half_max_queries(idx->d)) : std::max(1, mfma_gated_queries(idx));
        const bool have_shadow = __atomic_load_n(&idx->Xh, __ATOMIC_RELAXED) != nullptr;  // a heuristic read: ensure_shadow writes it under shadow_mu
        const double shared = (double)((nq + per - 1) / per) * (half && have_shadow ? 0.6 : 1.0);
        const double gathered = (double)nq * (double)rs->count / (double)n * 1.1;
        if (shared < gathered)
            return search_core(idx, ws, q, nq, k, normalize_q, nullptr, n, label_offset, D_dev, I_dev, true, rs->twin);
    }
    MVDB_TRY(search_core(idx, ws, q, nq, k, normalize_q, rs->rows, rs->count, 0, D_dev, I_dev));
    hipLaunchKernelGGL(map_subset_labels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ws->stream, I_dev, total,
                       (const int64_t*)rs->rows, label_offset);
    MVDB_HIP(hipGetLastError());
    return 0;
}

int mvdb_index_search_rowset(const mvdb_index* idx, const float* q_host, int nq, int k, int normalize_q,
                             const mvdb_rowset* rs, float* D_host, int64_t* I_host) {
    MVDB_TRY(check_search_args(idx, q_host, nq, k, D_host, I_host));
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    MVDB_TRY(rowset_check(idx, rs));
    DeviceGuard dg(idx->device);
    Workspace* ws = idx->acquire();
    if (!ws) return MVDB_ERR_HIP;
    int rc = 0;
    do {
        if ((rc = stage_queries(idx, ws, q_host, nq))) break;
        const size_t total = (size_t)nq * k;
        if ((rc = ws->out.reserve(total + (total + 1) / 2))) break;
        float* D_dev = reinterpret_cast<float*>(ws->out.p + total);
        if ((rc = rowset_search_core(idx, ws, ws->q.p, nq, k, normalize_q, rs, 0, D_dev, ws->out.p))) break;
        rc = fetch_results(ws, total, D_host, I_host);
    } while (0);
    idx->release(ws);
    return rc;
}

int mvdb_index_search_rowset_device(const mvdb_index* idx, const float* q_dev, int nq, int k, int normalize_q,
                                    const mvdb_rowset* rs, int64_t label_offset, float* D_dev, int64_t* I_dev, void* stream) {
    MVDB_TRY(check_search_args(idx, q_dev, nq, k, D_dev, I_dev));
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    MVDB_TRY(rowset_check(idx, rs));
    DeviceGuard dg(idx->device);
    Workspace* ws = idx->for_stream((hipStream_t)stream);
    if (!ws) return fail(MVDB_ERR_HIP, "workspace allocation failed");
    std::lock_guard<std::mutex> use(ws->use_mu);
    note_capture(ws);
    RetireScope keep(ws->captured ? &ws->retired : nullptr);
    const float* q = q_dev;
    if (idx->ld != idx->d) {
        MVDB_TRY(ws->q.reserve((size_t)nq * idx->ld));
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(((int64_t)nq * idx->ld + 255) / 256)),
                           dim3(256), 0, ws->stream, ws->q.p, q_dev, (int64_t)nq, idx->d, idx->ld);
        MVDB_HIP(hipGetLastError());
        q = ws->q.p;
    }
    return rowset_search_core(idx, ws, q, nq, k, normalize_q, rs, label_offset, D_dev, I_dev);
}

int mvdb_merge_topk_device(int metric, int nlists, int nq, int k, const float* D_dev,
                           int64_t list_stride_D, const int64_t* I_dev, int64_t list_stride_I,
                           float* D_out_dev, int64_t* I_out_dev, int device, void* stream) {
    if (!D_dev || !I_dev || !D_out_dev || !I_out_dev) return fail(MVDB_ERR_ARG, "NULL buffer");
    if (nlists <= 0 || nq <= 0 || k <= 0) return fail(MVDB_ERR_ARG, "non-positive size");
    MVDB_TRY(ensure_device(device));
    DeviceGuard dg(device);
    if (k > kMaxFusedK)  // more results than one wave holds: sort the gathered lists in LDS (collective.hip)
        return launch_merge_di_sort(metric, nlists, nq, k, D_dev, list_stride_D, I_dev, list_stride_I, D_out_dev,
                                    I_out_dev, device, (hipStream_t)stream);
    MergeDIArgs a{D_dev, I_dev, list_stride_D, list_stride_I, nlists, nq, k, metric, D_out_dev, I_out_dev};
    hipLaunchKernelGGL(merge_di_kernel, dim3(nq), dim3(kWave), 0, (hipStream_t)stream, a);
    MVDB_HIP(hipGetLastError());
    return 0;
}

int mvdb_normalize_l2(float* x_host, int64_t n, int d, int device) {
    if (n < 0 || d <= 0) return fail(MVDB_ERR_ARG, "bad shape");
    if (n == 0) return 0;
    if (!x_host) return fail(MVDB_ERR_ARG, "x is NULL");
    mvdb_index* tmp = nullptr;
    MVDB_TRY(mvdb_index_create(d, MVDB_METRIC_IP, device, &tmp));
    int rc = mvdb_index_add(tmp, x_host, n, 1);
    if (!rc) rc = mvdb_index_get_rows(tmp, 0, n, x_host);
    mvdb_index_free(tmp);
    return rc;
}

int mvdb_synth_fill_device(float* out_dev, int64_t n, int d, uint64_t seed, int64_t first_row,
                           int normalize, int device, void* stream) {
    if (!out_dev) return fail(MVDB_ERR_ARG, "out is NULL");
    if (n < 0 || d <= 0 || first_row < 0) return fail(MVDB_ERR_ARG, "bad shape");
    if (d % 4) return fail(MVDB_ERR_ARG, "mvdb_synth_fill_device needs d %% 4 == 0 (dense rows)");
    if (n == 0) return 0;
    MVDB_TRY(ensure_device(device));
    DeviceGuard dg(device);
    const int cus = device_cus(device);
    const int64_t total = n * (d / 4);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, (int64_t)cus * 16));
    hipLaunchKernelGGL(synth_fill_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out_dev, n,
                       (int64_t)d, d, seed, first_row);
    MVDB_HIP(hipGetLastError());
    if (normalize) {
        mvdb_index shape;  // only the geometry fields are read
        shape.d = d;
        shape.ld = d;
        shape.d4 = d / 4;
        shape.device = device;
        MVDB_TRY(normalize_range(&shape, out_dev, n, (hipStream_t)stream));
    }
    return 0;
}

double mvdb_half_eps(int d) { return d > 0 ? half_eps(d) : 0.0; }
int mvdb_half_max_queries(int d) { return d > 0 ? half_max_queries(d) : 0; }

int64_t mvdb_split_rerun_count(void) {
    // diagnostic: the counters live on the devices (the search path never reports to the host); finished work only —
    // synchronise the streams you searched on first (the host API does)
    std::lock_guard<std::mutex> lk(g_rerun_mu);
    int64_t total = 0;
    for (auto& kv : g_rerun_ctr) {
        DeviceGuard dg(kv.first);
        unsigned long long v = 0;
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(&v, kv.second, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess)
            total += (int64_t)v;
    }
    return total;
}

int mvdb_rescue_tile_stats(int64_t* listed, int64_t* total) {
    if (!listed || !total) return fail(MVDB_ERR_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(g_rerun_mu);
    *listed = *total = 0;
    for (auto& kv : g_rerun_ctr) {
        DeviceGuard dg(kv.first);
        unsigned long long v[2] = {0, 0};
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(v, kv.second + 1, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess) {
            *listed += (int64_t)v[0];
            *total += (int64_t)v[1];
        }
    }
    return 0;
}

int mvdb_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return 0;
}

int mvdb_prof_symbol(const char* name, char* out, int len) {
    if (!name || !out || len <= 0) return fail(MVDB_ERR_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto it = g_prof_sym.find(name);
    snprintf(out, (size_t)len, "%s", it == g_prof_sym.end() ? "" : it->second.c_str());
    return 0;
}

int mvdb_prof_read(const char* name, int64_t* launches, double* total_ms) {
    if (!name || !launches || !total_ms) return fail(MVDB_ERR_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int64_t cnt = 0;
    double ms = 0.0;
    for (auto it = g_prof.begin(); it != g_prof.end();) {
        ProfPair& p = it->second;
        if (p.name != name || !p.closed) {
            ++it;
            continue;
        }
        float t = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
            ++cnt;
            ms += t;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
        it = g_prof.erase(it);
    }
    *launches = cnt;
    *total_ms = ms;
    return 0;
}

}  // extern "C"
