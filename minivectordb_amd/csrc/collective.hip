// collective.hip — the exchange step of the row-partitioned multi-GPU search:
// ONE RCCL all-gather of every rank's packed per-shard top-k block over xGMI, issued from inside libmvdb.so
// (ncclAllGather on the caller's stream), followed by the k-way merge every rank runs redundantly.
//
// The reference never partitions a search (minivectordb/sharded_vector_database.py:598-662 searches ONE stacked
// matrix), so these entry points have no reference counterpart; the result contract they keep is that function's:
// top-k of the union of all shards, best first, ties to the lower global row.
//
// RCCL is bound at run time (dlopen of the librccl the process already holds — PyTorch-ROCm ships one — else the
// system one), so libmvdb.so keeps no link-time dependency on it and a single-GPU process never loads it.
#include <dlfcn.h>

#include <algorithm>
#include <mutex>

#include "common.hpp"
#include "topk_device.hpp"

using namespace mvdb;

namespace {

// ---- the five RCCL entry points used (signatures of /opt/rocm/include/rccl/rccl.h) ------------------------------
struct RcclUniqueId {
    char internal[128];
};
typedef struct ncclComm* RcclComm;
typedef int (*fn_GetUniqueId)(RcclUniqueId*);
typedef int (*fn_CommInitRank)(RcclComm*, int, RcclUniqueId, int);
typedef int (*fn_CommDestroy)(RcclComm);
typedef int (*fn_AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t);
typedef const char* (*fn_GetErrorString)(int);
constexpr int kRcclUint8 = 1;  // ncclUint8

struct Rccl {
    void* handle = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_AllGather AllGather = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
    std::string why;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)  // the copy this process already mapped (same HIP runtime as the caller's tensors)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!r.handle)
            for (const char* n : names)
                if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.handle) {
            const char* de = dlerror();  // ONE call: dlerror() clears the error state
            r.why = std::string("librccl.so not loadable: ") + (de ? de : "unknown");
            return;
        }
        r.GetUniqueId = (fn_GetUniqueId)dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (fn_CommInitRank)dlsym(r.handle, "ncclCommInitRank");
        r.CommDestroy = (fn_CommDestroy)dlsym(r.handle, "ncclCommDestroy");
        r.AllGather = (fn_AllGather)dlsym(r.handle, "ncclAllGather");
        r.GetErrorString = (fn_GetErrorString)dlsym(r.handle, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
            r.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
            r.handle = nullptr;
        }
    });
    return &r;
}

int rccl_fail(const char* what, int rc) {
    Rccl* r = rccl();
    return fail(MVDB_ERR_HIP, "%s failed: %s", what, r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
}

// ---- merge of more than 64 results per query: sort nlists * k keys in LDS ----------------------------------------
// One block per query.  Key = (score image << 32) | ~(list * k + slot): ties resolve to the lower list, then the
// lower slot — ascending global row, because shard bases ascend with the list index and every list is sorted
// (score desc, row asc).  P = pow2ceil(nlists * k) keys of 8 bytes in dynamic LDS (<= 128 KiB of the CU's 160 KiB).
struct MergeSortArgs {
    const float* D;
    const int64_t* I;
    int64_t strideD, strideI;
    int nlists, nq, k, metric, P;
    float* Dout;
    int64_t* Iout;
};

__global__ __launch_bounds__(1024) void merge_di_sort_kernel(MergeSortArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
    const int qi = blockIdx.x;
    const int total = a.nlists * a.k;
    for (int i = threadIdx.x; i < a.P; i += blockDim.x) {
        uint64_t key = 0;
        if (i < total) {
            const int l = i / a.k, j = i - l * a.k;
            const int64_t src = (int64_t)qi * a.k + j;
            if (a.I[l * a.strideI + src] >= 0) {
                const float d = a.D[l * a.strideD + src];
                key = make_key(a.metric == 0 ? d : -d, (uint32_t)i);
            }
        }
        skeys[i] = key;
    }
    __syncthreads();
    for (int size = 2; size <= a.P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < a.P / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t x = skeys[lo], y = skeys[hi];
                if ((x < y) == desc) {
                    skeys[lo] = y;
                    skeys[hi] = x;
                }
            }
            __syncthreads();
        }
    }
    for (int r = threadIdx.x; r < a.k; r += blockDim.x) {
        const uint64_t key = skeys[r];
        float d = a.metric == 0 ? -3.402823466e+38f : 3.402823466e+38f;
        int64_t id = -1;
        if (key) {
            const int i = (int)key_row(key);
            const int l = i / a.k, j = i - l * a.k;
            const int64_t src = (int64_t)qi * a.k + j;
            d = a.D[l * a.strideD + src];
            id = a.I[l * a.strideI + src];
        }
        a.Dout[(int64_t)qi * a.k + r] = d;
        a.Iout[(int64_t)qi * a.k + r] = id;
    }
}

}  // namespace

struct mvdb_comm {
    RcclComm comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

namespace mvdb {
// called by mvdb_merge_topk_device (mvdb.hip) for k > 64
int launch_merge_di_sort(int metric, int nlists, int nq, int k, const float* D, int64_t strideD, const int64_t* I,
                         int64_t strideI, float* Dout, int64_t* Iout, int device, hipStream_t stream) {
    int64_t P = 1;
    while (P < (int64_t)nlists * k) P <<= 1;
    if (P > 16384)
        return fail(MVDB_ERR_ARG, "merge of %d lists x %d results exceeds the 16384 keys one block sorts in LDS", nlists, k);
    const size_t lds = (size_t)P * sizeof(uint64_t);
    static std::mutex mu;
    static std::vector<std::pair<int, size_t>> done;  // (device, raised limit)
    if (lds > 48 * 1024) {
        std::lock_guard<std::mutex> lk(mu);
        bool ok = false;
        for (auto& e : done) ok |= e.first == device && e.second >= lds;
        if (!ok) {
            MVDB_HIP(hipFuncSetAttribute((const void*)merge_di_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         128 * 1024));
            done.emplace_back(device, (size_t)128 * 1024);
        }
    }
    MergeSortArgs a{D, I, strideD, strideI, nlists, nq, k, metric, (int)P, Dout, Iout};
    hipLaunchKernelGGL(merge_di_sort_kernel, dim3(nq), dim3(1024), lds, stream, a);
    MVDB_HIP(hipGetLastError());
    return 0;
}
}  // namespace mvdb

extern "C" {

int mvdb_comm_available(void) {
    Rccl* r = rccl();
    if (!r->handle) return fail(MVDB_ERR_HIP, "%s", r->why.c_str());
    return 0;
}

int mvdb_comm_unique_id(unsigned char* out128) {
    if (!out128) return fail(MVDB_ERR_ARG, "out is NULL");
    Rccl* r = rccl();
    if (!r->handle) return fail(MVDB_ERR_HIP, "%s", r->why.c_str());
    RcclUniqueId id;
    const int rc = r->GetUniqueId(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(out128, id.internal, sizeof(id.internal));
    return 0;
}

int mvdb_comm_create(const unsigned char* id128, int rank, int world, int device, mvdb_comm** out) {
    if (!out) return fail(MVDB_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!id128) return fail(MVDB_ERR_ARG, "unique id is NULL");
    if (world <= 0 || rank < 0 || rank >= world) return fail(MVDB_ERR_ARG, "rank %d outside [0,%d)", rank, world);
    MVDB_TRY(ensure_device(device));
    Rccl* r = rccl();
    if (!r->handle) return fail(MVDB_ERR_HIP, "%s", r->why.c_str());
    DeviceGuard dg(device);
    RcclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    RcclComm c = nullptr;
    const int rc = r->CommInitRank(&c, world, id, rank);
    if (rc) return rccl_fail("ncclCommInitRank", rc);
    mvdb_comm* m = new mvdb_comm();
    m->comm = c;
    m->rank = rank;
    m->world = world;
    m->device = device;
    *out = m;
    return 0;
}

int mvdb_comm_free(mvdb_comm* c) {
    if (!c) return 0;
    Rccl* r = rccl();
    if (c->comm && r->handle) {
        DeviceGuard dg(c->device);
        (void)r->CommDestroy(c->comm);
    }
    delete c;
    return 0;
}

int mvdb_comm_rank(const mvdb_comm* c) { return c ? c->rank : -1; }
int mvdb_comm_world(const mvdb_comm* c) { return c ? c->world : -1; }

int mvdb_allgather_topk(mvdb_comm* c, const void* local_dev, void* gathered_dev, int64_t nbytes_per_rank,
                        void* stream) {
    if (!c || !c->comm) return fail(MVDB_ERR_ARG, "communicator is NULL");
    if (!local_dev || !gathered_dev) return fail(MVDB_ERR_ARG, "NULL buffer");
    if (nbytes_per_rank <= 0) return fail(MVDB_ERR_ARG, "non-positive block size");
    Rccl* r = rccl();
    DeviceGuard dg(c->device);
    const int rc = r->AllGather(local_dev, gathered_dev, (size_t)nbytes_per_rank, kRcclUint8, c->comm, (hipStream_t)stream);
    if (rc) return rccl_fail("ncclAllGather", rc);
    return 0;
}

}  // extern "C"
