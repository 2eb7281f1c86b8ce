// half_scan.hpp — interface of half_scan.hip (the fp16 single-product nomination pass) to mvdb.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"

namespace mvdb {

constexpr int kHalfKeep = 16;      // nominees per (block, query) list and in the running list between phases
constexpr int kHalfRescore = 64;   // nominees per query re-scored in fp32 by half_certify_kernel
constexpr int kHalfMaxK = 12;      // results per query this pass certifies (floors are 16th-best scores: k must stay below 16)

struct HalfScanArgs {
    const float* X;
    int64_t n;            // rows [0, n) may be nominated (rows up to the end of the last tile are READ: index slack)
    int64_t ld;
    const _Float16* qf;   // [nqpad][d] fp16 image of s_q * q (half_queries_kernel), rows >= nq zero
    const float* qinv;    // [nqpad] 1 / (s_q * s_x): scaled score -> score
    float xscale;         // s_x (power of two): corpus elements are multiplied by it before the fp16 rounding
    int nq;
    uint64_t* cand;       // [nq, gridDim.x, 16] nominee keys per block (seed launch: [nq, gridDim.x, 32])
    int64_t tile0;        // 32-row tiles [tile0, tile1)
    int64_t tile1;
    const float* thr0;    // [nq] admission floors, or NULL
    unsigned int* stats;  // NULL, or [2]: list inserts, wave-tiles that reached the slow path (diagnostics)
    const uint32_t* mask = nullptr;  // NULL, or one bit per row (bit r & 31 of word r >> 5): only rows whose bit is set may be nominated
    const _Float16* Xh = nullptr;    // NULL, or the fp16 shadow of the corpus, [n][d] = fp16(xscale * X): the main launches stream IT
                                     // (flat_scan_h16_kernel: half the bytes, no conversion) where a kernel exists (half_shadow_dim)
    // second pass for REFUSED queries (launch_half_rescue): the launch is enabled on the device (*gate > gate_lo), and a query's
    // admission floor is thr0[q] - thr_eps * thr_qn[q] (the k-th exact score of its nominees less the nomination error)
    const int* gate = nullptr;
    int gate_lo = 0;
    float thr_eps = 0.f;
    const float* thr_qn = nullptr;
    const float* hn = nullptr;       // NULL, or [n + slack] |x_r|^2 / 2 per row (L2 metric over the shadow): rows are nominated by
                                     // q.x - |x|^2 / 2, which ranks exactly like the squared distance |q|^2 - 2 (q.x - |x|^2 / 2)
    // tile flags for the rescue pass (round 6, inner product, k <= 16).  A MAIN launch (tflags != NULL) raises bit t of its
    // query's row whenever tile t holds a score within flag_coef |q| below the query's running threshold; the rescue launch walks
    // the tiles some refused query flagged (tile_list / tile_count, written by rescue_tiles_kernel) instead of the whole shadow.
    uint32_t* tflags = nullptr;      // [nq][twords], zeroed by the caller
    int twords = 0;
    const float* flag_qn = nullptr;  // [nq] |q|
    float flag_coef = 0.f;
    const int* tile_list = nullptr;  // rescue launch: the tiles to scan ...
    const int* tile_count = nullptr; // ... and how many
};

struct HalfCertifyArgs {
    const uint64_t* keys;  // [nq, nlists, 16] gated, sorted per-block lists of the last phase (nlists may be 0)
    int nlists;
    const uint64_t* base;  // [nq, 16] running nominees of the earlier phases, or NULL
    const float* thr0;     // [nq] the floor the last phase was gated with (= 16th score of `base`), or NULL
    const float* X;
    int64_t ld;
    int d4;
    const float* q;        // [nq, ld] fp32
    const float* qnorm;    // [nq]
    float eps;             // half_eps(d) * max|x|, rounded up
    int k;
    int64_t label_offset;
    float* D;
    int64_t* I;
    int* uncertified;      // incremented once per query that fails the certificate
    int* failed;           // [nq] set to 1 for a query that fails it
    int l2 = 0;            // 1: the index' metric is squared L2.  Nomination is STILL by inner product (the keys hold approximate
                           // q.x); the nominees are re-scored as sum (q - x)^2 in fp32, ordered by smallest distance, and the
                           // certificate bounds every dropped row's distance from below through |x|^2 >= n2lo:
                           //   d(y) >= |q|^2 + n2lo - 2 (U + eps |q|)  >  r(k-th)        (topk_device.hpp: l2_certified)
    float n2lo = 0.f;      // lower bound of |x|^2 over the stored rows (l2 == 1 only)
                           // l2 == 2 (round 4, the pass ran with HalfScanArgs::hn): the keys hold approximate q.x - |x|^2 / 2 =: s(x),
                           // and d(y) = |q|^2 - 2 s(y) >= |q|^2 - 2 (U + eps |q| + eps_h) for every dropped row, whatever the rows'
                           // norms (the same inequality with n2lo = 0)
    float eps_h = 0.f;     // l2 == 2: bound on the error of the stored |x|^2 / 2 and of the subtraction, absolute
    float* floor_out = nullptr;  // NULL, or [nq]: for a REFUSED query the k-th exact score of its nominees less floor_margin |q| — a
                                 // lower bound of its k-th result in any fp32 summation order, the admission floor of its exact
                                 // re-run and of the rescue pass (-inf where fewer than k nominees were re-scored).  L2 forms
                                 // (round 6): with r the k-th exact DISTANCE of the nominees, every row of the top k has
                                 // q.x >= (|q|^2 + n2lo - r) / 2 (l2 = 1) resp. q.x - |x|^2 / 2 >= (|q|^2 - r) / 2 (l2 = 2): the
                                 // floor in the units the nomination keys are in, rounded down like l2_certified's bound
    float floor_margin = 0.f;    // 2 d 2^-24 max|x|: the two fp32 dot products (this kernel's, the re-run's) may differ by that much
};

// ---- the RESCUE pass: refused queries once more over the shadow, every row above the floor kept and re-scored ----------------
constexpr int kRescueKeep = 32;      // rows per (block, query) list of the rescue pass: a FULL list sends the query on to the exact pass
constexpr int kRescueBlocksPerCu = 2; // most workgroups per CU a rescue launch runs (its lists are sized by it)
constexpr int kRescueQueries = 128;  // compact queries per rescue launch
constexpr int kRescueCap = 8192;     // candidates per query the re-score holds (256 lists x 32)
struct HalfRescueArgs {
    const uint64_t* keys;   // [kRescueQueries, nlists, kRescueKeep]
    int nlists;
    const float* X;
    int64_t ld;
    int d4;
    const float* q;         // [kRescueQueries, ld] the compact queries of this launch
    int k;
    int64_t label_offset;
    float* D;               // [kRescueQueries, k] compact results
    int64_t* I;
    const int* gate;        // number of refused queries of the call
    int gate_lo;            // first compact query of this launch
    int* need;              // one word per exact pass of per_pass queries: raised when one of its queries stays unanswered
    int per_pass = 32;      // queries per exact pass (32 up to d = 512, 16 at the wider dimensions; L2: 1 — one gated scan per query)
    int l2 = 0;             // != 0: squared-L2 index — candidates are re-scored as sum (q - x)^2, smallest first
};
int launch_half_rescue_scan(int d, const HalfScanArgs& a, int device, hipStream_t stream, int* nblocks_out);
// The tile lists of a call's rescue launches (one per kRescueQueries compact refused queries: grid.y = slots): slot s lists the
// tiles whose bit some refused query [128 s, 128 s + 128) of the call raised (tflags == NULL: every tile), plus the seed's tiles.
struct RescueTilesArgs {
    const uint32_t* tflags;   // [queries of the call][twords] or NULL
    int twords;
    const int64_t* map;       // compact refused query -> query of the call
    const int* nfail;         // refused queries of the call
    int64_t seed_tiles;       // tiles [0, seed_tiles) were scanned by the seed launch only: always listed
    int64_t ntiles;
    int* lists;               // [slots][ntiles]
    int* counts;              // [slots], zeroed by the caller
    unsigned long long* stats;  // NULL, or [2]: tiles listed / tiles there were, summed over live slots (diagnostics)
};
int launch_rescue_tiles(const RescueTilesArgs& a, int slots, hipStream_t stream);
int launch_half_rescue_certify(const HalfRescueArgs& a, hipStream_t stream);
bool half_rescue_dim(int d);

// queries per corpus pass of the widest instantiation for dimension d (0: no kernel for this d)
int half_max_queries(int d);
// queries per pass the launcher will use for a chunk of `nq` queries (a multiple of 32 * d / 128)
int half_chunk_queries(int d, int nq);
double half_eps(int d);
float half_xscale(float row_norm_bound);

int launch_half_queries(const float* q, int64_t ld, int d, int nq, int nqpad, float xscale, _Float16* qf, float* qnorm,
                        float* qinv, hipStream_t stream);
// seed: one tile per block over [tile0, tile1), every score dumped ([nq, blocks, 32] keys); *nblocks_out = blocks
int launch_half_scan(int d, int nqpad, bool seed, const HalfScanArgs& a, const Knobs& kn, int device, hipStream_t stream, int* nblocks_out);
int launch_half_certify(const HalfCertifyArgs& a, int nq, hipStream_t stream);
// the fp16 shadow of rows [0, n) of X (ld floats per row) into Xh (d halves per row); dimensions the shadow kernels serve
int launch_half_shadow(const float* X, int64_t ld, int d, int64_t n, float xscale, _Float16* Xh, int device, hipStream_t stream);
bool half_shadow_dim(int d);
// Hn[r] = |X[r]|^2 / 2 for rows [0, n) (fp32; a wave per row, lane-strided fmas + butterfly: relative error < 2^-18 for d <= 4096)
int launch_half_norms(const float* X, int64_t ld, int d, int64_t n, float* Hn, int device, hipStream_t stream);

}  // namespace mvdb
