/*
 * mvdb.h — C-ABI of libmvdb.so, the MI355X (gfx950) replacement for the native calls on
 * MiniVectorDB's embed+search hot path.
 *
 * The reference (cnmoro/MiniVectorDB @ 2024-10-08) has no FFI of its own: the hot path is
 * executed inside third-party wheels (faiss-cpu, transformers/torch).  Each entry point below
 * names the reference call site it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; mvdb_last_error() returns a
 *     thread-local, NUL-terminated description of the last failure on the calling thread;
 *   - plain pointers and sizes only; "host" pointers are ordinary CPU memory owned by the caller
 *     for the duration of the call, "dev" pointers are device memory on the index' GPU;
 *   - the library copies on add (as faiss IndexFlatIP.add does) and owns all device memory;
 *   - result conventions match faiss: rows sorted by score descending, labels int64,
 *     missing slots label -1 / score -FLT_MAX (IP) or +FLT_MAX (L2);
 *   - ties are broken by ascending row number (deterministic);
 *   - mvdb_index_search* are re-entrant on one index from many host threads; add / reset /
 *     free take the index exclusively.
 *   - There is NO CPU fallback: without a usable HIP device every compute call fails.
 */
#ifndef MVDB_H
#define MVDB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVDB_METRIC_IP 0 /* inner product, larger is better (faiss.IndexFlatIP)            */
#define MVDB_METRIC_L2 1 /* squared L2, smaller is better (extension; not in the reference) */

#define MVDB_OK 0
#define MVDB_ERR_ARG 1     /* bad argument                       */
#define MVDB_ERR_HIP 2     /* HIP runtime / device failure       */
#define MVDB_ERR_NODEVICE 3 /* no usable gfx950 device            */
#define MVDB_ERR_OOM 4     /* device allocation failed           */

typedef struct mvdb_index mvdb_index;
typedef struct mvdb_encoder mvdb_encoder;

/* ---- library / device ------------------------------------------------------------------- */

/* Thread-local description of the last error on this thread ("" if none). */
const char* mvdb_last_error(void);

/* ABI version of this header (bumped on incompatible change). */
int mvdb_abi_version(void);

/* Number of visible HIP devices; fails with MVDB_ERR_NODEVICE when there is none. */
int mvdb_device_count(int* count);

/* ---- flat index --------------------------------------------------------------------------
 * Replaces faiss.IndexFlatIP(d)                  minivectordb/vector_database.py:43, :511
 *                                                minivectordb/sharded_vector_database.py:80, :639
 * The corpus lives in HBM as one row-major fp32 matrix, row stride = d rounded up to a
 * multiple of 4 floats (16-byte rows), zero padded. */
int mvdb_index_create(int d, int metric, int device, mvdb_index** out);
int mvdb_index_free(mvdb_index* idx);

/* The MVDB_* tuning / A-B hooks of the search path (docs/DESIGN_NOTES.md section 7) are read from the environment ONCE, by
 * mvdb_index_create; the search path itself never calls getenv.  This re-reads them for an existing index (A/B runs and tests
 * that flip a hook inside one process).  No reference counterpart. */
int mvdb_index_reload_env(mvdb_index* idx);

/* Per-index switches a caller sets in code rather than through the environment (the value survives until the next
 * mvdb_index_reload_env).  No reference counterpart.
 *   "shadow_single_query" (0 | 1, default 0): ONE query per call also goes through the certified nomination pass over the fp16
 *       shadow of the rows (>= 500,000 rows at d = 256 / 384 / 512): the same ids and fp32 scores as the exact scan — nominees
 *       re-scored in fp32, every result certified, uncertified queries re-run exactly — at about half the bytes per query
 *       (10M x 512: 2.84 -> 1.58 ms), for 2 more bytes per stored element.  Off by default: the single-query scan is the exact
 *       fp32 kernel the headline roofline is defined on.
 *   "half_shadow" (0 | 1, default 1): 0 = batches nominate from the fp32 rows (no second copy of the corpus).
 *   "compact_bytes" (> 0, default 512 MiB): staging buffer of mvdb_index_remove_rows. */
int mvdb_index_set_option(mvdb_index* idx, const char* name, long long value);

/* The opt-in single-query route over the shadow ("shadow_single_query") costs more than the exact scan whenever its certificate
 * is refused (clustered / duplicate-heavy corpora: the nomination pass AND the exact scan).  The library watches the refusals
 * (a device counter mirrored into host-mapped memory: no synchronisation) in windows of 32 such calls; when half of a window
 * was refused the next 512 single queries take the exact scan directly, then the route probes again.  Returns how many times
 * that has happened on this index (diagnostics; -1 for NULL).  Results are identical either way. */
long long mvdb_index_single_route_suspensions(const mvdb_index* idx);

/* Drop all rows (capacity is kept). */
int mvdb_index_reset(mvdb_index* idx);

/* Rows currently stored / dimension / device ordinal. */
int64_t mvdb_index_ntotal(const mvdb_index* idx);
int mvdb_index_dim(const mvdb_index* idx);
int mvdb_index_device(const mvdb_index* idx);

/* Rows held in the index' fp16 SHADOW (0: none).  An index of width d = 128 / 256 / 384 / 512 / 640 / 768 / 896 / 1024 (unpadded
 * rows) that answers a batch the certified pass takes — 2+ queries from 500k rows, 8+ from 100k, 33+ below (thresholds measured
 * at d = 512 and applied to every shadow width: the pass is bound by the shadow's bytes at all of them) — keeps, lazily, from
 * the first such search on (that search builds it: one blocking conversion pass and a hipMalloc inside an otherwise asynchronous
 * call), an fp16 copy of its rows next to the fp32 matrix (2 more bytes per element = +50 % of the index's device memory): the
 * nomination pass of those batches streams it (the fp32 matrix stays the home of the exact scans and of every returned score).
 * add extends it, remove_rows / reset / a re-allocation drop it (the next batch search rebuilds it: ~5 ms per 10M x 512 rows).
 * Without a shadow — option "half_shadow" = 0, MVDB_DISABLE_HALF_SHADOW=1, or its allocation failed — batches are answered by
 * the exact fp32 passes (32 queries per corpus pass; until round 6 a second nomination generation read the fp32 rows).
 * No reference counterpart. */
int64_t mvdb_index_shadow_rows(const mvdb_index* idx);

/* Reserve device capacity for at least n rows in total (amortises repeated add). */
int mvdb_index_reserve(mvdb_index* idx, int64_t n);

/* Append n rows from host memory x[n,d] (C-contiguous fp32).  normalize != 0 L2-normalises each
 * appended row on the device first (zero-norm rows are left untouched).
 * Replaces faiss.normalize_L2(self.embeddings) + index.add(self.embeddings)
 *                                                minivectordb/vector_database.py:45-46, :512
 *                                                minivectordb/sharded_vector_database.py:82-83, :640 */
int mvdb_index_add(mvdb_index* idx, const float* x_host, int64_t n, int normalize);

/* Same, rows already in device memory (x_dev[n,d], dense, same GPU). */
int mvdb_index_add_device(mvdb_index* idx, const float* x_dev, int64_t n, int normalize);

/* Append n synthetic rows generated on the device by the counter-based generator documented in
 * DESIGN.md (element (i,j) of the stream `seed` with i = first_row + local row); bit-identical
 * to oracle/synth.py on the host before normalisation.  Used by bench.py and the full-size
 * tests so that 10M x 512 corpora never cross PCIe.  No reference counterpart. */
int mvdb_index_add_synthetic(mvdb_index* idx, int64_t n, uint64_t seed, int64_t first_row,
                             int normalize);

/* Copy rows [row0,row0+n) back to host memory out[n,d] (what faiss calls reconstruct_n);
 * used for get_vector after the in-place normalisation side effect
 *                                                minivectordb/vector_database.py:45, :49-55 */
int mvdb_index_get_rows(const mvdb_index* idx, int64_t row0, int64_t n, float* out_host);

/* Remove the given rows (ascending or not, duplicates rejected) and compact the matrix so the
 * remaining rows keep their relative order — the numbering np.delete leaves behind.  Rows before
 * the first deleted one do not move.  One row (the reference's own delete), a run of rows or up to 8
 * scattered rows: the tail is shifted in place in ONE pass, every byte read once and written once
 * (row 5 of 10M x 512: 8.2 ms).  More: the tail is compacted in place through a bounded staging buffer
 * (512 MiB, kept by the index; 14.5 - 18 ms there).  Neither needs memory proportional to the index.
 *                                                minivectordb/vector_database.py:126, :139-152 */
int mvdb_index_remove_rows(mvdb_index* idx, const int64_t* rows_host, int64_t m);

/* k nearest rows for nq queries.  q_host[nq,d], D_host[nq,k], I_host[nq,k].
 * normalize_q != 0 L2-normalises each query on the device first.
 * Every path returns exact-fp32 scores of the exact top-k.  Which pass answers a call (csrc/mvdb.hip: search_core):
 *   1 query                      the exact fp32 scan (the reference's call shape; option "shadow_single_query" sends it through
 *                                the certified pass too, which suspends itself while its certificates are being refused);
 *   2+ queries, k <= 32, a width with a shadow (previous comment), enough rows to bury the pass's fixed cost (2 queries from
 *   500k rows, 8 from 100k, 33 below)
 *                                the CERTIFIED pass, 128 / 256 queries per pass over the fp16 shadow: ONE fp16 product nominates
 *                                64 rows per query, fp32 re-scores decide, a worst-case bound (mvdb_half_eps) certifies each
 *                                query.  A query whose certificate is refused (near-duplicate neighbourhoods) goes to the RESCUE
 *                                pass — once more over the shadow (inner product, k <= 16, ~400k rows and more, from the second
 *                                refused call of an index on: over the 32-row tiles the certified pass flagged for it, a few
 *                                percent of them — the flags are kept only while the index refuses certificates), every row above its floor kept
 *                                and re-scored in fp32: exact — and, where that cannot hold its neighbourhood, to a device-gated
 *                                exact fp32-MFMA pass
 *                                (mvdb_split_rerun_count counts the chunks that held a refused query);
 *   other batches                exact fp32 passes on the matrix cores: 32 (d <= 512) / 16 (d <= 1024) queries per corpus pass
 *                                at d = 64 and the multiples of 128; elsewhere the GEMM-tiled exact scan from 6 queries
 *                                (128 per launch, k <= 16), else one query at a time;  k > 64: scores + radix select.
 * A batch returns what nq separate calls return — ids identical AWAY FROM fp32 NEAR-TIES: rows whose scores lie within
 * ~2e-6 of each other (exact duplicates aside: those tie exactly and come back lowest row first on every path) may be ranked
 * either way by two fp32 summation orders, so on near-duplicate corpora a batch and a single call can swap such rows
 * (tests/bigcheck.py adjudicates every difference in float64).
 * MVDB_METRIC_L2 (an extension: the reference builds IndexFlatIP only): one query sums (q - x)^2 directly; several queries
 * share corpus passes — on the certified pass where the rows have one norm (nomination by inner product, distance re-score,
 * norm-range certificate) or, for ANY norms, by q.x - |x|^2 / 2 with per-row offsets beside the shadow, also under a bitmap;
 * else on the fp32 matrix cores as |q|^2 + |x|^2 - 2 q.x (d % 128 == 0, d <= 768; differences of the two forms are at the
 * rounding level of the norms).
 * Replaces faiss.normalize_L2(embedding) + index.search(embedding, search_k)
 *                                                minivectordb/vector_database.py:475, :497
 *                                                minivectordb/sharded_vector_database.py:604, :626 */
int mvdb_index_search(const mvdb_index* idx, const float* q_host, int nq, int k, int normalize_q,
                      float* D_host, int64_t* I_host);

/* Same with every buffer in device memory and the work enqueued on `stream` (a hipStream_t, or
 * NULL for the legacy default stream).  label_offset is added to every label (global row number of
 * this shard's first row).  This is the entry point the one-process-per-GPU sharded search uses
 * ahead of its RCCL all-gather.
 * Synchronisation: NONE, whatever the batch size.  The host never reads a certification flag: the queries a
 * certified batch pass could not certify are compacted, re-run on the exact kernels and scattered back by launches
 * that are enabled ON THE DEVICE (they return at once when every query certified).  The call can therefore be captured
 * into a hipGraph — after one eager call of the same shape on that stream has sized its workspace (allocation is not
 * capturable) — e.g. encoder forward -> search as ONE graph.  A captured graph names the stream's workspace buffers: from
 * the first capture on, that workspace never frees a buffer it outgrows (a later, LARGER eager call on the same stream
 * allocates new ones and parks the old until the index goes), so replaying the earlier graph stays valid.
 * One search at a time per (index, stream): concurrent calls naming the same stream are serialised.
 * Mutators (add / remove_rows / reset / reserve) wait for the searches enqueued on THIS index — its streams only, no
 * device-wide synchronise — and do their own work on a private non-blocking stream. */
int mvdb_index_search_device(const mvdb_index* idx, const float* q_dev, int nq, int k,
                             int normalize_q, int64_t label_offset, float* D_dev, int64_t* I_dev,
                             void* stream);

/* Search restricted to the m listed rows of the resident corpus (labels returned are positions
 * in rows_host[], exactly what the reference's throw-away sub-index returns).
 * Replaces self.embeddings[list(filtered)] -> IndexFlatIP.add -> search
 *                                                minivectordb/vector_database.py:510-514
 *                                                minivectordb/sharded_vector_database.py:636-642 */
int mvdb_index_search_subset(const mvdb_index* idx, const float* q_host, int nq, int k,
                             int normalize_q, const int64_t* rows_host, int64_t m, float* D_host,
                             int64_t* I_host);

/* Device-resident variant of mvdb_index_search_subset for the one-process-per-GPU sharded search: queries, the row
 * list (rows_dev[m], int64, every entry in [0, ntotal) — validated by the caller, this entry point does not
 * synchronise) and outputs live in device memory, work is enqueued on `stream`.  map_labels == 0: labels are
 * positions in rows_dev[] (+ label_offset), as above; map_labels != 0: labels are rows_dev[position] + label_offset,
 * i.e. global row numbers of a shard whose first row is label_offset.
 * Replaces the per-shard half of             minivectordb/sharded_vector_database.py:634-649 */
int mvdb_index_search_subset_device(const mvdb_index* idx, const float* q_dev, int nq, int k, int normalize_q,
                                    const int64_t* rows_dev, int64_t m, int map_labels, int64_t label_offset,
                                    float* D_dev, int64_t* I_dev, void* stream);

/* Search restricted to the rows whose bit is set in a BITMAP over the resident corpus: mask[(ntotal + 63) / 64] 64-bit
 * words, bit (r & 63) of word r >> 6 = row r; bits at or beyond ntotal are ignored.  Every row is scored in ONE pass at
 * the full scan's rate and rows with a clear bit are never offered to the top-k — the form for filters that keep MOST
 * rows (the reference's exclude-filters: a row list of ~ntotal entries gathered row by row is slower than the scan it
 * avoids, and its 8 bytes per row are the upload).  labels == 0: positions in the ascending list of the set rows — exactly
 * what mvdb_index_search_subset returns for that list; labels == 1: row numbers.  Exact score ties resolve to the lower
 * row number in both.  Fewer than k rows selected: the tail is -1 / -FLT_MAX (IP) as everywhere.
 * nq > 1 (inner product, d % 128 == 0, k <= 64): the batch shares corpus passes like an unfiltered one — 33+ queries on the
 * certified fp16 pass (the bit is consulted where a row is about to be nominated), fewer on the fp32-MFMA pass, which also
 * re-runs uncertified queries under the bitmap; other shapes answer the bitmap one query at a time.
 * Replaces the same per-query sub-index as mvdb_index_search_subset
 *                                                minivectordb/vector_database.py:508-523 (exclude filters :354-386)
 *                                                minivectordb/sharded_vector_database.py:634-649 */
int mvdb_index_search_masked(const mvdb_index* idx, const float* q_host, int nq, int k, int normalize_q,
                             const uint64_t* mask_host, int labels, float* D_host, int64_t* I_host);

/* Device-resident variant (mask_dev, queries and outputs in device memory, enqueued on `stream`, no synchronisation);
 * labels == 1: row number + label_offset. */
int mvdb_index_search_masked_device(const mvdb_index* idx, const float* q_dev, int nq, int k, int normalize_q,
                                    const uint64_t* mask_dev, int labels, int64_t label_offset, float* D_dev,
                                    int64_t* I_dev, void* stream);

/* A filter's rows kept RESIDENT on the device, so that consecutive searches under the same filter (the reference
 * re-evaluates its filter and re-gathers the rows for every query, vector_database.py:477-523) pay neither the host pass
 * over the list nor its upload again.  excluded == 0: the m listed rows (labels order ties by list position, as
 * mvdb_index_search_subset); excluded != 0: every row BUT the listed ones (an exclude-filter: m is small, the set is
 * ~ntotal rows).  The library picks the representation: a bitmap (n / 8 bytes; one full-rate pass per search) for excluded
 * sets and for sorted lists that keep >= 90 % of the rows, a device row list otherwise — a SORTED list (>= 1 row in 64) also
 * carries its bitmap, under which a BATCH of queries shares corpus passes where that beats one gathered scan per query (10M x
 * 512, 30 % of the rows: 64 queries 1.7 ms instead of 58 ms).  A set belongs to the index state it
 * was built against: rows appended later are not part of it; after a removal (rows renumbered) searching it fails with
 * MVDB_ERR_ARG. */
typedef struct mvdb_rowset mvdb_rowset;
int mvdb_rowset_create(const mvdb_index* idx, const int64_t* rows_host, int64_t m, int excluded, mvdb_rowset** out);
int64_t mvdb_rowset_size(const mvdb_rowset* rs);      /* rows selected */
int mvdb_rowset_is_bitmap(const mvdb_rowset* rs);
int mvdb_rowset_free(mvdb_rowset* rs);
/* labels are ROW NUMBERS of the index (not positions).  Replaces the same call sites as mvdb_index_search_subset. */
int mvdb_index_search_rowset(const mvdb_index* idx, const float* q_host, int nq, int k, int normalize_q,
                             const mvdb_rowset* rs, float* D_host, int64_t* I_host);
/* Device-resident variant (queries and outputs in device memory, enqueued on `stream`, no synchronisation; labels are row
 * numbers + label_offset): what a rank of the row-partitioned search runs for a filter whose LOCAL rows it keeps resident.
 * Replaces the per-shard half of             minivectordb/sharded_vector_database.py:634-649 */
int mvdb_index_search_rowset_device(const mvdb_index* idx, const float* q_dev, int nq, int k, int normalize_q,
                                    const mvdb_rowset* rs, int64_t label_offset, float* D_dev, int64_t* I_dev,
                                    void* stream);

/* Merge `nlists` sorted top-k lists per query into one [nq,k] result on the device.  List l lives
 * at D_dev + l*list_stride_D (floats, [nq,k]) and I_dev + l*list_stride_I (int64, [nq,k]) — the
 * layout one RCCL all-gather of each rank's packed {I,D} block produces.  Labels must already be
 * global; lists must be ordered by ascending shard base (ties resolve to the lower label).
 * k <= 64: one wave per query; larger k: nlists * k <= 16384 keys sorted in LDS by one block per query.
 * No reference counterpart (the reference never partitions a search). */
int mvdb_merge_topk_device(int metric, int nlists, int nq, int k, const float* D_dev,
                           int64_t list_stride_D, const int64_t* I_dev, int64_t list_stride_I,
                           float* D_out_dev, int64_t* I_out_dev, int device, void* stream);

/* ---- exchange step of the row-partitioned multi-GPU search -------------------------
 * One process per GPU; rank r holds rows [base_r, base_r + n_r) resident and answers from them
 * (mvdb_index_search_device with label_offset = base_r); ONE all-gather of every rank's packed result block
 * ([I: nq*k int64 | D: nq*k fp32], 12*k bytes per query) over xGMI; mvdb_merge_topk_device on every rank.
 * The communicator is RCCL's: rank 0 draws a unique id (mvdb_comm_unique_id), the launcher hands the 128 bytes
 * to every rank (torch.distributed / a file / MPI — any side channel), every rank calls mvdb_comm_create.
 * RCCL is bound at run time (dlopen); a single-GPU process never loads it.  No reference counterpart: the
 * reference searches one stacked matrix (minivectordb/sharded_vector_database.py:598-662); the contract kept is
 * that function's result — top-k of the union, ties to the lower global row. */
typedef struct mvdb_comm mvdb_comm;
int mvdb_comm_available(void); /* 0 when librccl could be bound in this process (every rank checks BEFORE the collective init) */
int mvdb_comm_unique_id(unsigned char* out128);
int mvdb_comm_create(const unsigned char* id128, int rank, int world, int device, mvdb_comm** out);
int mvdb_comm_free(mvdb_comm* comm);
int mvdb_comm_rank(const mvdb_comm* comm);
int mvdb_comm_world(const mvdb_comm* comm);

/* ncclAllGather of nbytes_per_rank bytes from local_dev into gathered_dev[world * nbytes_per_rank] (rank order),
 * enqueued on `stream`; returns without synchronising. */
int mvdb_allgather_topk(mvdb_comm* comm, const void* local_dev, void* gathered_dev, int64_t nbytes_per_rank,
                        void* stream);

/* In-place row-wise L2 normalisation of host matrix x[n,d] on the device.
 * Replaces faiss.normalize_L2                     minivectordb/vector_database.py:45, :475 */
int mvdb_normalize_l2(float* x_host, int64_t n, int d, int device);

/* Device generator of the synthetic stream into dense device memory out_dev[n,d] (queries). */
int mvdb_synth_fill_device(float* out_dev, int64_t n, int d, uint64_t seed, int64_t first_row,
                           int normalize, int device, void* stream);

/* ---- profiling hooks (bench.py's roofline leg) ---------------------------------------------
 * When enabled, every launch of the dominant kernels is bracketed by hipEvents on the launch
 * stream.  mvdb_prof_read drains the finished pairs of kernel `name` ("ip_scan", "ip_scan_mfma",
 * "ip_scan_gemm", "ip_scan_half", "ip_scan_half_seed", "ip_scan_rescue", "ip_scan_rerun", "ip_scan_scores",
 * "encoder") and returns the number of launches and their summed duration. */
int mvdb_prof_enable(int on);
int mvdb_prof_read(const char* name, int64_t* launches, double* total_ms);
/* The kernel instantiation last launched under label `name` while profiling was on, as rocprofv3 prints it
 * ("flat_scan_kernel<64, 2, 2, 0, 0, true, 0, false>"; "" if none): bench.py refuses a committed PMC profile whose kernel
 * differs from what the timed run launched. */
int mvdb_prof_symbol(const char* name, char* out, int len);

/* Number of chunks (up to 256 queries) of the certified batch passes that held a query which failed certification — those
 * queries were answered by the rescue pass or re-run on the exact fp32 kernels — since the library was loaded.
 * Diagnostic only. */
int64_t mvdb_split_rerun_count(void);

/* Tiles (32 rows) the rescue launches were handed since the library was loaded, and the tiles they would have scanned without
 * the tile flags the certified pass keeps (inner product, k <= 16, ~400k rows and more, while the index has been refusing
 * certificates; MVDB_TILE_FLAGS=1 / 0: always / never).  Diagnostic only; synchronises the devices. */
int mvdb_rescue_tile_stats(int64_t* listed, int64_t* total);

/* The certificate's error bound per unit |q| * max|x| at dimension d for the fp16 single-product nomination pass
 * (half_scan.hip: both operands rounded to fp16, d products accumulated in fp32, 64 nominees per query re-scored
 * in fp32): rounding of both operands, elements below fp16's normal range, worst-case fp32 accumulation, the fp32
 * re-score, |q| and the comparison (docs/DESIGN_NOTES.md section 4.3d; tests/test_split_bound.py).  Diagnostic.
 * mvdb_half_max_queries: queries per corpus pass of that pass at dimension d, 0 where it has no kernel. */
double mvdb_half_eps(int d);
int mvdb_half_max_queries(int d);

/* ---- encoder (BERT-architecture sentence encoder: e5-small / e5-large) ---------------------
 * Replaces self.model(**batch_dict) + average_pool + F.normalize
 *                                                minivectordb/embedding_model.py:66-70, :50-53 */
typedef struct mvdb_encoder_cfg {
    int vocab_size;
    int hidden;        /* H  */
    int layers;        /* L  */
    int heads;         /* nh (head_dim = H / nh) */
    int intermediate;  /* FFN width */
    int max_positions;
    int type_vocab;
    int position_offset; /* 0 for BERT; padding_idx+1 for XLM-R style position ids */
    float ln_eps;
    int pooling;         /* 0 = attention-masked mean (e5, embedding_model.py:50-53);
                            1 = first valid token / CLS (BGE-M3 dense_vecs, embedding_model.py:74-78) */
} mvdb_encoder_cfg;

/* Weight table: device pointers (fp32, PyTorch nn.Linear layout [out,in]) in the order given by
 * mvdb_encoder_weight_name(i) for i in [0, mvdb_encoder_weight_count(cfg)). The encoder keeps
 * the pointers (the caller — a torch state_dict — owns the memory). */
int mvdb_encoder_weight_count(const mvdb_encoder_cfg* cfg);
const char* mvdb_encoder_weight_name(const mvdb_encoder_cfg* cfg, int i);
int mvdb_encoder_create(const mvdb_encoder_cfg* cfg, const void* const* weight_ptrs_dev, int device,
                        mvdb_encoder** out);
int mvdb_encoder_free(mvdb_encoder* enc);

/* ids[B,S], mask[B,S] (int32, host) -> out[B,H] pooled (cfg.pooling) + L2-normalised (host).
 * compute: 0 = exact-fp32 MFMA; 2 = split-precision GEMMs on the fp16 matrix cores (a.w ~ al.wh + ah.wl + ah.wh with
 * (h, l) the fp16 RNE split of an fp32 value — 22 significant bits —, weights scaled per tensor by a power of two,
 * fp32 accumulate; needs |activation| <= 65504; embeddings within 6e-7 of transformers' fp32 output like the exact
 * mode's; the two attention products run the same split, softmax / LayerNorm / pooling stay fp32) — what the Python
 * drop-in uses by default.  (1, the single-bf16-product mode of earlier builds, was removed: MVDB_ERR_ARG.)
 * Up to 64 token slots the forward is ONE layer-walking launch in exact fp32 whatever `compute` says
 * (mvdb_encoder_walks below). */
int mvdb_encoder_forward(mvdb_encoder* enc, const int32_t* ids_host, const int32_t* mask_host,
                         int B, int S, int compute, float* out_host);
int mvdb_encoder_forward_device(mvdb_encoder* enc, const int32_t* ids_dev, const int32_t* mask_dev,
                                int B, int S, int compute, float* out_dev, float* hidden_dev,
                                void* stream);

/* Device address of ONE uint32 that every forward clears when it starts and ORs 1 into when a pooled row of a non-empty sentence
 * is not finite — the split-precision mode (compute = 2) takes activations as fp16 pieces, so an activation beyond 65504
 * overflows.  mvdb_encoder_forward_device never synchronises: its callers (and a graph that chains the forward into a search)
 * test this word on the device, or read it back behind their own synchronisation, and re-run in the exact mode (compute = 0) —
 * which is what mvdb_encoder_forward's Python wrapper does on the host path.  Valid until mvdb_encoder_free. */
const unsigned int* mvdb_encoder_overflow_flag(const mvdb_encoder* enc);

/* 1 when a forward of B x S token slots is of the shape the ONE layer-walking launch serves (csrc/encoder_walk.hpp: at most 64
 * token slots — one sentence per call is the reference's only shape, embedding_model.py:62-71; beyond 64 slots the per-op kernels
 * are faster since round 6 —, exact fp32 matrix cores whatever `compute` says), 0 when it runs the per-op kernels.
 * MVDB_ENCODER_WALK=0 (read at mvdb_encoder_create) switches the launch off.  A forward of that shape still takes the per-op
 * kernels when the caller is CAPTURING the stream, while another process holds the GPU's walking gate, and for a while after a
 * launch was abandoned (next paragraph). */
int mvdb_encoder_walks(const mvdb_encoder* enc, int B, int S);

/* The walking launch is a persistent grid whose workgroups wait for each other, so it completes only when all of them are
 * resident.  It is guarded three ways (csrc/encoder.hip "Walking launches ..."): one such launch at a time per device inside
 * a process (whatever encoder, stream or host thread); an advisory flock on /dev/shm/mvdb_walk_<GPU UUID>.lock across
 * processes (MVDB_WALK_LOCK=0 switches it off, MVDB_WALK_LOCK_DIR moves it); and BOUNDED waits inside the kernel: no wait
 * outlasts MVDB_WALK_DEADLINE_US (default 20000, read when the encoder first walks) — the launch then abandons itself, fills
 * `out` with NaN, raises the overflow word above and counts itself.  mvdb_encoder_forward notices that behind its own stream
 * wait and re-runs the forward on the per-op kernels within the same call; a caller of mvdb_encoder_forward_device sees the
 * overflow word (or `aborts` here) behind ITS stream wait and calls again.  After an abandoned launch the next 256 forwards of
 * the encoder take the per-op kernels.  aborts = launches abandoned so far (completed ones), fallbacks = forwards
 * mvdb_encoder_forward re-ran, suspended_calls = forwards left before the next walking attempt; any pointer may be NULL. */
int mvdb_encoder_walk_stats(const mvdb_encoder* enc, unsigned long long* aborts, unsigned long long* fallbacks,
                            int* suspended_calls);

/* Which tile form of the split-precision GEMM a batch of `tokens` packed tokens selects for an N-wide product on a
 * device with `compute_units` CUs: 256 or 192 = the 256-row form on 256 x 256 / 256 x 192 tiles (one eight-wave workgroup
 * per CU; N % 256 == 0 resp. N % 192 == 0 and the tiles make whole rounds of the CUs: >= 1 round, and >= 4 rounds or a
 * last round >= 85 % full), 0 = the 64- / 128-row forms.  The same predicate runs on the device, on the packed token
 * count, inside the paired launches of a forward; exported so that the rule is testable without a GPU. */
int mvdb_encoder_gemm_tile_form(int64_t tokens, int n, int compute_units);

/* Planes a SMALL batch's N = H GEMM (attention output projection, FFN2; on the wide shapes also QKV and FFN1) is split into over
 * K (0: not split): a K-step of the GEMM is a latency step — one barrier and one DMA round trip — so while the 64 x 128 tiles of
 * a batch leave CUs idle, K is cut into min(8, (k / 32) / 4, compute_units / tiles) planes (snapped to 2 / 3 / 4 / 6 / 8), summed in
 * plane order by the LayerNorm / image kernel behind the GEMM.  The same rule on the host, from the padded token count; exported
 * so that it is testable without a GPU (tests/test_encoder_tiles.py).  MVDB_GEMM_X3_SPLITK* environment switches: csrc/encoder.hip. */
int mvdb_encoder_splitk_planes(int64_t tokens, int n, int k, int compute_units);

#ifdef __cplusplus
}
#endif
#endif /* MVDB_H */
