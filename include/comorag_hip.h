/*
 * comorag_hip.h — C-ABI of libcomorag_hip.so, the MI355X (gfx950) dense-retrieval engine
 * behind ComoRAG's EmbeddingModel / EmbeddingStore Python API.
 *
 * The reference (EternityJune25/ComoRAG @ 2025-08-29) has no FFI: its boundary for this path is
 * duck-typed Python.  Each entry point below names the reference code whose
 * numeric work it replaces (paths relative to src/comorag/).  The Python host side
 * (comorag_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - every function returns int32 status: 0 = CMR_OK, < 0 = error; cmr_last_error() returns a
 *    thread-local NUL-terminated message valid until the next call on that thread.
 *  - host buffers are caller-owned, C-contiguous, never retained after return.
 *  - "_dev" variants take DEVICE pointers (e.g. torch tensors' data_ptr()) and a hipStream_t
 *    passed as void* (NULL = the legacy default stream, which is what torch's default stream
 *    handle is); they enqueue work on THAT stream and return without synchronising, so the work
 *    is ordered after the caller's earlier kernels on the stream and before its later ones.
 *    Calls on one stream must not be issued concurrently from several threads.
 *    Non-finite queries are reported (CMR_ERR_NONFINITE) by the synchronous host-buffer calls;
 *    the _dev and pipelined calls cannot report them without a sync: their outputs are then
 *    unspecified and cmr_index_query_status() tells, after the fact, that it happened.
 *  - handles are opaque; destroy(NULL) is a no-op.  All functions are thread-safe: searches on
 *    one index run concurrently (shared lock), append/destroy are exclusive; concurrent
 *    cmr_graph_ppr / cmr_index_ppr calls on one graph each take their own scratch vectors.
 *  - there is NO CPU fallback: without a visible gfx950 device every compute call fails with
 *    CMR_ERR_NO_DEVICE.
 *
 * Result order (exported tie rule): score descending, then row index ascending.  (Deviation from the reference, by
 * necessity: ComoRAG ranks with np.argsort(scores)[::-1] (ComoRAG.py:964), numpy's default introsort — not stable, so rows with
 * EQUAL scores, i.e. duplicated chunks, come out in an order numpy does not specify and that changes with the array around them.
 * The Python layer runs that very line on the GPU scores for corpora below 28672 rows and takes this library's order above;
 * tests/test_dropin_gpu.py pins both sides of the threshold.)  Scores on the
 * wire are RAW inner products; ComoRAG's min-max normalisation (utils/misc_utils.py:141-150) is
 * applied by the Python layer from out_min/out_max so its formula stays textually the
 * reference's.  Inputs must be finite (checked: CMR_ERR_NONFINITE).
 *
 * Encoder stages (cmr_encoder_*): the library computes the model's own functions — softmax attention, fp32 LayerNorm statistics, masked
 * mean-pool + L2 normalisation; the FFN's GELU runs in PyTorch, by default in the model's exact (erf) form.  The Python layer can OPT IN (`embedding_gelu = "epilogue"`) to hipBLASLt's bias + GELU GEMM epilogue,
 * whose GELU is the TANH form (<= 4.8e-4 per activation from the erf form: a different function from BGEEmbedding.py:120's, inside the
 * 1e-3 bar on the tested weights, never the default).
 *
 * Synchronous host-buffer calls (cmr_index_search, cmr_index_search_min_score, cmr_index_scores, and cmr_mindex_* on top of them) return
 * as soon as the results are in the caller's buffers: where the search's last kernel can report it, the host polls a word that kernel
 * stores behind its results in a pinned, device-mapped buffer instead of waiting for the stream (option sync_poll, default 1; ~5 us per
 * call).  The stream itself may still be finishing that kernel's epilogue when the call returns — invisible to the caller: every later
 * call on the index is ordered behind it.
 */
#ifndef COMORAG_HIP_H
#define COMORAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMR_ABI_VERSION 2

enum cmr_status {
    CMR_OK = 0,
    CMR_ERR_INVALID = -1,     /* bad argument */
    CMR_ERR_NO_DEVICE = -2,   /* no usable HIP device / wrong arch */
    CMR_ERR_HIP = -3,         /* a HIP runtime call failed (message has the call) */
    CMR_ERR_OOM = -4,
    CMR_ERR_NONFINITE = -5,   /* NaN/Inf in rows or queries */
    CMR_ERR_UNSUPPORTED = -6  /* e.g. k above CMR_MAX_K */
};

enum cmr_dtype { CMR_F32 = 0, CMR_BF16 = 1, CMR_F16 = 2 };

/* index flags */
#define CMR_FLAG_KEEP_F32 1u /* also keep a row-major fp32 shadow (exact re-score, C5) */

#define CMR_MAX_K 128        /* largest k served by the fused scan+top-k kernel */
#define CMR_MAX_K_2PASS 4096 /* largest k overall: above CMR_MAX_K the scores of a query block are
                                materialised in HBM and selected per row (retrieve_knn's k = 2047) */

typedef struct cmr_index cmr_index_t;

/* ---- library --------------------------------------------------------------------------- */
int32_t cmr_abi_version(void);
const char* cmr_last_error(void);
/* number of visible HIP devices (0 on a CPU-only host; never an error there) */
int32_t cmr_device_count(int32_t* n);
/* name / arch / CU count / HBM bytes of a device — bench.py prints this next to the roofline */
int32_t cmr_device_info(int32_t device_id, char* name, int32_t name_len, int32_t* n_cu,
                        int64_t* hbm_bytes);

/* ---- index lifecycle -------------------------------------------------------------------
 * Replaces the dense matrices ComoRAG.prepare_retrieval_objects materialises on the host
 * (ComoRAG.py:896-900: np.array(store.get_embeddings(keys)) x3-4) and
 * EmbeddingStore.get_embeddings' per-call N x D copy (embedding_store.py:150-157).
 * Storage: HBM-resident, MFMA-fragment-major panels of 32 rows; dim is padded
 * to a multiple of 128 with zeros.  capacity grows x2 (device-to-device hipMemcpyAsync).      */
int32_t cmr_index_create(int32_t device_id, int32_t dim, int32_t dtype, int64_t capacity_hint,
                         uint32_t flags, cmr_index_t** out);
int32_t cmr_index_destroy(cmr_index_t* idx);
int32_t cmr_index_size(cmr_index_t* idx, int64_t* n_rows);
int32_t cmr_index_info(cmr_index_t* idx, int32_t* dim, int32_t* dtype, int64_t* capacity_rows,
                       int64_t* device_bytes);

/* Append n rows (fp32, row-major [n, dim]); converted round-to-nearest-even to the index dtype
 * on the device.  Row ids are assigned densely in append order (id = previous size + i) — the
 * same order EmbeddingStore._upsert extends its lists (embedding_store.py:122-128), so
 * hash_id_to_idx stays valid as the row id.  The memory-pool appends of the probe loop
 * (utils/memory_utils.py:176,297-300) use this too (BASELINE config 4).                        */
int32_t cmr_index_append(cmr_index_t* idx, const float* rows_f32, int64_t n);
int32_t cmr_index_append_dev(cmr_index_t* idx, const float* rows_f32_dev, int64_t n, void* stream);

/* ---- search ----------------------------------------------------------------------------
 * Fused scan + per-query top-k + global min/max of the raw scores.
 * Replaces np.dot + min_max_normalize + argsort in ComoRAG.dense_passage_retrieval
 * (ComoRAG.py:950-967), get_fact_scores + link_top_k argsort (ComoRAG.py:937-948,1073),
 * get_similar_summaries (utils/embed_utils.py:152-158), the python-loop cosine of
 * MemoryPool.retrieve_similar_nodes (utils/memory_utils.py:213-227) and each
 * torch.mm + torch.topk block of retrieve_knn (utils/embed_utils.py:52-78); k <= CMR_MAX_K_2PASS.
 *   q        [nq, dim] fp32 (rounded to the index dtype for bf16/f16 indexes, as BASELINE.md §2)
 *   out_ids  [nq, k] int64 row ids, -1 padded when the index has < k rows
 *   out_scores [nq, k] fp32 raw inner products, descending, -inf padded
 *   out_min/out_max [nq] fp32 over ALL rows (NULL allowed)                                     */
int32_t cmr_index_search(cmr_index_t* idx, const float* q_f32, int32_t nq, int32_t k,
                         int64_t* out_ids, float* out_scores, float* out_min, float* out_max);
/* Threshold search: the k best rows among those with raw score >= min_score (-1 / -inf padded) — the fused kernel's
 * running threshold simply starts at min_score, so nothing below it is ever pushed, merged or written.  This is what
 * the synonymy self-join consumes: ComoRAG.add_synonymy_edges (ComoRAG.py:670-712) asks retrieve_knn for 2047 neighbours
 * of every entity and then stops at the first score < synonymy_edge_sim_threshold (0.8) or after 101 accepted ones
 * (:696-699), i.e. it reads ~1/20 of what the reference materialises.  k <= CMR_MAX_K.                                  */
int32_t cmr_index_search_min_score(cmr_index_t* idx, const float* q_f32, int32_t nq, int32_t k, float min_score,
                                   int64_t* out_ids, float* out_scores);
/* The same on device buffers, enqueued on `stream` without synchronising: the self-join issues one call per query block
 * (queries uploaded once, results downloaded once) instead of an upload, a sync and a download per block.                 */
int32_t cmr_index_search_min_score_dev(cmr_index_t* idx, const float* q_f32_dev, int32_t nq, int32_t k, float min_score,
                                       int64_t* out_ids_dev, float* out_scores_dev, void* stream);
/* The threshold search in throughput mode (as cmr_index_search_pipelined below: the index's own streams, packing of block i + 1
 * and the candidate merge of block i - 1 run beside the scan of block i) — the synonymy self-join enqueues all its query blocks
 * this way and waits for the last *done_event (every merge runs on one in-order stream).                                  */
int32_t cmr_index_search_min_score_pipelined(cmr_index_t* idx, const float* q_f32_dev, int32_t nq, int32_t k, float min_score,
                                             int64_t* out_ids_dev, float* out_scores_dev, void* wait_event, void** done_event);
int32_t cmr_index_search_dev(cmr_index_t* idx, const float* q_f32_dev, int32_t nq, int32_t k,
                             int64_t* out_ids_dev, float* out_scores_dev, float* out_min_dev,
                             float* out_max_dev, void* stream);

/* Throughput mode for a stream of independent query batches (serving; bench.py).  Work is enqueued on streams owned by
 * the index: query packing + sampling passes of batch i+1 overlap the HBM-bound main scan of batch i.  On a 256-CU device
 * main scans shorter than ~1 ms (shards up to ~4 M x 768 bf16 rows) of batches of <= 64 queries run on a CU-masked pair of
 * streams (n_cu - 64 CUs; consecutive scans alternate between the two, so one scan's workgroups take over the CUs the previous
 * scan's workgroups leave) and their pre-phases on the other 64 CUs, wide batches likewise on n_cu - 32 / 32 CUs; longer scans
 * run on unmasked streams with a trimmed grid.  wait_event (hipEvent_t or NULL): inputs are ready when it
 * completes.  *done_event (hipEvent_t owned by the index): outputs are complete when it does; it is re-recorded three
 * pipelined calls later (the pipeline has three slots), so wait on it (hipStreamWaitEvent / hipEventSynchronize) before
 * then.  k <= CMR_MAX_K.  Results are identical to cmr_index_search_dev.                                               */
int32_t cmr_index_search_pipelined(cmr_index_t* idx, const float* q_f32_dev, int32_t nq, int32_t k,
                                   int64_t* out_ids_dev, float* out_scores_dev, float* out_min_dev,
                                   float* out_max_dev, void* wait_event, void** done_event);

/* Row-shard support: every row id returned by the search entry points is offset by `base` (the
 * global id of local row 0), so per-shard candidates can be all-gathered and merged as they are.  */
int32_t cmr_index_set_id_base(cmr_index_t* idx, int64_t base);
/* The same for a shard that takes INCREMENTAL appends (BASELINE config 4 at N > 1): global ids stay dense in append order
 * (embedding_store.py:122-128, utils/memory_utils.py:294-300), an append goes to the currently shortest shard, so a shard
 * holds several runs ("blocks") of consecutive global ids: local rows [local_start[b], local_start[b+1]) are global ids
 * global_start[b] + 0, 1, ...  (local_start[0] = 0; both arrays ascending; the last block is open-ended and grows with
 * cmr_index_append).  Every id the search entry points return is translated through the table (one extra tiny launch
 * when there is more than one block), cmr_index_rescore / cmr_index_get_rows translate the ids they are given.  Global
 * ids must stay < 2^32 - 1 (the packed candidate exchange carries 32-bit rows): checked here and by append.            */
int32_t cmr_index_set_id_blocks(cmr_index_t* idx, int32_t n_blocks, const int64_t* local_start, const int64_t* global_start);
/* Route selectors.  Each option picks between implementations that return the SAME results (the tests hold the routes
 * against each other bit for bit; tools A/B kernel decisions with them).  Nothing is read from the environment by the
 * shipped library.  Names: scan_ring (8 | 16), scan_asm_ring (0 | 1), scan_grid, scan_no_sample, scan_no_wide (batches of
 * more than 64 queries as narrow passes), scan_no_tiny / scan_no_small / small_max_panels / tiny_multi (single-launch
 * paths), zero_copy, sample_single, sample_single_max, sample_tau_in_scan, sample_div, sample_maxmul, scan_fin (0: small synchronous
 * batches run the sampling / scan / merge chain instead of the scan with the finishing stage) / scan_fin_queries (<= 16) /
 * scan_fin_dense / scan_fin_spin / scan_fin_suppliers / scan_fin_cap, sync_poll (0: synchronous calls wait for the stream), wide_mode (1: register-resident wide kernel | 2: query-split grid of the narrow kernel),
 * stream_nt, pipe_reserve_cus, pipe_slots (2..4), wide_waves (4 | 8:
 * waves per workgroup of the batch-256 kernel at 768-d; 8 only in builds with -DCMR_WIDE8), pipe_cu_mask (0: never | 1 | 2: every scan; default: scans shorter than ~1 ms) and
 * pipe_dual_scan (0 | 1; default: scans shorter than ~1 ms) — the pipelined search's streams with explicit CU masks (scans
 * of <= 64-query batches on n_cu - 64 CUs, their pre-phases on the other 64) and two alternating scan streams; both must
 * be set before the first pipelined call.
 * Unknown names: CMR_ERR_INVALID.
 * The wide-batch kernel exists for padded dims 768 (256 queries per pass) and 1024 (128 per pass) in bf16 / f16; any other
 * dim and every fp32 index run a batch of B > 64 queries on the query-split grid of the narrow kernel (up to four query tiles
 * per corpus pass) — same results.                                                                                           */
int32_t cmr_index_set_option(cmr_index_t* idx, const char* name, int64_t value);
/* What the pipeline actually does (read-only): "pipe_dual_scan_active" / "pipe_dual_scan_wide_active" (the last pipelined <= 64-query / wide pass alternated between
 * the two scan streams: by default only scans shorter than ~1 ms do — launches that overlap have no per-launch duration, so a
 * caller that times kernels must know), "pipe_cu_mask_active", "pipe_scan_cus".                                            */
int32_t cmr_index_get_option(cmr_index_t* idx, const char* name, int64_t* value);
/* The pipeline's streams (which = 0 pre-phase and 1 first main-scan stream of the <= 64-query batches, 2 candidate merges /
 * outputs of every batch) as hipStream_t.  Work enqueued on stream 2 after a pipelined call is ordered after that call's
 * outputs and before the next use of the same output buffers (the RCCL exchange goes there).     */
int32_t cmr_index_pipeline_stream(cmr_index_t* idx, int32_t which, void** stream);

/* Did any _dev / pipelined search since the last call see a NaN/Inf query?  Synchronises the index's pipeline streams
 * and the streams used with the _dev API, reads and re-arms their flags.  *nonfinite = 0 / 1.                        */
int32_t cmr_index_query_status(cmr_index_t* idx, int32_t* nonfinite);

/* Helpers for callers that only hold opaque handles (e.g. a torch stream's cuda_stream):
 * make `stream` (hipStream_t) wait for `event` (hipEvent_t from *done_event), or block the host.  */
int32_t cmr_stream_wait_event(void* stream, void* event);
int32_t cmr_event_synchronize(void* event);

/* All N raw scores per query, out [nq, ld] fp32 (ld >= N; ld = N when 0).  For the callers that
 * consume every score: graph_search_with_fact_entities reads all (id, score) pairs of
 * dense_passage_retrieval (ComoRAG.py:1034-1042) and get_fact_scores returns the full vector
 * (ComoRAG.py:937-948).                                                                        */
int32_t cmr_index_scores(cmr_index_t* idx, const float* q_f32, int32_t nq, float* out, int64_t ld);
int32_t cmr_index_scores_dev(cmr_index_t* idx, const float* q_f32_dev, int32_t nq, float* out_dev,
                             int64_t ld, void* stream);

/* ALL N rows of each query by descending raw score (ties: ascending row id), out_ids / out_scores
 * [nq, N]; out_min / out_max [nq] (NULL allowed) are the last / first sorted score.  Replaces the
 * np.dot + full argsort of ComoRAG.dense_passage_retrieval (ComoRAG.py:958-966) for callers that
 * need the complete ranking: scan on MFMA, then a stable device radix sort.                       */
int32_t cmr_index_sorted_scores(cmr_index_t* idx, const float* q_f32, int32_t nq, int64_t* out_ids,
                                float* out_scores, float* out_min, float* out_max);

/* Exact fp32 re-score of candidate rows (BASELINE config 5 "cross-scores on top-100").  The
 * reference's rerank.py is an LLM filter with no numeric scoring (rerank.py:100-123); this is
 * the numeric stage the engine adds behind the same (indices, items, {'confidence'}) shape.
 * Uses the fp32 shadow when the index was created with CMR_FLAG_KEEP_F32, otherwise the stored
 * (rounded) rows with fp32 queries.  cand [nq, n_cand] int64 row ids as the search entry points
 * return them (i.e. including the cmr_index_set_id_base offset; -1 = skip); out_ids likewise.   */
int32_t cmr_index_rescore(cmr_index_t* idx, const float* q_f32, int32_t nq, const int64_t* cand,
                          int32_t n_cand, int32_t k, int64_t* out_ids, float* out_scores);

/* Gather rows back to the host as fp32 (dequantised), out [n, dim]; ids as returned by search (with the id base). */
int32_t cmr_index_get_rows(cmr_index_t* idx, const int64_t* ids, int64_t n, float* out);

/* ---- multi-shard merge ------------------------------------------------------------------
 * Final merge of per-shard candidates after the RCCL all-gather (no reference counterpart;
 * SURVEY.md §8e).  ids/scores [n_shards, nq, k] (ids already global, -1 = empty); same tie rule,
 * so the result equals a single-shard search.  Host version and device version.               */
int32_t cmr_merge_topk(const int64_t* ids, const float* scores, int32_t n_shards, int32_t nq,
                       int32_t k, int64_t* out_ids, float* out_scores);
int32_t cmr_merge_topk_dev(int32_t device_id, const int64_t* ids_dev, const float* scores_dev,
                           int32_t n_shards, int32_t nq, int32_t k, int64_t* out_ids_dev,
                           float* out_scores_dev, void* stream);

/* ---- graph stage: DPR-seeded personalised PageRank --------------------------------------
 * Replaces the tail of ComoRAG.graph_search_with_fact_entities (ComoRAG.py:1034-1044: every (passage id, normalised
 * DPR score) pair copied to the host and scattered into `passage_weights`) and ComoRAG.run_ppr (:1086-1105: igraph /
 * prpack personalised PageRank — undirected, edge attribute 'weight', damping 0.5 — then pagerank[passage_node_idxs]).
 *   cmr_graph_create           undirected weighted edge list (igraph's get_edgelist() + es['weight']; weight NULL = 1)
 *                              -> symmetric CSR in HBM.  A vertex without edges jumps according to the reset vector.
 *   cmr_graph_set_passage_vertices   vertex of every passage ROW of the index (ComoRAG's passage_node_idxs)
 *   cmr_graph_ppr              run_ppr alone: reset [n_vertices] (negative / NaN -> 0, ComoRAG.py:1090) -> all scores
 *   cmr_index_ppr              the fused path for one query: scan -> min_max_normalize(scores) * passage_node_weight
 *                              scattered into the reset vector on the device, + the (few) phrase seeds (duplicate seed
 *                              vertices are SUMMED; ComoRAG's own loop assigns, last wins, :1019-1021 — resolve before the call) -> PPR ->
 *                              out_doc_scores [n_rows] = pagerank[vertex of row]; 8 * n_rows bytes come back instead of
 *                              the 12 * N of the full ranking.  The caller sorts (np.argsort(doc_scores)[::-1], :1102).
 * Power iteration in fp64, ceil(log(tol/2)/log(damping)) steps (<= max_iter; *iters = steps taken), fixed summation
 * order.  prpack solves the same linear system directly to ~1e-10.                                                     */
typedef struct cmr_graph cmr_graph_t;
int32_t cmr_graph_create(int32_t device_id, int64_t n_vertices, int64_t n_edges, const int32_t* src, const int32_t* dst,
                         const double* weight, cmr_graph_t** out);
int32_t cmr_graph_destroy(cmr_graph_t* g);
int32_t cmr_graph_set_passage_vertices(cmr_graph_t* g, const int32_t* vertex_of_row, int64_t n_rows);
int32_t cmr_graph_ppr(cmr_graph_t* g, const double* reset, double damping, double tol, int32_t max_iter, double* out_scores,
                      int32_t* iters);
int32_t cmr_index_ppr(cmr_index_t* idx, cmr_graph_t* g, const float* q_f32, const int32_t* seed_vertices,
                      const double* seed_weights, int32_t n_seeds, double passage_node_weight, double damping, double tol,
                      int32_t max_iter, double* out_doc_scores, int32_t* iters);

/* ---- row-shard exchange ------------------------------------------------------------------
 * One process per GPU, each with a row shard (cmr_index_set_id_base makes its searches return global ids).  Per query
 * batch every rank packs its [nq, k] candidates into ONE u64 each — the order-preserving score code in the high word,
 * ~(global row) in the low word, so unsigned order == the exported order; global rows must be < 2^32 - 1 — all-gathers
 * them with a single collective (nq*k*8 bytes per rank; latency-bound over xGMI) and merges world*k keys per query.
 * The result equals a single-shard search by construction.  (No reference counterpart)
 *
 * cmr_pack_candidates_dev / cmr_merge_keys_dev are the two kernels on their own, for hosts that bring their own
 * collective (comorag_amd/sharded.py runs torch.distributed's all_gather_into_tensor between them).
 * cmr_comm_* wrap RCCL itself (bound with dlopen at first use — the RCCL already in the process when there is one):
 * rank 0 calls cmr_comm_unique_id and ships the 128 bytes to the other ranks by any channel, every rank calls
 * cmr_comm_create (collective), then cmr_comm_allgather_merge per batch on a stream of its choice — the index
 * pipeline's post stream (cmr_index_pipeline_stream(idx, 2)) orders it after the batch's outputs.                     */
#define CMR_COMM_ID_BYTES 128
typedef struct cmr_comm cmr_comm_t;
int32_t cmr_pack_candidates_dev(const int64_t* ids_dev, const float* scores_dev, int64_t n, uint64_t* keys_dev, void* stream);
int32_t cmr_merge_keys_dev(const uint64_t* keys_dev /*[n_shards][nq][k]*/, int32_t n_shards, int32_t nq, int32_t k,
                           int64_t* out_ids_dev, float* out_scores_dev, void* stream);
int32_t cmr_comm_unique_id(uint8_t* out_id128);
int32_t cmr_comm_create(int32_t world, int32_t rank, const uint8_t* id128, int32_t device_id, cmr_comm_t** out);
int32_t cmr_comm_destroy(cmr_comm_t* comm);
/* what the communicator was created with, and the rank count RCCL itself reports (ncclCommCount; 0 if unavailable) —
 * bench.py prints it so a multi-GPU line shows that the collective really spanned N ranks                              */
int32_t cmr_comm_info(cmr_comm_t* comm, int32_t* world, int32_t* rank, int32_t* rccl_ranks);
int32_t cmr_comm_allgather_merge(cmr_comm_t* comm, const int64_t* ids_dev, const float* scores_dev, int32_t nq, int32_t k,
                                 int64_t* out_ids_dev, float* out_scores_dev, void* stream);

/* ---- one process, several devices --------------------------------------------------------------
 * ComoRAG runs ONE process whose 16-thread pool (ComoRAG.try_answer, ComoRAG.py:432-453) calls tri_retrieve (:456-554) at
 * LLM-dependent times; a row-sharded index it can sit on must therefore be driven from one process.  cmr_mindex_t owns
 * n_shards cmr_index_t row shards, shard s on device device_ids[s] (a device may appear several times: logical shards — how a
 * 1-GPU box rehearses the layout).  Same conventions as cmr_index_t: thread-safe (searches concurrent, append exclusive), host
 * buffers in / out, exported tie rule, global row ids dense in append order (embedding_store.py:122-128,
 * utils/memory_utils.py:294-300) — results are identical to ONE cmr_index_t holding the same rows.
 *   append      rows go to the shards in blocks: a block opens on the currently shortest shard and holds
 *               max(append_block_rows, ceil(n / n_shards)) rows — a bulk append lands as contiguous row blocks, a memory pool's
 *               25-row appends fill blocks of append_block_rows (default 65536: a corpus of a few thousand rows lives on ONE shard
 *               and is searched without multi-shard overhead) that go round the shards; each shard translates
 *               its rows through its block table (cmr_index_set_id_blocks).  All or nothing: a chunk that fails (NaN/Inf rows,
 *               out of memory) rolls the shards that already took theirs back.
 *   search      begins on every non-empty shard before it finishes on any (from three shards on the per-shard enqueues are
 *               issued by per-shard worker threads); every shard's merge kernel writes its [nq, k] candidates straight into
 *               pinned, device-mapped HOST memory (nq * k * 12 bytes per shard over PCIe: no peer access, no collective), the
 *               host does the final merge of the sorted lists — north_star's "host-side final merge"; min / max combine.
 *   scores / sorted_scores / rescore / get_rows   as on cmr_index_t, over global ids.
 *   search_pipelined + collect   throughput mode: q_dev[s] = the batch's queries on shard s's device (entries of empty shards are
 *               ignored); returns at once with a ticket, up to four tickets may be uncollected; collect waits for the shards,
 *               merges on the host and frees the ticket.  k <= CMR_MAX_K.
 *   set_option  "append_block_rows", "parallel_min_shards" (default 3), "force_peer_staging" (1: append_dev stages every chunk through
 *               hipMemcpyPeer even when source and shard share a device — how a one-GPU box executes the cross-device append path);
 *               anything else goes to every shard (cmr_index_set_option) with NO search in flight (exclusive over the handle).
 *   shard       borrow shard s (profiling, reading options, tests).  Do NOT append to it or re-base it directly, and set route
 *               selectors through cmr_mindex_set_option, not on the borrowed handle, while other threads search.
 * cmr_mindex_plan_append is the routing rule on its own (pure host arithmetic, no device): chunks (shard, rows) for m appended
 * rows given the shards' sizes and the open block (cur_shard, cur_room); n_chunks always returns the chunk count.            */
typedef struct cmr_mindex cmr_mindex_t;
int32_t cmr_mindex_create(int32_t n_shards, const int32_t* device_ids, int32_t dim, int32_t dtype, int64_t capacity_hint,
                          uint32_t flags, cmr_mindex_t** out);
int32_t cmr_mindex_destroy(cmr_mindex_t* m);
int32_t cmr_mindex_size(cmr_mindex_t* m, int64_t* n_rows);
int32_t cmr_mindex_info(cmr_mindex_t* m, int32_t* n_shards, int32_t* device_ids /*[n_shards] or NULL*/,
                        int64_t* shard_rows /*[n_shards] or NULL*/, int64_t* device_bytes);
int32_t cmr_mindex_shard(cmr_mindex_t* m, int32_t s, cmr_index_t** out);
int32_t cmr_mindex_set_option(cmr_mindex_t* m, const char* name, int64_t value);
int32_t cmr_mindex_append(cmr_mindex_t* m, const float* rows_f32, int64_t n);
/* rows on device src_device (e.g. an encoder's output tensor; `stream` = the stream that produced them): chunks of shards on
 * that device are appended in place, the others go device to device (hipMemcpyPeer) first.  Returns when the rows are in.  */
int32_t cmr_mindex_append_dev(cmr_mindex_t* m, const float* rows_f32_dev, int64_t n, int32_t src_device, void* stream);
int32_t cmr_mindex_search(cmr_mindex_t* m, const float* q_f32, int32_t nq, int32_t k, int64_t* out_ids, float* out_scores,
                          float* out_min, float* out_max);
int32_t cmr_mindex_search_min_score(cmr_mindex_t* m, const float* q_f32, int32_t nq, int32_t k, float min_score,
                                    int64_t* out_ids, float* out_scores);
int32_t cmr_mindex_scores(cmr_mindex_t* m, const float* q_f32, int32_t nq, float* out, int64_t ld);
int32_t cmr_mindex_sorted_scores(cmr_mindex_t* m, const float* q_f32, int32_t nq, int64_t* out_ids, float* out_scores,
                                 float* out_min, float* out_max);
int32_t cmr_mindex_rescore(cmr_mindex_t* m, const float* q_f32, int32_t nq, const int64_t* cand, int32_t n_cand, int32_t k,
                           int64_t* out_ids, float* out_scores);
int32_t cmr_mindex_get_rows(cmr_mindex_t* m, const int64_t* ids, int64_t n, float* out);
int32_t cmr_mindex_search_pipelined(cmr_mindex_t* m, const float* const* q_dev /*[n_shards]*/, int32_t nq, int32_t k, void** ticket);
int32_t cmr_mindex_collect(cmr_mindex_t* m, void* ticket, int64_t* out_ids, float* out_scores, float* out_min, float* out_max);
/* host-side cost of the throughput mode since the last reset: collected batches, mean microseconds a shard's enqueue job took
 * on its worker thread, mean microseconds collect waited for the shards' events, mean microseconds of the host merge          */
int32_t cmr_mindex_profile(cmr_mindex_t* m, int32_t reset, int64_t* n_batches, double* enqueue_us_per_shard, double* wait_us_per_batch,
                           double* merge_us_per_batch);
int32_t cmr_mindex_plan_append(const int64_t* shard_rows, int32_t n_shards, int32_t cur_shard, int64_t cur_room, int64_t m,
                               int64_t block_rows, int32_t max_chunks, int32_t* out_shard, int64_t* out_count, int32_t* n_chunks,
                               int32_t* new_cur, int64_t* new_room);

/* ---- encoder tail -----------------------------------------------------------------------
 * Fused masked mean-pool + L2-normalise of the encoder's last hidden state; replaces
 * mean_pooling (embedding_model/BGEEmbedding.py:15-28) + F.normalize (:126-127, eps 1e-12).
 *   hidden_dev [b, l, d] of hidden_dtype (cmr_dtype), mask_dev [b, l] int64 (HF attention_mask),
 *   out_dev [b, d] fp32.  normalize = 0 gives the plain masked mean.                           */
int32_t cmr_pool_l2norm(int32_t device_id, const void* hidden_dev, int32_t hidden_dtype,
                        const int64_t* mask_dev, int32_t b, int32_t l, int32_t d, int32_t normalize,
                        float* out_dev, void* stream);

/* ---- encoder layer pieces -----------------------------------------------------------------
 * The two non-GEMM stages of a BERT layer inside `self.embedding_model(**inputs)` (embedding_model/BGEEmbedding.py:119; the
 * GEMMs stay PyTorch-ROCm / hipBLASLt as north_star prescribes).  Both take device pointers of 16-bit tensors (dtype =
 * CMR_BF16 or CMR_F16) and run on `stream`.
 *
 * cmr_encoder_attention: out[b*l, hidden] = softmax(Q K^T / sqrt(head_dim), keys >= lens[s] masked) V per sequence and head,
 *   with Q | K | V the three hidden-wide column groups of ONE packed projection qkv_dev[b*l, 3*hidden] (hidden = n_heads *
 *   head_dim, head h = columns h*head_dim.. of its group); lens_dev[b] int32 = real tokens of each right-padded sequence
 *   (transformers' BertSelfAttention + its additive mask).  head_dim must be 64.  Rows >= lens[s] of out are unspecified
 *   finite values (zeros where a whole 128-row block is padding): the pooling mask drops them.
 * cmr_encoder_embed_layernorm: out[t] = LayerNorm(word[ids[t]] + position[t mod l] + token_type[tt[t]]) for the rows = b*l tokens
 *   of a [b, l] mini-batch (transformers' BertEmbeddings with default position ids; token_type_dev NULL = type 0; ids outside a
 *   table are clamped into it).  position_offset: 0 for BERT; padding_idx + 1 for the RoBERTa family (XLM-R = bge-m3), whose
 *   position table starts there (RobertaEmbeddings.create_position_ids_from_input_ids on right-padded rows: token t of a sequence
 *   sits at position t + padding_idx + 1; the padded tail's embeddings are never used).
 * cmr_encoder_add_layernorm: out = LayerNorm(y + bias + residual) * gamma + beta over rows of d elements, fp32 statistics
 *   (BertSelfOutput / BertOutput after their dense GEMM); bias_dev / residual_dev may be NULL.                              */
/* The LAST layer's cmr_encoder_add_layernorm with the encoder tail folded in — mean_pooling over the tokens < lens[s]
 * (embedding_model/BGEEmbedding.py:15-28) and F.normalize (:126-127, eps 1e-12; normalize = 0: the plain masked mean) of the
 * [b, l, d] mini-batch it would have written: out_dev [b, d] fp32; the hidden state itself is never stored (a 16-token block per
 * workgroup -> one fp32 partial row, summed in block order by a small second launch; every value is rounded to 16 bits before it
 * is added, as the stored hidden state would have been).  l % 16 == 0, d % 8 == 0, d <= 2048, 16-byte aligned buffers, right-padded
 * sequences (lens_dev [b] int32): CMR_ERR_UNSUPPORTED otherwise — callers then run add_layernorm + cmr_pool_l2norm.             */
int32_t cmr_encoder_add_layernorm_pool(int32_t device_id, const void* y_dev, const void* bias_dev, const void* residual_dev,
                                       const void* gamma_dev, const void* beta_dev, float eps, int32_t b, int32_t l, int32_t d,
                                       int32_t dtype, const int32_t* lens_dev, int32_t normalize, float* partial_dev /* scratch:
                                       b * (l / 16) * d floats, caller-owned so that a stream capture allocates nothing */,
                                       float* out_dev, void* stream);
int32_t cmr_encoder_embed_layernorm(int32_t device_id, const int64_t* ids_dev, const int64_t* token_type_dev,
                                    const void* word_dev, const void* pos_dev, const void* type_dev,
                                    const void* gamma_dev, const void* beta_dev, float eps, int64_t rows, int32_t l,
                                    int32_t d, int32_t vocab, int32_t n_positions, int32_t n_types, int32_t position_offset,
                                    int32_t dtype, void* out_dev, void* stream);
/* cmr_encoder_embed_layernorm from RAGGED token ids: ids32_dev holds the b sequences' tokens back to back (int32), offsets_dev[b + 1]
 * their starts; row (s, t) of the [b, l] mini-batch embeds token t of sequence s (token 0 behind its end: padding rows, masked
 * downstream), token type 0.  What the host ships per mini-batch is then ONE int32 array lens | offsets | ids — no padded id, mask
 * or token-type tensors (the tokenizer call of embedding_model/BGEEmbedding.py:112-118 pads on the host and uploads three).     */
int32_t cmr_encoder_embed_layernorm_ragged(int32_t device_id, const int32_t* ids32_dev, const int32_t* offsets_dev, const void* word_dev,
                                           const void* pos_dev, const void* type_dev, const void* gamma_dev, const void* beta_dev, float eps,
                                           int32_t b, int32_t l, int32_t d, int32_t vocab, int32_t n_positions, int32_t position_offset,
                                           int32_t dtype, void* out_dev, void* stream);
int32_t cmr_encoder_attention(int32_t device_id, const void* qkv_dev, int32_t dtype, const int32_t* lens_dev,
                              int32_t b, int32_t l, int32_t n_heads, int32_t head_dim, void* out_dev, void* stream);
int32_t cmr_encoder_add_layernorm(int32_t device_id, const void* y_dev, const void* bias_dev,
                                  const void* residual_dev, const void* gamma_dev, const void* beta_dev,
                                  float eps, int64_t rows, int32_t d, int32_t dtype, void* out_dev, void* stream);

/* ---- measurement ------------------------------------------------------------------------
 * HIP-event timing of the dominant kernel (the corpus scan) on the stream it is launched on.
 * enable → run searches → collect returns launches, summed kernel ms and the algorithmic bytes
 * one launch reads (N_pad*Dpad*sizeof(elem) + query/result bytes), then resets.
 * on = 1: every main scan is timed; on = N > 1: every N-th one (the two event packets around a
 * scan cost ~25 us on the scan stream, 9 % of a 1 M-row step: sampling keeps the measurement from
 * slowing what it measures); on = 0: off.                                                      */
int32_t cmr_profile_enable(cmr_index_t* idx, int32_t on);
int32_t cmr_profile_collect(cmr_index_t* idx, int64_t* n_launches, double* total_ms,
                            double* bytes_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* COMORAG_HIP_H */
