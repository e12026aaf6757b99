// Host-callable launchers of the gfx950 kernels (defined in scan_kernels.hip / aux_kernels.hip).
// All launchers enqueue on `stream` and return the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;

struct CmrScanGeom {
    int dtype;      // CMR_DT_*
    int dpad;       // padded dim (multiple of 128)
    int ks;         // 1-KiB blocks per panel  (dpad/16 for 16-bit, dpad/8 for fp32)
    int nqt;        // query tiles of 32 per pass (1 or 2)
    int cap;        // per-(wave,query) candidate list capacity (128 or 256)
    int ring;       // register ring depth (8 or 16), ks % ring == 0
    int grid;       // workgroups
    int asm_ring;   // 1: hand-counted inline-asm load ring, 0: compiler-counted loads
    size_t lds;     // dynamic LDS bytes
    int wide_waves; // wide kernel at 768-d: 0 = default, 4 = one wave per SIMD x 2 tiles, 8 = two waves per SIMD x 1 tile
    int wide_abl;   // development builds only (-DCMR_DEV_KNOBS): ablation variant of the wide kernel
    int stream_default_policy;   // narrow top-k kernel: 1 = corpus loads with the default cache policy instead of non-temporal (query-split grid)
};

// max query tiles (1/2/0=unsupported) whose fragments fit LDS for this dtype/dpad
int cmr_scan_max_nqt(int dtype, int dpad);
// fills ks/lds for (dtype,dpad,nqt,cap); returns false if it does not fit
bool cmr_scan_geom(CmrScanGeom* g);

struct CmrScanArgs {
    const void* corpus;   // panel-major blocks
    const void* qfrag;    // [nqt][ks][64] uint4
    long long nrows;
    int npanels;
    int k;
    // top-k mode
    u64* lists;           // [W][nqt*32][cap]
    int* cnt;             // [W][nqt*32]
    float2* mm;           // [W][nqt*32]  (min,max)
    // scores mode
    float* scores;        // [nq][ld]
    long long ld;
    int nq;
    // sampling pass (top-k mode): sample_waves panels in chunks of 2^sample_chunk_log2 consecutive panels, chunk starts
    // sample_stride panels apart: sampled panel s is panel (s >> c) * sample_stride + (s & (2^c - 1)).  Narrow kernel:
    // wave w < sample_waves scans sampled panel w only; wide kernel: the grid's workgroups split the sampled panels.
    int sample_waves;
    int sample_stride;
    int sample_chunk_log2;
    const u64* tau_init;  // [nqt*32] initial threshold keys or nullptr
    // narrow kernel, <= 8 queries, k <= 64: the main pass derives its thresholds ITSELF from the lists of a preceding sampling
    // pass (sample_W lists of the same [W][32][cap] layout) instead of a merge launch between the two scans
    const u64* sample_lists;
    const int* sample_cnt;
    int sample_W;
    // narrow kernel, query-split grid: qgroups x (geom.grid) workgroups, group g scans the whole corpus for queries
    // g*nqt*32 ..; qfrag holds qgroups*nqt tiles, lists / cnt / mm are [qgroups][W][nqt*32](..), tau_init [qgroups][nqt*32]
    int qgroups;
    // narrow kernel with the finishing stage (cmr_launch_scan_fin): control words (CMR_FIN_CTL ints, zeroed once — the kernel re-arms
    // them), first-panel maxima [32][fin_ns], thresholds [32], dense candidate lists [32][fin_dcap], and the search's outputs
    int* fin;
    u64* fin_pmax;
    u64* fin_tau;
    u64* fin_dense;
    u64* fin_mm;          // [32][grid] scratch
    int fin_wgs;          // workgroups (the first to be through their first panels) whose waves' first panels supply the thresholds: 8 x fin_wgs <= CMR_FIN_SLOTS maxima
    int fin_first;        // workgroups of the first round (resident from the start: they cannot find thresholds when they begin); >= fin_wgs
    int fin_mul;          // workgroup b scans the panel ranges of virtual workgroup (b x fin_mul) mod grid (coprime with the grid; 1 = identity)
    int fin_dcap;
    int fin_spin;         // rounds (~1.5 us each) the other workgroups give the suppliers before they start without thresholds
    int64_t* out_ids;
    float* out_scores;
    float* out_min;
    float* out_max;
    long long id_base;
    // a word in pinned, device-mapped host memory (or nullptr) that the launch's LAST store sets once the results are written: 1 = final,
    // 2 = a list overflowed (the merge launch is still due).  The synchronous host API polls it instead of waiting for the end-of-kernel
    // signal (5.5 us per call on MI355X: tools/probe/poll_probe.hip) and launches the merge only in state 2.
    int* fin_done;
};
// control words (ints) — every counter on a 128-byte line of its own: device atomics on one line serialise
#define CMR_FIN_DONE 0        // workgroups whose waves have all published their first-panel maxima
#define CMR_FIN_READY 32      // bit q: the threshold of query q is published; bit 31: every first-panel maximum is
#define CMR_FIN_WGS 64        // workgroups finished
#define CMR_FIN_STATE 96      // result state: 1 = the scan wrote the final results itself, 2 = a list overflowed (the merge launch decides)
#define CMR_FIN_OVER 128      // some workgroup's staging area overflowed
#define CMR_FIN_DBG 136
#define CMR_FIN_DBG2 196      // development builds: per-wave sums (behind CMR_FIN_CLAIM's word, same line)
#define CMR_FIN_PUB 160       // supplying workgroups whose maxima are written through
#define CMR_FIN_CLAIM 192     // queries whose threshold somebody has taken on
#define CMR_FIN_MAX_QUERIES 16
#define CMR_FIN_DCNT(q) (224 + 32 * (q))     // dense list lengths
#define CMR_FIN_CTL (224 + 32 * 32)
#define CMR_FIN_LDS 4096      // bytes of LDS the finishing stage adds to the geometry's (the waves' min / max and first-panel maxima)
#define CMR_FIN_SLOTS 1024    // first-panel maxima the thresholds are taken from (one selection chunk of a wave)

hipError_t cmr_launch_scan_topk(const CmrScanGeom& g, const CmrScanArgs& a, hipStream_t s);
hipError_t cmr_launch_scan_scores(const CmrScanGeom& g, const CmrScanArgs& a, hipStream_t s);
// top-k scan of <= 32 queries (k <= 64) that also derives its thresholds (no sampling launch) and selects the final k best per
// query (no merge launch); a.fin[CMR_FIN_STATE] tells the merge launch behind it whether anything is left to do
hipError_t cmr_launch_scan_fin(const CmrScanGeom& g, const CmrScanArgs& a, hipStream_t s);
// wide-batch (register-resident queries) top-k scan: 4 waves x 2 (768-d) or 1 (1024-d) tiles of 32 queries;
// in sampling mode (sample_waves > 0) the grid's workgroups split the sample_waves strided panels among them
hipError_t cmr_launch_scan_wide(const CmrScanGeom& g, const CmrScanArgs& a, hipStream_t s);
int cmr_wide_queries(int dtype, int dpad);      // queries per pass of the wide kernel (0 = unavailable)
size_t cmr_wide_lds_bytes(int ks, int cap, int waves);

// queries fp32 [nq, dim] (device) -> fragment-ordered blocks of the index dtype, zero padded
hipError_t cmr_launch_prep_queries(int dtype, const float* q, int nq, int dim, int dpad, int nqt,
                                   void* qfrag, int* nonfinite_flag, hipStream_t s);
// threshold search: n initial threshold keys that admit exactly the scores >= min_score
hipError_t cmr_launch_fill_threshold(float min_score, int n, u64* tau, hipStream_t s);
// rows fp32 [n, dim] (device) -> panel-major blocks at row offset row0 (+ optional fp32 shadow)
hipError_t cmr_launch_convert_rows(int dtype, const float* rows, long long n, int dim, int dpad,
                                   long long row0, void* corpus, float* shadow, int* nonfinite_flag,
                                   hipStream_t s);
// per-wave candidate lists -> per-query top-k (ids/scores + min/max), or the sampling threshold
// grouped = true: the lists come from a query-split scan ([group][W][nq_stride]): query q reads group q / nq_stride
hipError_t cmr_launch_merge_query(const u64* lists, const int* cnt, int W, int nq_stride, int cap, int nq, int k,
                                  const float2* mm, long long id_base, int64_t* out_ids, float* out_scores,
                                  float* out_min, float* out_max, u64* out_tau, hipStream_t s, bool grouped = false,
                                  const int* skip_if_one = nullptr);      // device word: 1 = the results are final already (cmr_launch_scan_fin), return at once
// whole search of a small corpus in one launch (nq <= 16): kind 1 = <= 32 panels, kind 2 = hierarchical (up to max_panels panels,
// k <= 64; needs the arrival counter), 0 = not applicable.  `arrive`: a zeroed device int the launches of
// one stream share (nullptr: single-workgroup flat path only).
int cmr_tiny_kind(int nq, int npanels, int k, int multi, int max_panels);
size_t cmr_tiny_scratch_bytes(int nq, int npanels, int k, int multi, int max_panels);
hipError_t cmr_launch_tiny_search(int dtype, const void* corpus, const float* q, int nq, int dim, int dpad, long long nrows, int k, long long id_base,
                                  void* scratch, int64_t* out_ids, float* out_scores, float* out_min, float* out_max, int* flag, int* arrive,
                                  int max_panels, hipStream_t s, int* done = nullptr);      // done: as CmrScanArgs::fin_done (set to 1 behind the results)
hipError_t cmr_launch_tiny_scores(int dtype, const void* corpus, const float* q, int nq, int dim, int dpad, long long nrows, void* scratch,
                                  float* out, long long ld_out, int* flag, hipStream_t s, int* arrive = nullptr, int* done = nullptr);      // arrive (zeroed device int) + done: the last workgroup sets *done = 1 behind everybody's rows
// per-row top-k (k <= 4096) of a materialised score matrix [nq, ld]
hipError_t cmr_launch_topk_rows(const float* scores, long long ld, int n, int nq, int k, long long id_base,
                                int64_t* out_ids, float* out_scores, float* out_min, float* out_max, hipStream_t s);
// full descending sort of one query's scores (stable LSD radix sort; ties by ascending row)
size_t cmr_sort_workspace_bytes(long long n);
hipError_t cmr_launch_sort_scores(const float* scores, long long n, long long id_base, void* workspace, int64_t* out_ids,
                                  float* out_scores, hipStream_t s);
// shard merge: ids/scores [S][nq][k] -> [nq][k]
hipError_t cmr_launch_merge_shards(const int64_t* ids, const float* scores, int S, int nq, int k,
                                   int64_t* out_ids, float* out_scores, hipStream_t s);
// exact fp32 dot of queries with candidate rows, then top-k
// (candidate and output ids are global: local row + id_base)
hipError_t cmr_launch_rescore(int dtype, const void* corpus, const float* shadow, int dim, int dpad,
                              long long nrows, long long id_base, const float* q, int nq, const int64_t* cand, int n_cand,
                              int k, int64_t* out_ids, float* out_scores, hipStream_t s);
// gather rows as fp32
hipError_t cmr_launch_gather_rows(int dtype, const void* corpus, int dim, int dpad, long long nrows, long long id_base,
                                  const int64_t* ids, long long n, float* out, hipStream_t s);
// ids[i] (shard-local row, < 0 = empty) -> global id through the block table [local0[nb] | global0[nb]] (device)
hipError_t cmr_launch_remap_ids(int64_t* ids, long long n, const long long* tab, int nb, hipStream_t s);
// masked mean-pool + L2 norm
hipError_t cmr_launch_pool(const void* hidden, int hidden_dtype, const int64_t* mask, int b, int l, int d,
                           int normalize, float* partial, float* out, int splits, hipStream_t s);
int cmr_pool_splits(int b, int l, int d);
// encoder layer pieces (encoder_kernels.hip): masked attention over a packed [b*L, 3*hidden] projection (head width 64, 16-bit),
// LayerNorm(y + bias + residual) (16-bit, d % 4 == 0, d <= 2048; bias / residual may be NULL)
hipError_t cmr_launch_attention(const void* qkv, int dtype, const int* lens, int b, int L, int n_heads, void* out, hipStream_t s);
hipError_t cmr_launch_embed_layernorm(const long long* ids, const long long* tt, const void* word, const void* pos, const void* type,
                                      const void* gamma, const void* beta, float eps, long long rows, int L, int d, int vocab, int n_pos,
                                      int n_types, int pos_off, int dtype, void* out, hipStream_t s);
hipError_t cmr_launch_add_layernorm(const void* y, const void* bias, const void* res, const void* gamma, const void* beta, float eps,
                                    long long rows, int d, int dtype, void* out, hipStream_t s);
// the LAST layer's LayerNorm(y + bias + residual) with the masked mean-pool + L2-normalise of the encoder tail folded in: the
// [b, l, d] hidden state is never written (l % 16 == 0, d % 8 == 0, d <= 2048; partial = b * (l / 16) * d floats of scratch)
hipError_t cmr_launch_add_layernorm_pool(const void* y, const void* bias, const void* res, const void* gamma, const void* beta, float eps, int b, int l,
                                         int d, int dtype, const int* lens, int normalize, float* partial, float* out, hipStream_t s);
// BertEmbeddings from RAGGED token ids (int32, sequences back to back; off[b + 1] their starts): row (s, t) of the [b, L] mini-batch
hipError_t cmr_launch_embed_layernorm_ragged(const int* ids32, const int* off, const void* word, const void* pos, const void* type, const void* gamma,
                                             const void* beta, float eps, long long rows, int L, int d, int vocab, int n_pos, int pos_off, int dtype,
                                             void* out, hipStream_t s);
