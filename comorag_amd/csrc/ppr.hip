// Personalised PageRank seeded by dense-retrieval scores, on the device.
//
// Replaces, for one query, the tail of ComoRAG.graph_search_with_fact_entities (src/comorag/ComoRAG.py:1034-1044: all N
// (passage id, normalised DPR score) pairs are copied to the host, scattered one by one into `passage_weights` through
// passage_node_keys -> node_name_to_vertex_idx) and ComoRAG.run_ppr (:1086-1105: igraph's prpack personalised PageRank,
// undirected, edge attribute 'weight', damping 0.5, reset = node weights; then doc_scores = pagerank[passage_node_idxs]).
// Here the N scores never leave HBM: scan -> global min / max -> scatter of min_max(score) * passage_node_weight into the
// reset vector through the passage-row -> vertex map -> power iteration on a CSR copy of the graph -> gather of the
// passage vertices; only n_passages doubles come back (8 * n_passages bytes instead of 12 * N + the igraph call).
//
// Semantics restated from igraph_personalized_pagerank(PRPACK): reset vector r (negative / NaN entries -> 0, then divided
// by its sum); transition i -> j with probability w_ij / s_i (s_i = sum of i's incident weights; undirected = both
// directions); a vertex without edges jumps according to r; x = d * (P^T x + (sum of dangling x) * r) + (1 - d) * r.
// prpack solves this system directly to ~1e-10; a power iteration contracts by d per step, so ceil(log(tol/2)/log(d))
// steps reach tol in the 1-norm (43 steps for tol = 1e-12 at d = 0.5).  fp64 throughout, fixed summation order
// (pull-style CSR rows, block-ordered reductions): results are reproducible bit for bit.
// (Round 3 built and measured a single-launch variant for ComoRAG-sized graphs — one workgroup, x / y in LDS, a thread's
// rows advanced together: 1064 us per query at 5 K passages / 1.5 K entities against 551 us for this chain of ~50 launches;
// one CU's L2-latency-bound row walks lose against 43 grid-wide steps.  Dropped.  gpurun_out of the round: profiles/r3_measurements.md.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "../../include/comorag_hip.h"
#include "cmr_kernels.h"

int cmr_fail(int code, const char* fmt, ...);                                                                     // api.hip
// api.hip: scores of ONE host query into a device buffer of the index's workspace (shared index lock held, workspace
// reserved for the calling thread) until cmr_index_scores_release, which also reports a non-finite query
int cmr_index_scores_to_device(cmr_index_t* idx, const float* q_host, float** scores_dev, long long* n, void** stream);
int cmr_index_scores_release(cmr_index_t* idx);

struct PprScratch {
    double *reset = nullptr, *x = nullptr, *y = nullptr, *red = nullptr, *out = nullptr;
    float2* mm = nullptr;              // per-block (min, max) partials of the raw scores
    int* seed_v = nullptr;
    double* seed_w = nullptr;
    long long seed_cap = 0, out_cap = 0;
    // the power iteration of THIS scratch as an instantiated hipGraph (clean + normalise + `iters` steps: every argument is
    // one of this scratch's pointers or a graph constant), keyed by (damping, iters); replayed with one hipGraphLaunch
    hipGraphExec_t iter_exec = nullptr;
    double iter_damping = 0.0;
    int iter_count = 0;
    double* iter_result = nullptr;
    hipStream_t own = nullptr;         // cmr_graph_ppr's stream (capturable, unlike the legacy default stream)
    void release() {
        if (iter_exec) (void)hipGraphExecDestroy(iter_exec);
        if (own) (void)hipStreamDestroy(own);
        for (void* p : {(void*)reset, (void*)x, (void*)y, (void*)red, (void*)out, (void*)mm, (void*)seed_v, (void*)seed_w})
            if (p) (void)hipFree(p);
    }
};

struct cmr_graph {
    int device = 0;
    long long nv = 0, ne = 0;          // vertices, directed CSR entries (2 x undirected edges, self-loops once)
    long long* rowptr = nullptr;       // [nv + 1]
    int* col = nullptr;                // [ne]
    double* wnorm = nullptr;           // [ne]  w_ij / s_j of the SOURCE j of the pulled term (so y_i = sum_j wnorm * x_j)
    int* dangling = nullptr;           // vertices with no incident weight (internal ids, ascending)
    long long n_dangling = 0;
    // Vertices are REORDERED at cmr_graph_create by degree class (the exported ids stay the caller's): internal order = rows of more
    // than PPR_WAVE_DEG entries (a wave each), rows of PPR_ONE_DEG + 1 .. PPR_WAVE_DEG entries (eight lanes each), rows of <= PPR_ONE_DEG
    // entries (ONE thread each, stored as 4-slot ELL records: one 16-byte column load + two 16-byte weight loads per thread).  A step then
    // runs ~4x fewer waves than eight lanes for every row did — the step is latency x occupancy bound: waves in flight x ~4 dependent
    // memory round trips each — and a hub vertex of thousands of edges is no longer a 1000-iteration tail on eight lanes.
    long long n_wave = 0, n_oct = 0, n_one = 0;
    int4* ell_col = nullptr;           // [n_one] columns of a short row (absent slots: column 0 with weight 0)
    double* ell_w = nullptr;           // [n_one][4]
    int* perm = nullptr;               // device: caller's vertex id -> internal vertex
    std::vector<int> perm_h;           // the same on the host (seed vertices, the passage map)
    int* vertex_of_row = nullptr;      // passage row -> INTERNAL vertex
    long long n_rows = 0;
    // Per-call scratch.  ComoRAG runs graph_search_with_fact_entities from up to 16 threads at once (ComoRAG.try_answer's
    // ThreadPoolExecutor, ComoRAG.py:437) and ctypes releases the GIL: every call takes its own set of vectors from this
    // pool (grown on demand, one set per concurrent caller), so concurrent queries on one graph never share a reset / x / y.
    std::mutex mu;                     // guards the pool and the passage-vertex map's replacement
    std::vector<PprScratch*> pool;
    int users = 0;                     // calls in flight (cmr_graph_set_passage_vertices waits for none)
    std::atomic<bool> use_graph{true}; // power iteration replayed as a captured hipGraph (cleared if capture ever fails)
    std::condition_variable idle;
};

#define PPR_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return cmr_fail(e_ == hipErrorOutOfMemory ? CMR_ERR_OOM : CMR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------ kernels
#define PPR_T 256
#define PPR_RED_BLOCKS 256

__device__ __forceinline__ double block_sum(double v, double* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < PPR_T / 64; ++w) t += sh[w];      // fixed order
    __syncthreads();
    return t;
}

// per-block partial (min, max) of the raw scores
__global__ __launch_bounds__(PPR_T) void ppr_minmax_partial_kernel(const float* __restrict__ s, long long n, float2* __restrict__ part) {
    __shared__ float smn[PPR_T / 64], smx[PPR_T / 64];
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (long long i = (long long)blockIdx.x * PPR_T + threadIdx.x; i < n; i += (long long)gridDim.x * PPR_T) { const float v = s[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < PPR_T / 64; ++w) { mn = fminf(mn, smn[w]); mx = fmaxf(mx, smx[w]); }
        part[blockIdx.x] = make_float2(mn, mx);
    }
}

// reset[vertex_of_row[i]] = min_max_normalize(scores)[i] * pnw  (the reference's fp32 formula, utils/misc_utils.py:141-150;
// applied twice there (ComoRAG.py:963, :1035) — the second application is the identity), then the product in fp64 as
// numpy does for float32 * python float
__global__ __launch_bounds__(PPR_T) void ppr_scatter_kernel(const float* __restrict__ s, long long n, const float2* __restrict__ part, int nparts,
                                                            const int* __restrict__ vertex_of_row, double pnw, double* __restrict__ reset) {
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (int b = 0; b < nparts; ++b) { mn = fminf(mn, part[b].x); mx = fmaxf(mx, part[b].y); }
    const float range = mx - mn;
    const long long i = (long long)blockIdx.x * PPR_T + threadIdx.x;
    if (i >= n) return;
    const float norm = range == 0.0f ? 1.0f : (s[i] - mn) / range;
    reset[vertex_of_row[i]] = (double)norm * pnw;
}

__global__ __launch_bounds__(PPR_T) void ppr_seed_kernel(const int* __restrict__ v, const double* __restrict__ w, long long n, double* __restrict__ reset) {
    const long long i = (long long)blockIdx.x * PPR_T + threadIdx.x;
    if (i < n) reset[v[i]] += w[i];          // seed vertices are distinct here: the host sums duplicates first (merge_seeds), in input order
}

// reset <- max(reset, 0) with NaN -> 0 (ComoRAG.py:1090); partial sums per block
__global__ __launch_bounds__(PPR_T) void ppr_clean_sum_kernel(double* __restrict__ r, long long nv, double* __restrict__ part) {
    __shared__ double sh[PPR_T / 64];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * PPR_T + threadIdx.x; i < nv; i += (long long)gridDim.x * PPR_T) {
        double v = r[i];
        if (!(v >= 0.0)) v = 0.0;
        r[i] = v;
        acc += v;
    }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// r <- r / sum, x <- r
__global__ __launch_bounds__(PPR_T) void ppr_normalise_kernel(double* __restrict__ r, double* __restrict__ x, long long nv, const double* __restrict__ part, int nparts) {
    double tot = 0.0;
    for (int b = 0; b < nparts; ++b) tot += part[b];
    const long long i = (long long)blockIdx.x * PPR_T + threadIdx.x;
    if (i >= nv) return;
    const double v = tot > 0.0 ? r[i] / tot : 1.0 / (double)nv;
    r[i] = v;
    x[i] = v;
}

// mass sitting on vertices without edges (single block, fixed order)
__global__ __launch_bounds__(PPR_T) void ppr_dangling_kernel(const double* __restrict__ x, const int* __restrict__ dang, long long nd, double* __restrict__ out) {
    __shared__ double sh[PPR_T / 64];
    double acc = 0.0;
    for (long long i = threadIdx.x; i < nd; i += PPR_T) acc += x[dang[i]];
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = t;
}

// y_i = d * (sum_{j in N(i)} wnorm_ij * x_j + D * r_i) + (1 - d) * r_i
// Three row classes in one launch (block ranges; see cmr_graph): a wave per long row, PPR_LPR lanes per medium row, a thread per
// short row.  Every row has ONE summation order: lanes stride over a row's entries (coalesced col / wnorm reads) and sum theirs in
// ascending order, the partials are combined by a fixed xor tree; a short row adds its four slots in slot order — reproducible bit
// for bit.  (With one thread per vertex for every row a step lasted as long as the LONGEST row's chain of dependent loads; with
// eight lanes for every row — round 2 — a 3-entry passage row left five lanes idle and the step ran 150 K waves at 1 M passages.)
#define PPR_LPR 8
#define PPR_ONE_DEG 4
#define PPR_WAVE_DEG 256
// A lane's share of a row: entries e0, e0 + STRIDE, ... < e1, summed in that order.  Four entries per round: their column / weight loads,
// then their four gathers, are independent and in flight together — a lane's chain is (rowptr -> columns -> x) per ROUND, not per
// entry (an entity row of 19 entries on eight lanes was three dependent col -> x round trips; now one).  An absent entry contributes
// 0.0 * x[0]: adding +0.0 changes nothing, so the sum equals the one-entry-at-a-time loop bit for bit.
template <int STRIDE>
__device__ __forceinline__ double ppr_row_sum(long long e0, long long e1, const int* __restrict__ col, const double* __restrict__ wnorm,
                                              const double* __restrict__ x) {
    double acc = 0.0;
    for (long long e = e0; e < e1; e += 4 * STRIDE) {
        int c[4];
        double w[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long ee = e + (long long)u * STRIDE;
            const bool ok = ee < e1;
            c[u] = ok ? col[ee] : 0;
            w[u] = ok ? wnorm[ee] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += w[u] * xv[u];
    }
    return acc;
}
__global__ __launch_bounds__(PPR_T) void ppr_step_kernel(const long long* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ wnorm,
                                                         const int4* __restrict__ ell_col, const double* __restrict__ ell_w,
                                                         const double* __restrict__ x, const double* __restrict__ r, const double* __restrict__ dmass,
                                                         double d, long long n_wave, long long n_oct, long long n_one, unsigned b_wave, unsigned b_oct,
                                                         double* __restrict__ y) {
    const double D = dmass ? dmass[0] : 0.0;
    if (blockIdx.x < b_wave) {                                   // a wave per row
        const long long i = (long long)blockIdx.x * (PPR_T / 64) + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
        double acc = 0.0;
        if (i < n_wave) acc = ppr_row_sum<64>(rowptr[i] + lane, rowptr[i + 1], col, wnorm, x);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (i < n_wave && lane == 0) y[i] = d * (acc + D * r[i]) + (1.0 - d) * r[i];
    } else if (blockIdx.x < b_wave + b_oct) {                    // eight lanes per row
        const long long gt = (long long)(blockIdx.x - b_wave) * PPR_T + threadIdx.x;
        const long long i = n_wave + gt / PPR_LPR;
        const int sub = (int)(gt % PPR_LPR);
        const bool in = i < n_wave + n_oct;
        double acc = 0.0;
        if (in) acc = ppr_row_sum<PPR_LPR>(rowptr[i] + sub, rowptr[i + 1], col, wnorm, x);
#pragma unroll
        for (int off = PPR_LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (in && sub == 0) y[i] = d * (acc + D * r[i]) + (1.0 - d) * r[i];
    } else {                                                     // a thread per row: four ELL slots
        const long long t = (long long)(blockIdx.x - b_wave - b_oct) * PPR_T + threadIdx.x;
        if (t >= n_one) return;
        const long long i = n_wave + n_oct + t;
        const int4 c = ell_col[t];
        const double2 w01 = reinterpret_cast<const double2*>(ell_w)[2 * t], w23 = reinterpret_cast<const double2*>(ell_w)[2 * t + 1];
        const double x0 = x[c.x], x1 = x[c.y], x2 = x[c.z], x3 = x[c.w];      // four independent gathers in flight
        double acc = w01.x * x0;
        acc += w01.y * x1;
        acc += w23.x * x2;
        acc += w23.y * x3;
        y[i] = d * (acc + D * r[i]) + (1.0 - d) * r[i];
    }
}

// caller's order <-> internal order
__global__ __launch_bounds__(PPR_T) void ppr_permute_in_kernel(const double* __restrict__ src, const int* __restrict__ perm, long long nv, double* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * PPR_T + threadIdx.x;
    if (i < nv) dst[perm[i]] = src[i];
}
__global__ __launch_bounds__(PPR_T) void ppr_permute_out_kernel(const double* __restrict__ src, const int* __restrict__ perm, long long nv, double* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * PPR_T + threadIdx.x;
    if (i < nv) dst[i] = src[perm[i]];
}

__global__ __launch_bounds__(PPR_T) void ppr_gather_kernel(const double* __restrict__ x, const int* __restrict__ vertex_of_row, long long n, double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * PPR_T + threadIdx.x;
    if (i < n) out[i] = x[vertex_of_row[i]];
}

static unsigned blocks_for(long long n) { return (unsigned)std::max<long long>(1, (n + PPR_T - 1) / PPR_T); }

// One set of per-call vectors out of the graph's pool (allocated on first use, one per concurrent caller).
static int scratch_acquire(cmr_graph* g, PprScratch** out) {
    *out = nullptr;
    PprScratch* sc = nullptr;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        if (!g->pool.empty()) { sc = g->pool.back(); g->pool.pop_back(); }
        g->users++;
    }
    auto give_up = [&](hipError_t e, const char* what) {
        if (sc) { sc->release(); delete sc; }
        { std::lock_guard<std::mutex> lk(g->mu); g->users--; }
        g->idle.notify_all();
        return cmr_fail(e == hipErrorOutOfMemory ? CMR_ERR_OOM : CMR_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    };
    if (!sc) {
        sc = new PprScratch();
        hipError_t e = hipMalloc((void**)&sc->reset, (size_t)g->nv * 8);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->x, (size_t)g->nv * 8);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->y, (size_t)g->nv * 8);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->red, (PPR_RED_BLOCKS + 8) * 8);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->mm, PPR_RED_BLOCKS * sizeof(float2));
        if (e != hipSuccess) return give_up(e, "PPR scratch");
    }
    if (sc->out_cap < g->n_rows) {
        if (sc->out) (void)hipFree(sc->out);
        sc->out = nullptr; sc->out_cap = 0;
        hipError_t e = hipMalloc((void**)&sc->out, std::max<size_t>((size_t)g->n_rows * 8, 8));
        if (e != hipSuccess) return give_up(e, "PPR output scratch");
        sc->out_cap = g->n_rows;
    }
    *out = sc;
    return CMR_OK;
}
static void scratch_release(cmr_graph* g, PprScratch* sc) {
    { std::lock_guard<std::mutex> lk(g->mu); g->pool.push_back(sc); g->users--; }
    g->idle.notify_all();
}
struct ScratchGuard {
    cmr_graph* g; PprScratch* sc;
    ~ScratchGuard() { if (sc) scratch_release(g, sc); }
};

static int ppr_iters(double damping, double tol, int max_iter) {
    int iters = (int)std::ceil(std::log(std::max(tol, 1e-300) / 2.0) / std::log(std::min(std::max(damping, 1e-12), 1.0 - 1e-12)));
    iters = std::max(1, std::min(iters, max_iter > 0 ? max_iter : 1000));
    if (damping <= 0.0) iters = 1;
    return iters;
}

// reset (device, raw) -> normalised -> power iteration -> *result holds the stationary vector (sc->x or sc->y)
static void ppr_iterate_launches(cmr_graph* g, PprScratch* sc, double damping, int iters, hipStream_t s, double** result) {
    const int nparts = (int)std::min<long long>(PPR_RED_BLOCKS, blocks_for(g->nv));
    hipLaunchKernelGGL(ppr_clean_sum_kernel, dim3(nparts), dim3(PPR_T), 0, s, sc->reset, g->nv, sc->red);
    hipLaunchKernelGGL(ppr_normalise_kernel, dim3(blocks_for(g->nv)), dim3(PPR_T), 0, s, sc->reset, sc->x, g->nv, sc->red, nparts);
    double *x = sc->x, *y = sc->y;
    for (int it = 0; it < iters; ++it) {
        if (g->n_dangling) hipLaunchKernelGGL(ppr_dangling_kernel, dim3(1), dim3(PPR_T), 0, s, x, g->dangling, g->n_dangling, sc->red + PPR_RED_BLOCKS);
        const unsigned b_wave = g->n_wave ? (unsigned)((g->n_wave + PPR_T / 64 - 1) / (PPR_T / 64)) : 0u;
        const unsigned b_oct = g->n_oct ? blocks_for(g->n_oct * PPR_LPR) : 0u, b_one = g->n_one ? blocks_for(g->n_one) : 0u;
        hipLaunchKernelGGL(ppr_step_kernel, dim3(std::max(1u, b_wave + b_oct + b_one)), dim3(PPR_T), 0, s, g->rowptr, g->col, g->wnorm, g->ell_col, g->ell_w, x,
                           sc->reset, g->n_dangling ? sc->red + PPR_RED_BLOCKS : nullptr, damping, g->n_wave, g->n_oct, g->n_one, b_wave, b_oct, y);
        std::swap(x, y);
    }
    *result = x;
}

// The iteration is ~45-90 dependent launches of a few microseconds each: launch-bound.  They are captured ONCE per scratch
// into a hipGraph (every argument is a pointer of this scratch, the graph's CSR arrays or a constant) and replayed with a
// single hipGraphLaunch per query; any other (damping, iteration count) re-captures.  If capture or instantiation fails the
// plain launches run — same kernels, same order, same results.
static int ppr_iterate(cmr_graph* g, PprScratch* sc, double damping, double tol, int max_iter, hipStream_t s, int* iters_out, double** result) {
    const int iters = ppr_iters(damping, tol, max_iter);
    if (iters_out) *iters_out = iters;
    if (g->use_graph && (!sc->iter_exec || sc->iter_damping != damping || sc->iter_count != iters)) {
        if (sc->iter_exec) { (void)hipGraphExecDestroy(sc->iter_exec); sc->iter_exec = nullptr; }
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            double* res = nullptr;
            ppr_iterate_launches(g, sc, damping, iters, s, &res);
            if (hipStreamEndCapture(s, &graph) == hipSuccess && graph) {
                if (hipGraphInstantiate(&sc->iter_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    sc->iter_damping = damping; sc->iter_count = iters; sc->iter_result = res;
                } else {
                    sc->iter_exec = nullptr;
                }
                (void)hipGraphDestroy(graph);
            }
        }
        (void)hipGetLastError();                 // a failed capture must not poison the plain path below
        if (!sc->iter_exec) g->use_graph = false;
    }
    if (g->use_graph && sc->iter_exec) {
        PPR_TRY(hipGraphLaunch(sc->iter_exec, s));
        *result = sc->iter_result;
        return CMR_OK;
    }
    ppr_iterate_launches(g, sc, damping, iters, s, result);
    PPR_TRY(hipGetLastError());
    return CMR_OK;
}

static int ensure_seeds(PprScratch* sc, long long n) {
    if (n <= sc->seed_cap) return CMR_OK;
    if (sc->seed_v) PPR_TRY(hipFree(sc->seed_v));
    if (sc->seed_w) PPR_TRY(hipFree(sc->seed_w));
    sc->seed_v = nullptr; sc->seed_w = nullptr; sc->seed_cap = 0;
    PPR_TRY(hipMalloc((void**)&sc->seed_v, (size_t)n * 4));
    PPR_TRY(hipMalloc((void**)&sc->seed_w, (size_t)n * 8));
    sc->seed_cap = n;
    return CMR_OK;
}

// Duplicate seed vertices are SUMMED on the host, in input order: the seed kernel then adds every vertex once — no lost update,
// no atomics, one fixed summation order.  (This is the sparse API's own rule.  The reference's loop ASSIGNS:
// `phrase_weights[phrase_id] = fact_score` (ComoRAG.py:1019-1021, the last fact that names a phrase wins, then `/= num_chunk`);
// comorag_amd/hooks.py resolves that on the host and passes distinct vertices, so the rule here never meets a duplicate there.
// A caller of the sparse C-ABI who wants last-wins must resolve duplicates before the call.)
static void merge_seeds(const int32_t* v, const double* w, int n, std::vector<int>& ov, std::vector<double>& ow) {
    std::vector<std::pair<int, int>> order((size_t)n);
    for (int i = 0; i < n; ++i) order[i] = {v[i], i};
    std::stable_sort(order.begin(), order.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
    ov.clear(); ow.clear();
    for (int i = 0; i < n; ++i) {
        if (!ov.empty() && ov.back() == order[i].first) ow.back() += w[order[i].second];
        else { ov.push_back(order[i].first); ow.push_back(w[order[i].second]); }
    }
}

// ------------------------------------------------------------------------------------------ C-ABI
extern "C" {

int32_t cmr_graph_create(int32_t device_id, int64_t n_vertices, int64_t n_edges, const int32_t* src, const int32_t* dst, const double* weight,
                         cmr_graph_t** out) {
    if (!out) return cmr_fail(CMR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_vertices <= 0 || n_vertices >= (1ll << 31) || n_edges < 0 || (n_edges > 0 && (!src || !dst))) return cmr_fail(CMR_ERR_INVALID, "bad graph arguments");
    PPR_TRY(hipSetDevice(device_id));
    // symmetric CSR on the host: every undirected edge (u, v) contributes v to u's row and u to v's row (a self-loop once,
    // with its weight counted once in the strength), rows in ascending neighbour order so the sums have one fixed order
    std::vector<double> strength((size_t)n_vertices, 0.0);
    std::vector<long long> deg((size_t)n_vertices + 1, 0);
    for (int64_t e = 0; e < n_edges; ++e) {
        const int u = src[e], v = dst[e];
        if (u < 0 || v < 0 || u >= n_vertices || v >= n_vertices) return cmr_fail(CMR_ERR_INVALID, "edge %lld has a vertex outside [0, %lld)", (long long)e, (long long)n_vertices);
        const double w = weight ? weight[e] : 1.0;
        if (!(w >= 0.0)) return cmr_fail(CMR_ERR_INVALID, "edge %lld has a negative or NaN weight", (long long)e);
        strength[u] += w; deg[u + 1]++;
        if (u != v) { strength[v] += w; deg[v + 1]++; }
    }
    // degree classes and the internal order: long rows, medium rows, short rows — each class in the caller's order
    std::vector<int> perm((size_t)n_vertices);
    long long n_wave = 0, n_oct = 0, n_one = 0;
    for (int64_t i = 0; i < n_vertices; ++i) {
        const long long dg = deg[i + 1];
        if (dg > PPR_WAVE_DEG) ++n_wave; else if (dg > PPR_ONE_DEG) ++n_oct; else ++n_one;
    }
    {
        long long at_w = 0, at_o = n_wave, at_1 = n_wave + n_oct;
        for (int64_t i = 0; i < n_vertices; ++i) {
            const long long dg = deg[i + 1];
            perm[i] = (int)(dg > PPR_WAVE_DEG ? at_w++ : dg > PPR_ONE_DEG ? at_o++ : at_1++);
        }
    }
    for (int64_t i = 0; i < n_vertices; ++i) deg[i + 1] += deg[i];
    const long long ne = deg[n_vertices];
    std::vector<std::pair<int, double>> ent((size_t)ne);          // (INTERNAL neighbour, weight) by the caller's row
    std::vector<long long> fill(deg.begin(), deg.end() - 1);
    for (int64_t e = 0; e < n_edges; ++e) {
        const int u = src[e], v = dst[e];
        const double w = weight ? weight[e] : 1.0;
        ent[fill[u]++] = {perm[v], w};
        if (u != v) ent[fill[v]++] = {perm[u], w};
    }
    std::vector<double> strength_int((size_t)n_vertices);
    for (int64_t i = 0; i < n_vertices; ++i) strength_int[perm[i]] = strength[i];
    // internal CSR of the long and medium rows, ELL records of the short ones; a row's entries in ascending INTERNAL neighbour order
    // (stable: parallel edges keep their input order) — the one summation order of the row
    const long long n_csr = n_wave + n_oct;
    std::vector<long long> rowptr((size_t)n_csr + 1, 0);
    for (int64_t i = 0; i < n_vertices; ++i)
        if (perm[i] < n_csr) rowptr[perm[i] + 1] = deg[i + 1] - deg[i];
    for (long long i = 0; i < n_csr; ++i) rowptr[i + 1] += rowptr[i];
    std::vector<int> col((size_t)rowptr[n_csr]);
    std::vector<double> wn((size_t)rowptr[n_csr]);
    std::vector<int> ecol((size_t)n_one * 4, 0);
    std::vector<double> ew((size_t)n_one * 4, 0.0);
    std::vector<int> dang;
    for (int64_t i = 0; i < n_vertices; ++i) {
        std::stable_sort(ent.begin() + deg[i], ent.begin() + deg[i + 1], [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
        const int pi = perm[i];
        for (long long e = deg[i]; e < deg[i + 1]; ++e) {
            const int j = ent[e].first;
            const double wv = strength_int[j] > 0.0 ? ent[e].second / strength_int[j] : 0.0;      // mass leaving j along this edge
            if (pi < n_csr) { col[rowptr[pi] + (e - deg[i])] = j; wn[rowptr[pi] + (e - deg[i])] = wv; }
            else { ecol[(size_t)(pi - n_csr) * 4 + (e - deg[i])] = j; ew[(size_t)(pi - n_csr) * 4 + (e - deg[i])] = wv; }
        }
    }
    for (int64_t i = 0; i < n_vertices; ++i)
        if (!(strength[i] > 0.0)) dang.push_back(perm[i]);
    std::sort(dang.begin(), dang.end());
    cmr_graph* g = new cmr_graph();
    g->device = device_id; g->nv = n_vertices; g->ne = ne; g->n_dangling = (long long)dang.size();
    g->n_wave = n_wave; g->n_oct = n_oct; g->n_one = n_one;
    auto up = [&](void** p, const void* h, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*p, h, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    hipError_t e = up((void**)&g->rowptr, rowptr.data(), (size_t)(n_csr + 1) * 8);
    if (e == hipSuccess) e = up((void**)&g->col, col.data(), col.size() * 4);
    if (e == hipSuccess) e = up((void**)&g->wnorm, wn.data(), wn.size() * 8);
    if (e == hipSuccess) e = up((void**)&g->ell_col, ecol.data(), ecol.size() * 4);
    if (e == hipSuccess) e = up((void**)&g->ell_w, ew.data(), ew.size() * 8);
    if (e == hipSuccess) e = up((void**)&g->dangling, dang.data(), dang.size() * 4);
    if (e == hipSuccess) e = up((void**)&g->perm, perm.data(), perm.size() * 4);
    if (e != hipSuccess) { cmr_graph_destroy(g); return cmr_fail(e == hipErrorOutOfMemory ? CMR_ERR_OOM : CMR_ERR_HIP, "graph upload: %s", hipGetErrorString(e)); }
    g->perm_h = std::move(perm);
    *out = g;
    return CMR_OK;
}

int32_t cmr_graph_destroy(cmr_graph_t* g) {
    if (!g) return CMR_OK;
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();
    for (void* p : {(void*)g->rowptr, (void*)g->col, (void*)g->wnorm, (void*)g->dangling, (void*)g->vertex_of_row, (void*)g->ell_col, (void*)g->ell_w, (void*)g->perm})
        if (p) (void)hipFree(p);
    for (PprScratch* sc : g->pool) { sc->release(); delete sc; }
    delete g;
    return CMR_OK;
}

int32_t cmr_graph_set_passage_vertices(cmr_graph_t* g, const int32_t* vertex_of_row, int64_t n_rows) {
    if (!g || (n_rows > 0 && !vertex_of_row) || n_rows < 0) return cmr_fail(CMR_ERR_INVALID, "bad argument");
    for (int64_t i = 0; i < n_rows; ++i)
        if (vertex_of_row[i] < 0 || vertex_of_row[i] >= g->nv) return cmr_fail(CMR_ERR_INVALID, "row %lld maps to vertex %d outside the graph", (long long)i, vertex_of_row[i]);
    PPR_TRY(hipSetDevice(g->device));
    std::unique_lock<std::mutex> lk(g->mu);
    g->idle.wait(lk, [&] { return g->users == 0; });       // no query may be reading the old map
    if (g->vertex_of_row) PPR_TRY(hipFree(g->vertex_of_row));
    g->vertex_of_row = nullptr; g->n_rows = 0;
    PPR_TRY(hipMalloc((void**)&g->vertex_of_row, std::max<size_t>((size_t)n_rows * 4, 8)));
    if (n_rows) {
        std::vector<int> internal((size_t)n_rows);
        for (int64_t i = 0; i < n_rows; ++i) internal[i] = g->perm_h[vertex_of_row[i]];
        PPR_TRY(hipMemcpy(g->vertex_of_row, internal.data(), (size_t)n_rows * 4, hipMemcpyHostToDevice));
    }
    g->n_rows = n_rows;
    return CMR_OK;
}

int32_t cmr_graph_ppr(cmr_graph_t* g, const double* reset, double damping, double tol, int32_t max_iter, double* out_scores, int32_t* iters) {
    if (!g || !reset || !out_scores) return cmr_fail(CMR_ERR_INVALID, "NULL argument");
    PPR_TRY(hipSetDevice(g->device));
    PprScratch* sc = nullptr;
    int rc = scratch_acquire(g, &sc);
    if (rc) return rc;
    ScratchGuard guard{g, sc};
    if (!sc->own) PPR_TRY(hipStreamCreateWithFlags(&sc->own, hipStreamNonBlocking));
    hipStream_t s = sc->own;                                // concurrent callers do not queue behind each other on the null stream
    auto body = [&]() -> int {
        // the caller's vertex order on both sides of the ABI, the internal (degree-class) order between them
        PPR_TRY(hipMemcpyAsync(sc->x, reset, (size_t)g->nv * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(ppr_permute_in_kernel, dim3(blocks_for(g->nv)), dim3(PPR_T), 0, s, sc->x, g->perm, g->nv, sc->reset);
        double* res = nullptr;
        int rc_ = ppr_iterate(g, sc, damping, tol, max_iter, s, iters, &res);
        if (rc_) return rc_;
        double* tmp = res == sc->x ? sc->y : sc->x;
        hipLaunchKernelGGL(ppr_permute_out_kernel, dim3(blocks_for(g->nv)), dim3(PPR_T), 0, s, res, g->perm, g->nv, tmp);
        PPR_TRY(hipGetLastError());
        PPR_TRY(hipMemcpyAsync(out_scores, tmp, (size_t)g->nv * 8, hipMemcpyDeviceToHost, s));
        return CMR_OK;
    };
    rc = body();
    const hipError_t es = hipStreamSynchronize(s);          // also on error paths: nothing may still use the scratch when it goes back
    if (!rc && es != hipSuccess) rc = cmr_fail(CMR_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(es));
    return rc;
}

int32_t cmr_index_ppr(cmr_index_t* idx, cmr_graph_t* g, const float* q_f32, const int32_t* seed_vertices, const double* seed_weights, int32_t n_seeds,
                      double passage_node_weight, double damping, double tol, int32_t max_iter, double* out_doc_scores, int32_t* iters) {
    if (!idx || !g || !q_f32 || !out_doc_scores || (n_seeds > 0 && (!seed_vertices || !seed_weights)) || n_seeds < 0)
        return cmr_fail(CMR_ERR_INVALID, "bad argument");
    if (!g->vertex_of_row) return cmr_fail(CMR_ERR_INVALID, "cmr_graph_set_passage_vertices was not called");
    for (int i = 0; i < n_seeds; ++i)
        if (seed_vertices[i] < 0 || seed_vertices[i] >= g->nv) return cmr_fail(CMR_ERR_INVALID, "seed vertex %d outside the graph", seed_vertices[i]);
    std::vector<int> sv;
    std::vector<double> sw;
    merge_seeds(seed_vertices, seed_weights, n_seeds, sv, sw);
    for (int& v : sv) v = g->perm_h[v];                     // distinct vertices after the merge: their order no longer matters
    const int ns = (int)sv.size();
    PprScratch* sc = nullptr;
    int rc = scratch_acquire(g, &sc);
    if (rc) return rc;
    ScratchGuard guard{g, sc};
    float* scores = nullptr;
    long long n = 0;
    void* st = nullptr;
    rc = cmr_index_scores_to_device(idx, q_f32, &scores, &n, &st);       // scan; the scores stay in HBM (index lock held until release)
    if (rc) return rc;
    hipStream_t s = (hipStream_t)st;
    auto body = [&]() -> int {
        if (n != g->n_rows) return cmr_fail(CMR_ERR_INVALID, "index has %lld rows, the passage-vertex map %lld", n, g->n_rows);
        int rc_ = ensure_seeds(sc, std::max(ns, 1));
        if (rc_) return rc_;
        if (ns) {
            PPR_TRY(hipMemcpyAsync(sc->seed_v, sv.data(), (size_t)ns * 4, hipMemcpyHostToDevice, s));
            PPR_TRY(hipMemcpyAsync(sc->seed_w, sw.data(), (size_t)ns * 8, hipMemcpyHostToDevice, s));
        }
        PPR_TRY(hipMemsetAsync(sc->reset, 0, (size_t)g->nv * 8, s));
        const int nparts = (int)std::min<long long>(PPR_RED_BLOCKS, blocks_for(n));
        if (n) {
            hipLaunchKernelGGL(ppr_minmax_partial_kernel, dim3(nparts), dim3(PPR_T), 0, s, scores, n, sc->mm);
            hipLaunchKernelGGL(ppr_scatter_kernel, dim3(blocks_for(n)), dim3(PPR_T), 0, s, scores, n, sc->mm, nparts, g->vertex_of_row, passage_node_weight, sc->reset);
        }
        if (ns) hipLaunchKernelGGL(ppr_seed_kernel, dim3(blocks_for(ns)), dim3(PPR_T), 0, s, sc->seed_v, sc->seed_w, (long long)ns, sc->reset);
        double* res = nullptr;
        rc_ = ppr_iterate(g, sc, damping, tol, max_iter, s, iters, &res);
        if (rc_) return rc_;
        if (n) hipLaunchKernelGGL(ppr_gather_kernel, dim3(blocks_for(n)), dim3(PPR_T), 0, s, res, g->vertex_of_row, n, sc->out);
        PPR_TRY(hipGetLastError());
        PPR_TRY(hipMemcpyAsync(out_doc_scores, sc->out, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        return CMR_OK;
    };
    rc = body();
    // Whatever happened, the stream is drained before the workspace and the scratch go back: kernels enqueued ahead of a
    // failing call would otherwise still be running on buffers the next caller reuses.
    const hipError_t es = hipStreamSynchronize(s);
    const int rc_rel = cmr_index_scores_release(idx);       // CMR_ERR_NONFINITE if the query held NaN / Inf
    if (rc) return rc;
    if (es != hipSuccess) return cmr_fail(CMR_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(es));
    return rc_rel;
}

}  // extern "C"
