// Small kernels around the corpus scan: operand packing (queries, appended rows), candidate
// merges, exact re-score, row gather and the encoder tail (masked mean-pool + L2 normalise).
// All are launch-latency or HBM bound; none uses MFMA.
#include <algorithm>

#include "cmr_device.h"
#include "cmr_kernels.h"
#include "cmr_select.h"

// ------------------------------------------------------------------------------------------
// Pack element value by dtype.
template <int DT> struct Pack;
template <> struct Pack<CMR_DT_BF16> { static __device__ __forceinline__ unsigned short cvt(float f) { return cmr_f2bf(f); } };
template <> struct Pack<CMR_DT_F16>  { static __device__ __forceinline__ unsigned short cvt(float f) { return cmr_f2h(f); } };

__device__ __forceinline__ bool cmr_finite(float f) { return (__float_as_uint(f) & 0x7F800000u) != 0x7F800000u; }

// One thread builds one 16-byte lane slot of one block from a row-major fp32 source.
//   src_row(row) -> pointer to the row's dim floats, or nullptr for a zero row
template <int DT>
__device__ __forceinline__ uint4 cmr_pack_slot(const float* row, int dim, int ks, int lane, bool& bad) {
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (row) {
        if (DT == CMR_DT_F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = cmr_blk_k<CMR_DT_F32>(ks, lane, e);
                const float f = k < dim ? row[k] : 0.0f;
                bad |= !cmr_finite(f);
                w[e] = __float_as_uint(f);
            }
        } else {
            const int k0 = cmr_blk_k<DT == CMR_DT_F32 ? CMR_DT_BF16 : DT>(ks, lane, 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                const float f = k < dim ? row[k] : 0.0f;
                bad |= !cmr_finite(f);
                unsigned short h;
                if (DT == CMR_DT_BF16) h = Pack<CMR_DT_BF16>::cvt(f); else h = Pack<CMR_DT_F16>::cvt(f);
                w[e >> 1] |= (unsigned)h << (16 * (e & 1));
            }
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// queries [nq, dim] fp32 -> qfrag [nqt][ks][64] uint4.   grid = nqt*ks blocks of 64 threads.
template <int DT>
__global__ __launch_bounds__(64) void prep_queries_kernel(const float* __restrict__ q, int nq, int dim, int ks_total,
                                                          uint4* __restrict__ qfrag, int* __restrict__ flag) {
    const int blk = blockIdx.x;          // t*ks_total + ks
    const int t = blk / ks_total, ks = blk % ks_total;
    const int lane = threadIdx.x;
    const int qi = t * 32 + (lane & 31);
    bool bad = false;
    const uint4 v = cmr_pack_slot<DT>(qi < nq ? q + (size_t)qi * dim : nullptr, dim, ks, lane, bad);
    qfrag[(size_t)blk * 64 + lane] = v;
    if (bad) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (every writer writes 1) the flag may live in mapped host memory: visible there before the done word
}

// tau[i] = (smallest candidate key whose score is >= min_score) - 1, so that key > tau <=> score >= min_score
__global__ void fill_threshold_kernel(float min_score, int n, u64* __restrict__ tau) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) tau[i] = cmr_make_key(min_score, 0xFFFFFFFFu) - 1ull;
}
hipError_t cmr_launch_fill_threshold(float min_score, int n, u64* tau, hipStream_t s) {
    hipLaunchKernelGGL(fill_threshold_kernel, dim3((n + 255) / 256), dim3(256), 0, s, min_score, n, tau);
    return hipGetLastError();
}

hipError_t cmr_launch_prep_queries(int dtype, const float* q, int nq, int dim, int dpad, int nqt, void* qfrag,
                                   int* flag, hipStream_t s) {
    const int ks = dtype == CMR_DT_F32 ? dpad / 8 : dpad / 16;
    dim3 grid(nqt * ks), block(64);
    uint4* o = reinterpret_cast<uint4*>(qfrag);
    switch (dtype) {
        case CMR_DT_BF16: hipLaunchKernelGGL(prep_queries_kernel<CMR_DT_BF16>, grid, block, 0, s, q, nq, dim, ks, o, flag); break;
        case CMR_DT_F16:  hipLaunchKernelGGL(prep_queries_kernel<CMR_DT_F16>, grid, block, 0, s, q, nq, dim, ks, o, flag); break;
        case CMR_DT_F32:  hipLaunchKernelGGL(prep_queries_kernel<CMR_DT_F32>, grid, block, 0, s, q, nq, dim, ks, o, flag); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// rows [n, dim] fp32 -> panel-major blocks starting at global row row0.
// Work item = (panel-local row, half, ks) of one touched panel: a wave writes up to 64 adjacent
// 16-byte slots of ONE block (coalesced), reading 32-byte pieces of 32 source rows.
template <int DT>
__global__ __launch_bounds__(256) void convert_rows_kernel(const float* __restrict__ rows, long long n, int dim, int ks_total,
                                                           long long row0, uint4* __restrict__ corpus,
                                                           float* __restrict__ shadow, int* __restrict__ flag) {
    const long long first_panel = row0 / CMR_PANEL_ROWS;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;   // over touched_panels * ks * 64
    const int lane = (int)(gid & 63);
    const long long blk = gid >> 6;
    const long long panel = first_panel + blk / ks_total;
    const int ks = (int)(blk % ks_total);
    const long long row = panel * CMR_PANEL_ROWS + (lane & 31);
    if (row < row0 || row >= row0 + n) return;   // slot belongs to an older row or to padding
    const float* src = rows + (size_t)(row - row0) * dim;
    bool bad = false;
    const uint4 v = cmr_pack_slot<DT>(src, dim, ks, lane, bad);
    corpus[((size_t)panel * ks_total + ks) * 64 + lane] = v;
    if (bad) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (every writer writes 1) the flag may live in mapped host memory: visible there before the done word
    if (shadow && DT != CMR_DT_F32) {
        // fp32 shadow, row-major [*, dim]: this thread owns the same 8 k of the row
        const int k0 = cmr_blk_k<DT == CMR_DT_F32 ? CMR_DT_BF16 : DT>(ks, lane, 0);
        for (int e = 0; e < 8; ++e)
            if (k0 + e < dim) shadow[(size_t)row * dim + k0 + e] = src[k0 + e];
    }
}

hipError_t cmr_launch_convert_rows(int dtype, const float* rows, long long n, int dim, int dpad, long long row0,
                                   void* corpus, float* shadow, int* flag, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int ks = dtype == CMR_DT_F32 ? dpad / 8 : dpad / 16;
    const long long p_first = row0 / CMR_PANEL_ROWS, p_last = (row0 + n - 1) / CMR_PANEL_ROWS;
    const long long items = (p_last - p_first + 1) * ks * 64;
    dim3 grid((unsigned)((items + 255) / 256)), block(256);
    uint4* o = reinterpret_cast<uint4*>(corpus);
    switch (dtype) {
        case CMR_DT_BF16: hipLaunchKernelGGL(convert_rows_kernel<CMR_DT_BF16>, grid, block, 0, s, rows, n, dim, ks, row0, o, shadow, flag); break;
        case CMR_DT_F16:  hipLaunchKernelGGL(convert_rows_kernel<CMR_DT_F16>, grid, block, 0, s, rows, n, dim, ks, row0, o, shadow, flag); break;
        case CMR_DT_F32:  hipLaunchKernelGGL(convert_rows_kernel<CMR_DT_F32>, grid, block, 0, s, rows, n, dim, ks, row0, o, shadow, flag); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Candidate merge: ONE workgroup (256 threads, one wave per SIMD) per query gathers the W per-wave lists of the
// scan (sparse: the sampling threshold keeps them at a few entries each) and selects the k best
// keys.  The concatenated lists are consumed in chunks of 4096 keys held in registers (16 per
// thread) with the running top-k carried over, so any total (up to W*cap) is handled; the common
// case is a single chunk.
//   sample pass (out_tau != nullptr): publishes (k-th best key) - 1 as the main scan's threshold
//   main pass: writes ids/scores (+ id_base) and reduces the per-wave min/max partials.
// Two shapes of the same kernel: 256 threads x 16 keys per thread (few lists: the sampling merges, the pipelined batches whose
// merges run beside the next scan) and 1024 threads x 4 keys (a synchronous caller's main-pass merge over thousands of lists is
// a latency chain behind the scan: the list-length prefix, the gather and every select pass touch a quarter of the slots per
// thread).  MERGE_POOL keys per chunk either way.
#define MERGE_POOL 4096

// Radix select of the k largest keys held in the block's registers (MERGE_PER_THREAD+1 per
// thread, 0 = empty): 8 passes over the key bytes, most significant first.  Each pass histograms
// the current digit of the keys that still match the selected prefix (256 bins = 256 threads), a
// suffix scan finds the digit that contains the k-th largest key, and the prefix grows by one byte;
// after the last pass the prefix IS the k-th largest key (keys are unique).  Keys >= it are
// collected (exactly k of them) and ordered by rank counting.  Cost is independent of k.
//   scratch: hist[256] ints, wsum[MERGE_WAVES] ints, sel[4] ints (digit, k_rem, winner counter, bin count), cand[k] u64
template <int MERGE_THREADS, int MERGE_PER_THREAD>
__device__ __forceinline__ void merge_select_regs(u64 (&e)[MERGE_PER_THREAD + 1], int k, int* hist, int* wsum, int* sel,
                                                  u64* cand, u64* res) {
    constexpr int MERGE_WAVES = MERGE_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // how many keys are there at all?
    int mine = 0;
#pragma unroll
    for (int j = 0; j <= MERGE_PER_THREAD; ++j) mine += e[j] != 0 ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0) wsum[wave] = mine;
    if (tid == 0) sel[2] = 0;
    for (int i = tid; i < k; i += MERGE_THREADS) { res[i] = 0; cand[i] = 0; }
    __syncthreads();
    int n_tot = 0;
#pragma unroll
    for (int w = 0; w < MERGE_WAVES; ++w) n_tot += wsum[w];
    __syncthreads();

    u64 thr = 1;   // n_tot <= k: every key qualifies
    if (n_tot > k) {
        u64 pval = 0, pmask = 0;
        int k_rem = k;
        for (int pass = 7; pass >= 0; --pass) {
            const int shift = pass * 8;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j <= MERGE_PER_THREAD; ++j) {
                const u64 key = e[j];
                const bool act = key != 0 && (key & pmask) == pval;
                const int dg = (int)((key >> shift) & 255u);
                const u64 am = __ballot(act);
                if (am) {   // wave-uniform
                    const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)am) - 1);
                    const int d0 = __builtin_amdgcn_readlane(dg, first);
                    if (__ballot(act && dg != d0) == 0) {      // all active lanes share the digit: one add
                        if (lane == first) atomicAdd(&hist[d0], __popcll(am));
                    } else if (act) {
                        atomicAdd(&hist[dg], 1);
                    }
                }
            }
            __syncthreads();
            const int c = tid < 256 ? hist[tid] : 0;      // 256 bins: the first four waves scan them
            int s = c;     // suffix sum inside the wave: bins tid..(wave end)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_down(s, off);
                if (lane + off < 64) s += o;
            }
            if (lane == 0 && wave < 4) wsum[wave] = s;
            __syncthreads();
            int higher = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) higher += w > wave ? wsum[w] : 0;
            const int S = s + higher;        // keys whose digit >= tid
            const int Sgt = S - c;           // keys whose digit >  tid
            if (tid < 256 && S >= k_rem && Sgt < k_rem) { sel[0] = tid; sel[1] = k_rem - Sgt; sel[3] = c; }
            __syncthreads();
            pval |= (u64)(unsigned)sel[0] << shift;
            pmask |= 0xFFull << shift;
            k_rem = sel[1];
            // Every key that still matches the prefix is wanted: done — the prefix with zero low bits separates the
            // winners from the rest.  Scores are almost unique, so this is the rule after three or four of the eight
            // passes (the low word of a key is the row: it only ever decides ties).
            if (sel[3] == k_rem) break;
        }
        thr = pval;
    }
    // collect the winners (exactly min(k, n_tot) keys >= thr), then order them
#pragma unroll
    for (int j = 0; j <= MERGE_PER_THREAD; ++j)
        if (e[j] != 0 && e[j] >= thr) cand[atomicAdd(&sel[2], 1)] = e[j];
    __syncthreads();
    for (int t = tid; t < k; t += MERGE_THREADS) {
        const u64 c = cand[t];
        if (c == 0) continue;
        int r = 0;
        for (int i = 0; i < k; ++i) r += cand[i] > c ? 1 : 0;
        res[r] = c;
    }
    __syncthreads();
}

template <int MERGE_THREADS, int MERGE_PER_THREAD>
__global__ __launch_bounds__(MERGE_THREADS) void merge_query_kernel(const u64* __restrict__ lists, const int* __restrict__ cnt,
                                                                    int W, int nq_stride, int cap, int k,
                                                                    const float2* __restrict__ mm, long long id_base,
                                                                    int64_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                                                    float* __restrict__ out_min, float* __restrict__ out_max,
                                                                    u64* __restrict__ out_tau, int grouped, const int* __restrict__ skip_if_one) {
    if (skip_if_one && *skip_if_one == 1) return;      // (block-uniform) the scan's finishing stage wrote the final results itself
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    u64* cand = reinterpret_cast<u64*>(sm);                 // k
    u64* res = cand + k;                                    // k
    int* prefix = reinterpret_cast<int*>(res + k);          // W+1
    constexpr int MERGE_WAVES = MERGE_THREADS / 64;
    int* wsum = prefix + (W + 1);                           // MERGE_WAVES
    int* hist = wsum + MERGE_WAVES;                         // 256
    int* sel = hist + 256;                                  // 4
    float* red = reinterpret_cast<float*>(sel + 4);         // 2*MERGE_WAVES
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // query-split scans keep one [W][nq_stride] array set per query group: output query blockIdx.x is query q of group g
    const int g = grouped ? blockIdx.x / nq_stride : 0;
    const int q = blockIdx.x - g * nq_stride;
    lists += (size_t)g * W * nq_stride * cap;
    cnt += (size_t)g * W * nq_stride;
    if (mm) mm += (size_t)g * W * nq_stride;
    const int qo = blockIdx.x;                              // where the query's results go

    // exclusive prefix sum of the W list lengths
    const int CH = (W + MERGE_THREADS - 1) / MERGE_THREADS;
    int c[16];
    int local = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j >= CH) break;
        const int w = tid * CH + j;
        if (w < W) { int v = cnt[(size_t)w * nq_stride + q]; c[j] = v < cap ? v : cap; }
        local += c[j];
    }
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w2 = 0; w2 < wave; ++w2) wbase += wsum[w2];
    int run = wbase + incl - local;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j >= CH) break;
        const int w = tid * CH + j;
        if (w < W) prefix[w] = run;
        run += c[j];
    }
    if (tid == MERGE_THREADS - 1) prefix[W] = run;
    for (int i = tid; i < k; i += MERGE_THREADS) res[i] = 0;
    __syncthreads();

    // Chunks of MERGE_POOL elements of the concatenated lists, gathered straight into registers
    // (binary search of the prefix array per element: 16 independent searches per thread); the
    // running top-k is carried in the 17th register slot of the first k threads.
    const int total = prefix[W];
    int base = 0;
    do {
        u64 e[MERGE_PER_THREAD + 1];
        e[MERGE_PER_THREAD] = (base > 0 && tid < k) ? res[tid] : 0ull;
        // 16 branch-free binary searches advanced in lock step (independent LDS reads per step),
        // then all 16 global loads issued back to back.
        int pos[MERGE_PER_THREAD];
#pragma unroll
        for (int j = 0; j < MERGE_PER_THREAD; ++j) pos[j] = 0;
        for (int step = 2048; step > 0; step >>= 1) {        // W <= 4096 lists
#pragma unroll
            for (int j = 0; j < MERGE_PER_THREAD; ++j) {
                const int v = base + tid + MERGE_THREADS * j;
                const int np = pos[j] + step;
                const int pv = prefix[np < W ? np : W];      // prefix[W] = total > any valid v
                pos[j] = (np < W && pv <= v) ? np : pos[j];  // largest w with prefix[w] <= v
            }
        }
#pragma unroll
        for (int j = 0; j < MERGE_PER_THREAD; ++j) {
            const int v = base + tid + MERGE_THREADS * j;
            const int w = pos[j];
            const size_t off = ((size_t)w * nq_stride + q) * cap + (size_t)(v < total ? v - prefix[w] : 0);
            const u64 key = lists[off];                      // always in bounds; masked below
            e[j] = v < total ? key : 0ull;
        }
        __syncthreads();     // every thread has read its carried key before the rounds rewrite res
        merge_select_regs<MERGE_THREADS, MERGE_PER_THREAD>(e, k, hist, wsum, sel, cand, res);
        base += MERGE_POOL;
    } while (base < total);

    if (out_tau) {
        if (tid == 0) out_tau[qo] = res[k - 1] ? res[k - 1] - 1 : 0ull;
        return;
    }
    for (int i = tid; i < k; i += MERGE_THREADS) {
        const u64 key = res[i];
        out_ids[(size_t)qo * k + i] = key ? (int64_t)cmr_key_row(key) + id_base : -1;
        out_scores[(size_t)qo * k + i] = key ? cmr_key_score(key) : -__builtin_inff();
    }
    if (out_min || out_max) {
        float mn = __builtin_inff(), mx = -__builtin_inff();
        for (int w = tid; w < W; w += MERGE_THREADS) {
            const float2 v = mm[(size_t)w * nq_stride + q];
            mn = fminf(mn, v.x); mx = fmaxf(mx, v.y);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
        if (lane == 0) { red[wave] = mn; red[MERGE_WAVES + wave] = mx; }
        __syncthreads();
        if (tid == 0) {
            for (int w2 = 1; w2 < MERGE_WAVES; ++w2) { mn = fminf(mn, red[w2]); mx = fmaxf(mx, red[MERGE_WAVES + w2]); }
            if (out_min) out_min[qo] = mn;
            if (out_max) out_max[qo] = mx;
        }
    }
}

hipError_t cmr_launch_merge_query(const u64* lists, const int* cnt, int W, int nq_stride, int cap, int nq, int k,
                                  const float2* mm, long long id_base, int64_t* out_ids, float* out_scores,
                                  float* out_min, float* out_max, u64* out_tau, hipStream_t s, bool grouped, const int* skip_if_one) {
    if (W > 16 * 256) return hipErrorInvalidValue;
    // a handful of queries over more than a thousand lists (a synchronous call's main pass): the 1024-thread shape
    const bool big = nq <= 8 && W > 1024;
    const int waves = big ? 16 : 4;
    const size_t lds = (size_t)2 * k * 8 + (size_t)(W + 1) * 4 + waves * 4 + 256 * 4 + 4 * 4 + 2 * waves * 4;
    auto launch = [&](auto kern, int threads) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);     // a constant: per function, not per launch
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nq), dim3(threads), lds, s, lists, cnt, W, nq_stride, cap, k, mm, id_base, out_ids, out_scores, out_min, out_max, out_tau, grouped ? 1 : 0, skip_if_one);
        return hipGetLastError();
    };
    if (big) return launch(merge_query_kernel<1024, 4>, 1024);
    return launch(merge_query_kernel<256, 16>, 256);
}

// Shard merge: ids/scores [S][nq][k] (global ids, -1 = empty) -> [nq][k], same order rule.
// Keys are rebuilt on the fly; ids up to 2^63 are kept by ranking on (score, id) pairs directly.
__global__ __launch_bounds__(64) void merge_shards_kernel(const int64_t* __restrict__ ids, const float* __restrict__ scores, int S,
                                                          int nq, int k, int64_t* __restrict__ out_ids,
                                                          float* __restrict__ out_scores) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const int n = S * k;
    // rank by counting (n = S*k is small: 8*20 = 160).  Entry i beats j if score higher, or equal and id lower.
    for (int i = lane; i < k; i += 64) { out_ids[(size_t)q * k + i] = -1; out_scores[(size_t)q * k + i] = -__builtin_inff(); }
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        const int s = i / k, j = i % k;
        const int64_t id = ids[((size_t)s * nq + q) * k + j];
        if (id < 0) continue;
        const float sc = scores[((size_t)s * nq + q) * k + j] + 0.0f;
        int rank = 0;
        for (int o = 0; o < n; ++o) {
            const int so = o / k, jo = o % k;
            const int64_t ido = ids[((size_t)so * nq + q) * k + jo];
            if (ido < 0) continue;
            const float sco = scores[((size_t)so * nq + q) * k + jo] + 0.0f;
            rank += (sco > sc) || (sco == sc && ido < id);
        }
        if (rank < k) { out_ids[(size_t)q * k + rank] = id; out_scores[(size_t)q * k + rank] = sc; }
    }
}

hipError_t cmr_launch_merge_shards(const int64_t* ids, const float* scores, int S, int nq, int k, int64_t* out_ids,
                                   float* out_scores, hipStream_t s) {
    hipLaunchKernelGGL(merge_shards_kernel, dim3(nq), dim3(64), 0, s, ids, scores, S, nq, k, out_ids, out_scores);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Small corpora, few queries (<= 16; ComoRAG's per-question calls see 6 chunks to a few thousand facts / entities, one
// query at a time): the whole search in ONE launch — query packing (fp32 -> MFMA B-operands in LDS, non-finite check),
// the MFMA scan (same block order, hence the same fp32 chains and bit-identical scores, as scan_kernel), selection and
// min / max.  The general path is a chain of 3-5 dependent launches there (pack, [sample scan, merge,] scan, merge):
// 84 us per call at 10 K rows of which the scans are 24.
//   * <= 32 panels (1024 rows): up to four workgroups scan 8 panels each (a wave streams its panels one after the
//     other, 3.6 us each), the LAST one to arrive — release fence, ticket from a self-re-arming counter, acquire fence —
//     selects among all rows (one wave per query, <= 16 keys per lane).
//   * more panels (HIER, up to 256 workgroups, up to the caller's panel limit — 6144 panels = 192 K rows by default —, k <= 64):
//     every workgroup also selects the k best of ITS rows per query (more than 1024 rows: in chunks of 960 with the running
//     k best carried along) and publishes them with its min / max; the last one to arrive selects among the workgroups'
//     candidates, its eight waves sharing that round when there are <= 4 queries.  The k best rows overall are among the k
//     best of their workgroup, so the result is exact.
//   * scores mode (out_full != nullptr: all N raw scores of every query, what dense_passage_retrieval / get_fact_scores
//     consume): scan, then every workgroup copies its rows to the caller's buffer with coalesced stores — no selection.
// (selection by one wave: cmr_select.h — tiny_select / tiny_select_stream)

// geometry of the single-launch search: 0 = not applicable, 1 = flat (<= 32 panels), 2 = hierarchical
struct TinyGeom { int kind, nwg, ppw; size_t off_cand, off_mm, bytes; };
static TinyGeom tiny_geom(int nq, int npanels, int k, bool multi, int max_panels) {
    TinyGeom g{0, 1, 32, 0, 0, 0};
    if (nq > 16 || npanels <= 0 || k <= 0) return g;
    const size_t scores = ((size_t)nq * npanels * CMR_PANEL_ROWS * sizeof(float) + 255) & ~(size_t)255;
    if (npanels <= 32) {
        g.kind = 1;
        g.nwg = (multi && npanels > 8) ? (npanels + 7) / 8 : 1;
        g.ppw = g.nwg == 1 ? 32 : 8;
        g.bytes = scores;
        return g;
    }
    if (!multi || k > 64 || npanels > max_panels) return g;
    // workgroups: as many as there are 8-panel slices (up to 256) — but with >= 5 queries the final round is one wave per
    // query over workgroups x k candidates, serial chunks of 960: keep it to one chunk while that costs <= 4 panels per wave
    int maxwg = std::min(256, (npanels + 7) / 8);
    if (nq > 4 && (npanels + 31) / 32 <= 1024 / k) maxwg = std::min(maxwg, 1024 / k);
    const int ppw = 8 * ((npanels + 8 * maxwg - 1) / (8 * maxwg));
    g.kind = 2;
    g.ppw = ppw;
    g.nwg = (npanels + ppw - 1) / ppw;
    g.off_cand = scores;
    g.off_mm = g.off_cand + (((size_t)g.nwg * nq * k * sizeof(u64) + 255) & ~(size_t)255);
    g.bytes = g.off_mm + (size_t)g.nwg * nq * sizeof(float2);
    return g;
}

template <int DT>
__global__ __launch_bounds__(512) void tiny_search_kernel(const v4u* __restrict__ corpus, const float* __restrict__ q, int nq, int dim, int ks_total,
                                                          int nrows, int npanels, int k, long long id_base, float* __restrict__ scratch,
                                                          int64_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                                          float* __restrict__ out_min, float* __restrict__ out_max, int* __restrict__ flag, int stage_raw,
                                                          int* __restrict__ arrive, int ppw, u64* __restrict__ cand, float2* __restrict__ part_mm,
                                                          float* __restrict__ out_full, long long ld_out, int* __restrict__ done) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    __shared__ int ticket;
    __shared__ u64 tiny_stage[8][128];      // per wave: the selection's surviving keys (two per lane for 64 < k <= 128)
    __shared__ u64 tiny_carry[8][64];       // per wave: the running k best of a chunked selection
    __shared__ u64 tiny_fin[8][64];         // the last workgroup's per-wave pre-selections of the final round
    uint4* qf = reinterpret_cast<uint4*>(sm);                 // [nqt][ks][64], then (stage_raw) the fp32 queries [nq][dim]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nqt = (nq + 31) / 32;
    const bool hier = cand != nullptr;
    bool bad = false;
    // The synchronous host API maps the queries from pinned host memory (no copy in front of the launch): fetch them with
    // independent, unconditional loads — eight in flight per thread, one PCIe round trip for a single query — into LDS
    // and pack from there; packing straight from q would walk the link once per dependent conditional load.
    const float* qsrc = q;
    if (stage_raw) {
        float* qraw = reinterpret_cast<float*>(sm + (size_t)nqt * ks_total * 1024);
        const int nflt = nq * dim;
        for (int i0 = tid; i0 < nflt; i0 += 8 * 512) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * 512; v[u] = q[i < nflt ? i : nflt - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * 512; if (i < nflt) qraw[i] = v[u]; }
        }
        __syncthreads();
        qsrc = qraw;
    }
    for (int i = tid; i < nqt * ks_total * 64; i += 512) {
        const int l = i & 63, ks = (i >> 6) % ks_total, t = (i >> 6) / ks_total;
        const int qi = t * 32 + (l & 31);
        qf[i] = cmr_pack_slot<DT>(qi < nq ? qsrc + (size_t)qi * dim : nullptr, dim, ks, l, bad);
    }
    if (bad) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (every writer writes 1) the flag may live in mapped host memory: visible there before the done word
    __syncthreads();
    const int ld = npanels * CMR_PANEL_ROWS;
    const v4u* qv = reinterpret_cast<const v4u*>(qf);
    const int nwg = gridDim.x;
    const int p_lo = blockIdx.x * ppw, p_hi = p_lo + ppw < npanels ? p_lo + ppw : npanels;      // this workgroup's panels
    for (int p = p_lo + wave; p < p_hi; p += 8) {
        for (int t = 0; t < nqt; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            for (int ks0 = 0; ks0 < ks_total; ks0 += 16) {          // sixteen 1-KiB blocks in flight per wave (ks_total is a multiple of 8)
                v4u buf[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) buf[j] = corpus[((size_t)p * ks_total + (ks0 + j < ks_total ? ks0 + j : ks_total - 1)) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (ks0 + j < ks_total) acc = CmrBlk<DT>::mma(buf[j], qv[(t * ks_total + ks0 + j) * 64 + lane], acc);   // D[row][query], k order
            }
            const int qi = t * 32 + (lane & 31);
            if (qi < nq) {
#pragma unroll
                for (int r = 0; r < 16; ++r) scratch[(size_t)qi * ld + p * CMR_PANEL_ROWS + cmr_acc_row(r, lane)] = acc[r];
            }
        }
    }
    if (out_full) {     // SCORES MODE (cmr_index_scores): this workgroup's rows of every query, coalesced, straight to the caller's buffer
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int row_lo = p_lo * CMR_PANEL_ROWS, row_hi = p_hi * CMR_PANEL_ROWS < nrows ? p_hi * CMR_PANEL_ROWS : nrows;
        if (!done) {
            for (int qi = 0; qi < nq; ++qi)
                for (int row = row_lo + tid; row < row_hi; row += 512) out_full[(size_t)qi * ld_out + row] = scratch[(size_t)qi * ld + row];
            return;
        }
        // a synchronous caller polls the done word of its mapped buffer: the rows leave by system-scope stores (visible to the host once
        // acknowledged), every workgroup is counted behind its own, and the last one to arrive sets the word
        for (int qi = 0; qi < nq; ++qi)
            for (int row = row_lo + tid; row < row_hi; row += 512)
                __hip_atomic_store(&out_full[(size_t)qi * ld_out + row], scratch[(size_t)qi * ld + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int t = nwg > 1 ? atomicAdd(arrive, 1) : 0;
            if (t == nwg - 1) {
                if (nwg > 1) *arrive = 0;                               // re-armed for the next launch
                __hip_atomic_store(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    u64* stage = tiny_stage[wave];
    if (hier) {         // the k best of this workgroup's rows, per query
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int row_lo = p_lo * CMR_PANEL_ROWS, nloc = (p_hi - p_lo) * CMR_PANEL_ROWS;
        for (int qi = wave; qi < nq; qi += 8) {
            float mn = __builtin_inff(), mx = -__builtin_inff();
            u64* dst = cand + ((size_t)blockIdx.x * nq + qi) * k;
            tiny_select_stream(nloc, k, stage, tiny_carry[wave], lane,
                               [&](int i) -> u64 {
                                   const int row = row_lo + i;
                                   if (i >= nloc || row >= nrows) return 0ull;
                                   const float v = scratch[(size_t)qi * ld + row];
                                   mn = fminf(mn, v); mx = fmaxf(mx, v);
                                   return v == v ? cmr_make_key(v, (unsigned)row) : 0ull;
                               },
                               [&](int r, u64 kk) { dst[r] = kk; });
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
            if (lane == 0) part_mm[(size_t)blockIdx.x * nq + qi] = make_float2(mn, mx);
        }
    }
    if (nwg > 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // my scores / candidates are in L2 before my ticket is
        __syncthreads();
        if (tid == 0) ticket = atomicAdd(arrive, 1);
        __syncthreads();
        if (ticket != nwg - 1) return;
        if (tid == 0) *arrive = 0;                                  // everybody has arrived: re-armed for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // the other workgroups' stores, not stale L1 lines
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // Final selection among the workgroups' candidates.  With <= 4 queries the last workgroup's eight waves share it: P = 8,
    // 4 or 2 waves per query pre-select over a P-th of the workgroups each (one chunk instead of up to six serial ones for
    // one wave), then one wave per query selects among the P x k survivors.
    const int fin_parts = (hier && nwg * k > 1024) ? (nq == 1 ? 8 : nq == 2 ? 4 : nq <= 4 ? 2 : 1) : 1;
    if (fin_parts > 1) {
        const int qi = wave / fin_parts, part = wave % fin_parts;
        if (qi < nq) {
            const int b0 = (int)((long long)part * nwg / fin_parts), b1 = (int)((long long)(part + 1) * nwg / fin_parts);
            const int n1 = (b1 - b0) * k;
            u64* dst = tiny_fin[wave];
            tiny_select_stream(n1, k, stage, tiny_carry[wave], lane,
                               [&](int i) -> u64 { return i < n1 ? cand[((size_t)(b0 + i / k) * nq + qi) * k + i % k] : 0ull; },
                               [&](int r, u64 kk) { dst[r] = kk; });
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    for (int qi = wave; qi < nq; qi += 8) {
        float mn = __builtin_inff(), mx = -__builtin_inff();
        // (system-scope stores: a synchronous caller polls the done word of its mapped buffer — plain stores are acknowledged before they are
        // visible to the host, and the word behind s_waitcnt vmcnt(0) would overtake them)
        auto emit = [&](int r, u64 kk) {
            __hip_atomic_store(&out_ids[(size_t)qi * k + r], kk ? (int64_t)cmr_key_row(kk) + id_base : (int64_t)-1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&out_scores[(size_t)qi * k + r], kk ? cmr_key_score(kk) : -__builtin_inff(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        };
        if (!hier) {
            tiny_select_stream(nrows < 1024 ? nrows : 1024, k, stage, tiny_carry[wave], lane,
                               [&](int row) -> u64 {
                                   if (row >= nrows) return 0ull;
                                   const float v = scratch[(size_t)qi * ld + row];
                                   mn = fminf(mn, v); mx = fmaxf(mx, v);
                                   return v == v ? cmr_make_key(v, (unsigned)row) : 0ull;
                               },
                               emit);
        } else {
            for (int b = lane; b < nwg; b += 64) {
                const float2 v = part_mm[(size_t)b * nq + qi];
                mn = fminf(mn, v.x); mx = fmaxf(mx, v.y);
            }
            if (fin_parts > 1) {          // the waves' pre-selections (above), P x k keys in LDS
                const int n2 = fin_parts * k;
                tiny_select_stream(n2, k, stage, tiny_carry[wave], lane,
                                   [&](int i) -> u64 { return i < n2 ? tiny_fin[qi * fin_parts + i / k][i % k] : 0ull; }, emit);
            } else {
                const int ncand = nwg * k;
                tiny_select_stream(ncand, k, stage, tiny_carry[wave], lane,
                                   [&](int i) -> u64 { return i < ncand ? cand[((size_t)(i / k) * nq + qi) * k + i % k] : 0ull; }, emit);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
        if (lane == 0) {
            if (out_min) __hip_atomic_store(&out_min[qi], mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (out_max) __hip_atomic_store(&out_max[qi], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (done) {     // a synchronous caller polls this word (mapped host memory) instead of waiting for the end-of-kernel signal: results first
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int cmr_tiny_kind(int nq, int npanels, int k, int multi, int max_panels) { return tiny_geom(nq, npanels, k, multi != 0, max_panels).kind; }
size_t cmr_tiny_scratch_bytes(int nq, int npanels, int k, int multi, int max_panels) { return tiny_geom(nq, npanels, k, multi != 0, max_panels).bytes; }

static hipError_t tiny_launch(int dtype, const void* corpus, const float* q, int nq, int dim, int dpad, long long nrows, int k, long long id_base,
                              void* scratch, int64_t* out_ids, float* out_scores, float* out_min, float* out_max, int* flag, int* arrive,
                              int max_panels, float* out_full, long long ld_out, hipStream_t s, int* done = nullptr) {
    const int ks = dtype == CMR_DT_F32 ? dpad / 8 : dpad / 16;
    const int npanels = (int)((nrows + CMR_PANEL_ROWS - 1) / CMR_PANEL_ROWS);
    const int nqt = (nq + 31) / 32;
    size_t lds = (size_t)nqt * ks * 1024;
    TinyGeom g = tiny_geom(nq, npanels, k, arrive != nullptr, max_panels);
    if (out_full) {     // scores mode: no selection, as many workgroups as there are 8-panel slices (up to 256)
        if (nq > 16 || npanels <= 0) return hipErrorInvalidValue;
        const int maxwg = std::min(256, (npanels + 7) / 8);
        g.kind = 1;
        g.ppw = 8 * ((npanels + 8 * maxwg - 1) / (8 * maxwg));
        g.nwg = (npanels + g.ppw - 1) / g.ppw;
    }
    constexpr size_t kDynLds = 160 * 1024 - 17 * 1024;                          // the kernel's static LDS (selection stage, carry and final rows) takes 16.3 KiB
    if (!g.kind || lds > kDynLds) return hipErrorInvalidValue;
    // the kernel's selections: kind 1 streams <= 1024 rows, kind 2 up to workgroups x k candidates (> 1024 possible) — cmr_select.h's rule
    if (!out_full && !tiny_select_stream_ok(g.kind == 1 ? (long long)npanels * CMR_PANEL_ROWS : (long long)g.nwg * k + 1025, k)) return hipErrorInvalidValue;
    const int stage_raw = lds + (size_t)nq * dim * 4 <= kDynLds ? 1 : 0;          // fp32 at 1024-d: the operands alone take 128 KiB
    if (stage_raw) lds += (size_t)nq * dim * 4;
    const v4u* c = reinterpret_cast<const v4u*>(corpus);
    float* scores = reinterpret_cast<float*>(scratch);
    u64* cand = g.kind == 2 ? reinterpret_cast<u64*>(reinterpret_cast<char*>(scratch) + g.off_cand) : nullptr;
    float2* pmm = g.kind == 2 ? reinterpret_cast<float2*>(reinterpret_cast<char*>(scratch) + g.off_mm) : nullptr;
#define TS(DT)                                                                                                                       \
    {                                                                                                                                \
        /* a constant: the attribute is per function, not per launch, and host threads launch concurrently with different sizes */              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tiny_search_kernel<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDynLds); \
        if (e != hipSuccess) return e;                                                                                               \
        hipLaunchKernelGGL(tiny_search_kernel<DT>, dim3(g.nwg), dim3(512), lds, s, c, q, nq, dim, ks, (int)nrows, npanels, k, id_base, scores, out_ids, \
                           out_scores, out_min, out_max, flag, stage_raw, arrive, g.ppw, cand, pmm, out_full, ld_out, (out_full && !arrive) ? nullptr : done); \
    }
    switch (dtype) {
        case CMR_DT_BF16: TS(CMR_DT_BF16) break;
        case CMR_DT_F16: TS(CMR_DT_F16) break;
        case CMR_DT_F32: TS(CMR_DT_F32) break;
        default: return hipErrorInvalidValue;
    }
#undef TS
    return hipGetLastError();
}

hipError_t cmr_launch_tiny_search(int dtype, const void* corpus, const float* q, int nq, int dim, int dpad, long long nrows, int k, long long id_base,
                                  void* scratch, int64_t* out_ids, float* out_scores, float* out_min, float* out_max, int* flag, int* arrive,
                                  int max_panels, hipStream_t s, int* done) {
    return tiny_launch(dtype, corpus, q, nq, dim, dpad, nrows, k, id_base, scratch, out_ids, out_scores, out_min, out_max, flag, arrive, max_panels,
                       nullptr, 0, s, done);
}

// all raw scores [nq, ld_out] of a small corpus (nq <= 16) in one launch; scratch: nq * npanels * 32 floats
hipError_t cmr_launch_tiny_scores(int dtype, const void* corpus, const float* q, int nq, int dim, int dpad, long long nrows, void* scratch,
                                  float* out, long long ld_out, int* flag, hipStream_t s, int* arrive, int* done) {
    return tiny_launch(dtype, corpus, q, nq, dim, dpad, nrows, 1, 0, scratch, nullptr, nullptr, nullptr, nullptr, flag, arrive, 1 << 30, out, ld_out, s, arrive ? done : nullptr);
}

// ------------------------------------------------------------------------------------------
// Large-k selection over a materialised score matrix: one workgroup per row of scores [nq, ld]
// picks the k (<= 4096) best (score desc, column asc) — the device half of retrieve_knn's
// torch.topk(k = 2047) (utils/embed_utils.py:56-78) once cmr_index_scores_dev produced the block.
//   1. radix select (4 x 8 bits, MSB first) of the k-th largest order-preserving score code
//   2. ordered stream compaction: every column with a larger code, plus the FIRST (lowest column)
//      `need` columns with the threshold code — exactly k keys, the exported tie rule
//   3. bitonic sort of the k keys in LDS, write ids / scores
#define TOPK_ROWS_MAX 4096
__device__ __forceinline__ unsigned cmr_score_code(float v) {
    unsigned u = __float_as_uint(v + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ scores, long long ld, int n, int k, int kpad,
                                                        long long id_base, int64_t* __restrict__ out_ids,
                                                        float* __restrict__ out_scores, float* __restrict__ out_min,
                                                        float* __restrict__ out_max) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    u64* cand = reinterpret_cast<u64*>(sm);              // kpad
    int* hist = reinterpret_cast<int*>(cand + kpad);     // 256
    int* wsum = hist + 256;                              // 8  (2 x 4 waves)
    int* sel = wsum + 8;                                 // 4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = scores + (size_t)blockIdx.x * ld;
    const int kk = k < n ? k : n;

    unsigned pval = 0, pmask = 0;
    int k_rem = kk;
    if (n > kk) {
        for (int pass = 3; pass >= 0; --pass) {
            const int shift = pass * 8;
            hist[tid] = 0;
            __syncthreads();
            for (int i0 = 0; i0 < n; i0 += 256) {
                const int i = i0 + tid;
                const unsigned code = i < n ? cmr_score_code(row[i]) : 0u;
                const bool act = i < n && (code & pmask) == pval;
                const int dg = (int)((code >> shift) & 255u);
                const u64 am = __ballot(act);
                if (am) {
                    const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)am) - 1);
                    const int d0 = __builtin_amdgcn_readlane(dg, first);
                    if (__ballot(act && dg != d0) == 0) { if (lane == first) atomicAdd(&hist[d0], __popcll(am)); }
                    else if (act) atomicAdd(&hist[dg], 1);
                }
            }
            __syncthreads();
            const int c = hist[tid];
            int s = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_down(s, off); if (lane + off < 64) s += o; }
            if (lane == 0) wsum[wave] = s;
            __syncthreads();
            int higher = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) higher += w > wave ? wsum[w] : 0;
            const int S = s + higher, Sgt = S - c;
            if (S >= k_rem && Sgt < k_rem) { sel[0] = tid; sel[1] = k_rem - Sgt; }
            __syncthreads();
            pval |= (unsigned)sel[0] << shift;
            pmask |= 0xFFu << shift;
            k_rem = sel[1];
        }
    }
    // pval = k-th largest code, k_rem = how many columns with exactly that code are needed (n <= kk: all)
    const unsigned thr = n > kk ? pval : 0u;
    const int need_eq = n > kk ? k_rem : n;       // (when thr = 0 every code is > thr, need_eq unused)
    for (int i = tid; i < kpad; i += 256) cand[i] = 0;
    __syncthreads();
    int base_gt = 0, base_eq = 0;                 // running counts (uniform)
    const int n_gt_total = kk - (n > kk ? need_eq : 0);
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const unsigned code = i < n ? cmr_score_code(row[i]) : 0u;
        const bool gt = i < n && code > thr;
        const bool eq = i < n && n > kk && code == thr;
        // ordered positions inside this tile of 256 columns
        const u64 bg = __ballot(gt), be = __ballot(eq);
        const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int pg = __popcll(bg & below), pe = __popcll(be & below);
        if (lane == 0) { wsum[wave] = __popcll(bg); wsum[4 + wave] = __popcll(be); }
        __syncthreads();
        int og = 0, oe = 0, tg = 0, te = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { og += w < wave ? wsum[w] : 0; oe += w < wave ? wsum[4 + w] : 0; tg += wsum[w]; te += wsum[4 + w]; }
        const u64 key = ((u64)code << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
        if (gt) cand[base_gt + og + pg] = key;                                  // all of them: exactly n_gt_total overall
        if (eq) { const int slot = base_eq + oe + pe; if (slot < need_eq) cand[n_gt_total + slot] = key; }
        base_gt += tg; base_eq += te;
        __syncthreads();
    }
    // bitonic sort, descending, kpad = pow2 >= kk (zeros = empty sink to the end)
    for (int size = 2; size <= kpad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < kpad / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const u64 a = cand[lo], b = cand[hi];
                if ((a < b) == desc) { cand[lo] = b; cand[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 256) {
        const u64 key = i < kk ? cand[i] : 0ull;
        out_ids[(size_t)blockIdx.x * k + i] = key ? (int64_t)cmr_key_row(key) + id_base : -1;
        out_scores[(size_t)blockIdx.x * k + i] = key ? cmr_key_score(key) : -__builtin_inff();
    }
    if (out_min || out_max) {
        float mn = __builtin_inff(), mx = -__builtin_inff();
        for (int i = tid; i < n; i += 256) { const float v = row[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
        float* red = reinterpret_cast<float*>(wsum);
        __syncthreads();
        if (lane == 0) { red[wave] = mn; red[4 + wave] = mx; }
        __syncthreads();
        if (tid == 0) {
            if (out_min) out_min[blockIdx.x] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
            if (out_max) out_max[blockIdx.x] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        }
    }
}

hipError_t cmr_launch_topk_rows(const float* scores, long long ld, int n, int nq, int k, long long id_base,
                                int64_t* out_ids, float* out_scores, float* out_min, float* out_max, hipStream_t s) {
    if (k <= 0 || k > TOPK_ROWS_MAX) return hipErrorInvalidValue;
    int kpad = 2;
    while (kpad < (k < n ? k : n)) kpad <<= 1;
    const size_t lds = (size_t)kpad * 8 + 256 * 4 + 8 * 4 + 4 * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(topk_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);     // a constant: per function, not per launch
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(topk_rows_kernel, dim3(nq), dim3(256), lds, s, scores, ld, n, k, kpad, id_base, out_ids, out_scores,
                       out_min, out_max);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Full descending sort of one query's N scores (ComoRAG.dense_passage_retrieval returns ALL N ids
// by descending score, ComoRAG.py:964-966; graph_search_with_fact_entities consumes every pair,
// :1034-1042).  Stable LSD radix sort, 4-bit digits, 8 passes over the complemented order-preserving
// score code with the row index as payload: starting from ascending rows, stability yields exactly
// the exported tie rule (score desc, row asc).  Per pass: per-tile digit histogram -> one exclusive
// scan over [digit][tile] -> stable scatter (ballot ranks inside a wave, waves and 256-key slices
// in order).
#define SORT_TILE 2048
#define SORT_THREADS 256

__global__ __launch_bounds__(256) void sort_init_kernel(const float* __restrict__ scores, unsigned n, unsigned* __restrict__ code,
                                                        unsigned* __restrict__ val) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) { code[i] = ~cmr_score_code(scores[i]); val[i] = i; }
}

__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const unsigned* __restrict__ code, unsigned n, int shift,
                                                                  unsigned ntiles, unsigned* __restrict__ hist) {
    __shared__ unsigned cnt[16];
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
    __syncthreads();
    const unsigned base = blockIdx.x * SORT_TILE;
    for (unsigned j = threadIdx.x; j < SORT_TILE; j += SORT_THREADS) {
        const unsigned i = base + j;
        if (i < n) atomicAdd(&cnt[(code[i] >> shift) & 15u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 16) hist[threadIdx.x * ntiles + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive scan of m = 16*ntiles counters, one workgroup (m is ~8 K at 1 M rows, ~78 K at 10 M)
__global__ __launch_bounds__(1024) void sort_scan_kernel(unsigned* __restrict__ hist, unsigned m) {
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry_s;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (unsigned base = 0; base < m; base += 1024 * 8) {
        unsigned v[8], local = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const unsigned i = base + tid * 8 + j; v[j] = i < m ? hist[i] : 0u; local += v[j]; }
        unsigned incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned wbase = 0;
        for (unsigned w = 0; w < wave; ++w) wbase += wsum[w];
        unsigned run = carry_s + wbase + incl - local;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const unsigned i = base + tid * 8 + j; if (i < m) hist[i] = run; run += v[j]; }
        __syncthreads();
        if (tid == 1023) carry_s = run;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const unsigned* __restrict__ code_in, const unsigned* __restrict__ val_in,
                                                                     unsigned n, int shift, unsigned ntiles,
                                                                     const unsigned* __restrict__ offs, unsigned* __restrict__ code_out,
                                                                     unsigned* __restrict__ val_out) {
    __shared__ unsigned run[16];          // keys of this tile already placed, per digit
    __shared__ unsigned wcnt[4][16];      // per-wave digit counts of the current 256-key slice
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 16) run[tid] = offs[tid * ntiles + blockIdx.x];
    __syncthreads();
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (unsigned s = 0; s < SORT_TILE; s += SORT_THREADS) {
        const unsigned i = blockIdx.x * SORT_TILE + s + tid;
        const bool ok = i < n;
        const unsigned c = ok ? code_in[i] : 0u;
        const unsigned d = ok ? (c >> shift) & 15u : 16u;
        unsigned rank = 0;
#pragma unroll
        for (unsigned v = 0; v < 16; ++v) {
            const u64 b = __ballot(d == v);
            if (d == v) rank = (unsigned)__popcll(b & lt);
            if (lane == 0) wcnt[wave][v] = (unsigned)__popcll(b);
        }
        __syncthreads();
        if (ok) {
            unsigned pos = run[d] + rank;
            for (unsigned w = 0; w < wave; ++w) pos += wcnt[w][d];
            code_out[pos] = c;
            val_out[pos] = val_in[i];
        }
        __syncthreads();
        if (tid < 16) run[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void sort_finish_kernel(const unsigned* __restrict__ val, const float* __restrict__ scores, unsigned n,
                                                          long long id_base, int64_t* __restrict__ out_ids,
                                                          float* __restrict__ out_scores) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) { const unsigned r = val[i]; out_ids[i] = (int64_t)r + id_base; out_scores[i] = scores[r]; }
}

size_t cmr_sort_workspace_bytes(long long n) {
    const size_t ntiles = (size_t)((n + SORT_TILE - 1) / SORT_TILE);
    return (size_t)4 * n * 4 + 16 * ntiles * 4 + 256;
}

// scores [n] (device) -> out_ids [n], out_scores [n] sorted descending (ties: ascending row)
hipError_t cmr_launch_sort_scores(const float* scores, long long n, long long id_base, void* workspace, int64_t* out_ids,
                                  float* out_scores, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n >= (1ll << 32)) return hipErrorInvalidValue;
    const unsigned un = (unsigned)n;
    const unsigned ntiles = (un + SORT_TILE - 1) / SORT_TILE;
    unsigned* code0 = reinterpret_cast<unsigned*>(workspace);
    unsigned* val0 = code0 + n;
    unsigned* code1 = val0 + n;
    unsigned* val1 = code1 + n;
    unsigned* hist = val1 + n;
    const unsigned g256 = (un + 255) / 256;
    hipLaunchKernelGGL(sort_init_kernel, dim3(g256), dim3(256), 0, s, scores, un, code0, val0);
    unsigned *ci = code0, *vi = val0, *co = code1, *vo = val1;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 4 * pass;
        hipLaunchKernelGGL(sort_hist_kernel, dim3(ntiles), dim3(SORT_THREADS), 0, s, ci, un, shift, ntiles, hist);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, s, hist, 16u * ntiles);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(ntiles), dim3(SORT_THREADS), 0, s, ci, vi, un, shift, ntiles, hist, co, vo);
        unsigned* t = ci; ci = co; co = t;
        t = vi; vi = vo; vo = t;
    }
    hipLaunchKernelGGL(sort_finish_kernel, dim3(g256), dim3(256), 0, s, vi, scores, un, id_base, out_ids, out_scores);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Shard-local row -> global id through the shard's block table (a row shard that took incremental appends holds several
// runs of consecutive global ids: block b = local rows [local0[b], local0[b+1]) = global ids global0[b] + ...).
// tab = [local0[0..nb) | global0[0..nb)], local0 ascending, local0[0] = 0.  Negative ids (empty slots) pass through.
__global__ __launch_bounds__(256) void remap_ids_kernel(int64_t* __restrict__ ids, long long n, const long long* __restrict__ tab, int nb) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t r = ids[i];
    if (r < 0) return;
    int lo = 0, hi = nb - 1;                 // last block with local0 <= r
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid] <= r) lo = mid; else hi = mid - 1;
    }
    ids[i] = r - tab[lo] + tab[nb + lo];
}
hipError_t cmr_launch_remap_ids(int64_t* ids, long long n, const long long* tab, int nb, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(remap_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ids, n, tab, nb);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Row access in the panel-major layout.
template <int DT>
__device__ __forceinline__ float cmr_load_elem(const unsigned char* corpus, int ks_total, long long row, int kidx) {
    const long long panel = row / CMR_PANEL_ROWS;
    const int r = (int)(row % CMR_PANEL_ROWS);
    if (DT == CMR_DT_F32) {
        const int ks = kidx >> 3, w = kidx & 7, e = w >> 1, h = w & 1;
        const size_t off = (((size_t)panel * ks_total + ks) * 64 + h * 32 + r) * 16 + e * 4;
        return *reinterpret_cast<const float*>(corpus + off);
    } else {
        const int ks = kidx >> 4, w = kidx & 15, h = w >> 3, e = w & 7;
        const size_t off = (((size_t)panel * ks_total + ks) * 64 + h * 32 + r) * 16 + e * 2;
        const unsigned short v = *reinterpret_cast<const unsigned short*>(corpus + off);
        return DT == CMR_DT_BF16 ? cmr_bf2f(v) : cmr_h2f(v);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const unsigned char* __restrict__ corpus, int dim, int ks_total,
                                                          long long nrows, long long id_base, const int64_t* __restrict__ ids, long long n,
                                                          float* __restrict__ out) {
    const long long i = blockIdx.x;
    const int64_t row = ids[i] - id_base;
    for (int kx = threadIdx.x; kx < dim; kx += 256)
        out[(size_t)i * dim + kx] = (row >= 0 && row < nrows) ? cmr_load_elem<DT>(corpus, ks_total, row, kx) : 0.0f;
}

hipError_t cmr_launch_gather_rows(int dtype, const void* corpus, int dim, int dpad, long long nrows, long long id_base,
                                  const int64_t* ids, long long n, float* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int ks = dtype == CMR_DT_F32 ? dpad / 8 : dpad / 16;
    const unsigned char* c = reinterpret_cast<const unsigned char*>(corpus);
    dim3 grid((unsigned)n), block(256);
    switch (dtype) {
        case CMR_DT_BF16: hipLaunchKernelGGL(gather_rows_kernel<CMR_DT_BF16>, grid, block, 0, s, c, dim, ks, nrows, id_base, ids, n, out); break;
        case CMR_DT_F16:  hipLaunchKernelGGL(gather_rows_kernel<CMR_DT_F16>, grid, block, 0, s, c, dim, ks, nrows, id_base, ids, n, out); break;
        case CMR_DT_F32:  hipLaunchKernelGGL(gather_rows_kernel<CMR_DT_F32>, grid, block, 0, s, c, dim, ks, nrows, id_base, ids, n, out); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Exact fp32 re-score: one workgroup per query; each wave takes candidates round-robin, the dot
// is a fixed-order per-lane fmaf chain + butterfly (deterministic), then block top-k on keys
// whose "row" field is the candidate row id.
template <int DT>
__global__ __launch_bounds__(256) void rescore_kernel(const unsigned char* __restrict__ corpus, const float* __restrict__ shadow,
                                                      int dim, int ks_total, long long nrows, long long id_base, const float* __restrict__ q,
                                                      const int64_t* __restrict__ cand, int n_cand, int k,
                                                      int64_t* __restrict__ out_ids, float* __restrict__ out_scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    u64* pool = reinterpret_cast<u64*>(sm);      // n_cand
    u64* wbest = pool + n_cand;                  // 4
    u64* res = wbest + 4;                        // k
    const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qv = q + (size_t)qi * dim;
    for (int c = wave; c < n_cand; c += 4) {
        const int64_t row = cand[(size_t)qi * n_cand + c] - id_base;      // candidates carry global ids
        u64 key = 0;
        if (row >= 0 && row < nrows) {
            float acc = 0.0f;
            for (int kx = lane; kx < dim; kx += 64) {
                const float x = shadow ? shadow[(size_t)row * dim + kx] : cmr_load_elem<DT>(corpus, ks_total, row, kx);
                acc = fmaf(x, qv[kx], acc);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            key = cmr_make_key(acc, (unsigned)row);
        }
        if (lane == 0) pool[c] = key;
    }
    __syncthreads();
    // duplicates in cand would break uniqueness: keep the first occurrence only
    for (int c = tid; c < n_cand; c += 256) {
        const u64 key = pool[c];
        if (!key) continue;
        for (int o = 0; o < c; ++o) if (pool[o] == key) { pool[c] = 0; break; }
    }
    __syncthreads();
    cmr_block_select(pool, n_cand, k, res, wbest);
    for (int i = tid; i < k; i += 256) {
        const u64 key = res[i];
        out_ids[(size_t)qi * k + i] = key ? (int64_t)cmr_key_row(key) + id_base : -1;
        out_scores[(size_t)qi * k + i] = key ? cmr_key_score(key) : -__builtin_inff();
    }
}

hipError_t cmr_launch_rescore(int dtype, const void* corpus, const float* shadow, int dim, int dpad, long long nrows, long long id_base,
                              const float* q, int nq, const int64_t* cand, int n_cand, int k, int64_t* out_ids,
                              float* out_scores, hipStream_t s) {
    const int ks = dtype == CMR_DT_F32 ? dpad / 8 : dpad / 16;
    const unsigned char* c = reinterpret_cast<const unsigned char*>(corpus);
    const size_t lds = ((size_t)n_cand + 4 + k) * 8;
    dim3 grid(nq), block(256);
#define RS(DT)                                                                                                        \
    {                                                                                                                 \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rescore_kernel<DT>),                         \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);                    \
        if (e != hipSuccess) return e;                                                                                \
        hipLaunchKernelGGL(rescore_kernel<DT>, grid, block, lds, s, c, shadow, dim, ks, nrows, id_base, q, cand, n_cand, k, \
                           out_ids, out_scores);                                                                      \
    }
    switch (dtype) {
        case CMR_DT_BF16: RS(CMR_DT_BF16) break;
        case CMR_DT_F16: RS(CMR_DT_F16) break;
        case CMR_DT_F32: RS(CMR_DT_F32) break;
        default: return hipErrorInvalidValue;
    }
#undef RS
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Encoder tail.  hidden [b, l, d] -> masked sum over l (split over `splits` workgroups per
// batch row, fixed order inside each, partial[b][split][d]) -> finalize: sum partials in split
// order, divide by the token count, L2-normalise (eps 1e-12, as F.normalize).
template <typename T> __device__ __forceinline__ float cmr_to_f(T v);
template <> __device__ __forceinline__ float cmr_to_f<float>(float v) { return v; }
struct bf16_t { unsigned short v; };
struct f16_t { unsigned short v; };
template <> __device__ __forceinline__ float cmr_to_f<bf16_t>(bf16_t v) { return cmr_bf2f(v.v); }
template <> __device__ __forceinline__ float cmr_to_f<f16_t>(f16_t v) { return cmr_h2f(v.v); }

// Latency, not bandwidth, was the first bound here (34 us for 40 MB at b = 32: each token row was a mask load, a
// branch and a dependent 16-B load).  Both kernels now issue their loads in batches of POOL_U before the first add;
// the adds keep the token / split order, so the sums are bit-identical to the sequential loop.
#define POOL_U 8
template <typename T>
__global__ __launch_bounds__(256) void pool_partial_kernel(const T* __restrict__ hidden, const int64_t* __restrict__ mask, int l,
                                                           int d, int splits, float* __restrict__ partial) {
    const int b = blockIdx.x, sp = blockIdx.y;
    const int per = (l + splits - 1) / splits;
    const int t0 = sp * per, t1 = min(l, t0 + per);
    const int64_t* mrow = mask + (size_t)b * l;
    // attention mask of this split as bit sets, one coalesced load per 64 tokens, taken before any lane drops out
    // of the column loop (host keeps `per` <= 256, cmr_pool_splits; longer splits use the per-token loads)
    const int lane = threadIdx.x & 63;
    const bool bitmask = per <= 256;
    u64 bits[4] = {0ull, 0ull, 0ull, 0ull};
    if (bitmask) {
#pragma unroll
        for (int cidx = 0; cidx < 4; ++cidx) {
            const int tt = t0 + cidx * 64 + lane;
            bits[cidx] = __ballot(tt < t1 && mrow[tt < l ? tt : 0] != 0);
        }
    }
    // each thread owns columns tid*VEC .. (+VEC) strided by 256*VEC: loads are contiguous per token row
    constexpr int VEC = 16 / sizeof(T);
    for (int c0 = threadIdx.x * VEC; c0 < d; c0 += 256 * VEC) {
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.0f;
        const bool full = c0 + VEC <= d && (d % VEC) == 0;
        if (full && bitmask) {
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) {
                const u64 bs = bits[cidx];
                if (bs == 0ull) continue;
                const int tc = t0 + cidx * 64;
                for (int j0 = 0; j0 < 64 && (bs >> j0) != 0ull; j0 += POOL_U) {
                    uint4 raw[POOL_U];
#pragma unroll
                    for (int j = 0; j < POOL_U; ++j)
                        raw[j] = ((bs >> (j0 + j)) & 1ull)
                                     ? *reinterpret_cast<const uint4*>(hidden + ((size_t)b * l + tc + j0 + j) * d + c0)
                                     : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                    for (int j = 0; j < POOL_U; ++j) {
                        if (!((bs >> (j0 + j)) & 1ull)) continue;
                        const T* e4 = reinterpret_cast<const T*>(&raw[j]);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[e] += cmr_to_f<T>(e4[e]);
                    }
                }
            }
        } else {
            for (int t = t0; t < t1; ++t) {
                if (mrow[t] == 0) continue;
                const T* rowp = hidden + ((size_t)b * l + t) * d + c0;
                for (int e = 0; e < VEC; ++e) if (c0 + e < d) acc[e] += cmr_to_f<T>(rowp[e]);
            }
        }
        for (int e = 0; e < VEC; ++e)
            if (c0 + e < d) partial[((size_t)b * splits + sp) * d + c0 + e] = acc[e];
    }
}

__global__ __launch_bounds__(256) void pool_finalize_kernel(const float* __restrict__ partial, const int64_t* __restrict__ mask,
                                                            int l, int d, int splits, int normalize, float* __restrict__ out) {
    __shared__ float red[4];
    __shared__ float cnt_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // token count
    float c = 0.0f;
    for (int t = tid; t < l; t += 256) c += mask[(size_t)b * l + t] != 0 ? 1.0f : 0.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0) red[wave] = c;
    __syncthreads();
    if (tid == 0) cnt_s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    const float count = cnt_s;
    float ss = 0.0f;
    for (int cx = tid; cx < d; cx += 256) {
        float s = 0.0f;
        for (int sp = 0; sp < splits; sp += POOL_U) {
            float v[POOL_U];
#pragma unroll
            for (int j = 0; j < POOL_U; ++j) v[j] = sp + j < splits ? partial[((size_t)b * splits + sp + j) * d + cx] : 0.0f;
#pragma unroll
            for (int j = 0; j < POOL_U; ++j) if (sp + j < splits) s += v[j];
        }
        s = s / count;
        out[(size_t)b * d + cx] = s;
        ss += s * s;
    }
    if (!normalize) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    __syncthreads();
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
    for (int cx = tid; cx < d; cx += 256) out[(size_t)b * d + cx] = out[(size_t)b * d + cx] / nrm;
}

int cmr_pool_splits(int b, int l, int d) {
    // enough workgroups to cover the chip (>= ~1024) without making partials dominate traffic
    int s = (1024 + b - 1) / b;
    if (s > l / 8) s = l / 8;
    if (s < 1) s = 1;
    if (s < (l + 255) / 256) s = (l + 255) / 256;     // <= 256 tokens per split: the kernel's bit-set fast path
    if (s > 64) s = 64;
    (void)d;
    return s;
}

hipError_t cmr_launch_pool(const void* hidden, int hidden_dtype, const int64_t* mask, int b, int l, int d, int normalize,
                           float* partial, float* out, int splits, hipStream_t s) {
    dim3 grid(b, splits), block(256);
    switch (hidden_dtype) {
        case CMR_DT_F32: hipLaunchKernelGGL(pool_partial_kernel<float>, grid, block, 0, s, reinterpret_cast<const float*>(hidden), mask, l, d, splits, partial); break;
        case CMR_DT_BF16: hipLaunchKernelGGL(pool_partial_kernel<bf16_t>, grid, block, 0, s, reinterpret_cast<const bf16_t*>(hidden), mask, l, d, splits, partial); break;
        case CMR_DT_F16: hipLaunchKernelGGL(pool_partial_kernel<f16_t>, grid, block, 0, s, reinterpret_cast<const f16_t*>(hidden), mask, l, d, splits, partial); break;
        default: return hipErrorInvalidValue;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pool_finalize_kernel, dim3(b), dim3(256), 0, s, partial, mask, l, d, splits, normalize, out);
    return hipGetLastError();
}
