// Corpus scan for gfx950 (MI355X): query x corpus inner products on MFMA with the per-query
// running top-k (and global min/max) fused into the epilogue, so the corpus is read from HBM
// exactly once per query batch and no score matrix is ever written.
//
// Replaces np.dot(M, q.T) + min/max + argsort of ComoRAG.dense_passage_retrieval
// (src/comorag/ComoRAG.py:950-967), get_fact_scores (:937-948) and the torch.mm + torch.topk
// blocks of retrieve_knn (src/comorag/utils/embed_utils.py:52-78).
//
// Shape of the kernel:
//  * corpus lives in HBM as panels of 32 rows, each panel a run of 1-KiB blocks that ARE the
//    A-operand of v_mfma_f32_32x32x16_{bf16,f16} (or 4x v_mfma_f32_32x32x2_f32) in lane order,
//    so one wave reads one block with one perfectly coalesced global_load_dwordx4 — no LDS
//    staging and no layout shuffle for the streamed operand;
//  * the query batch (<= 64 queries) is the B-operand, pre-packed in the same block order and
//    held in LDS for the whole kernel (96 KiB at 64 x 768 bf16), read with conflict-free
//    ds_read_b128;
//  * every wave owns a contiguous range of panels and streams it through a register ring of R
//    blocks (R KiB in flight per wave, 8 waves per workgroup) — waves never synchronise;
//  * after a panel's MFMAs a lane holds 16 scores of ONE query (C/D layout: col = lane & 31):
//    it folds them into the query's running min/max and compares the panel max with the
//    query's threshold tau (the k-th best key this wave has seen).  Only on a hit does the wave
//    take the slow path: push (score,row) keys into the (wave,query) list in global scratch and,
//    when a list is nearly full, compact it to its k best (rank by counting) and raise tau.
//    Expected pushes per query per wave are k*ln(rows/k): the slow path is rare by construction.
#include "cmr_device.h"
#include "cmr_kernels.h"

#define MODE_TOPK 0
#define MODE_SCORES 1

struct ScanP {
    const v4u* corpus;
    const v4u* qfrag;
    long long nrows;
    int npanels;
    int ks;
    int k;
    u64* lists;
    int* cnt;
    float2* mm;
    float* scores;
    long long ld;
    int nq;
    int sample_waves;      // > 0: sampling pass — wave w (< sample_waves) scans the single panel w*sample_stride
    int sample_stride;
    const u64* tau_init;   // per-query initial threshold keys (from the sampling pass) or nullptr
};

template <int CAP>
__device__ __forceinline__ void topk_slow_path(const f32x16& acc, long long row0, long long nrows, int k,
                                               u64& tau_key, float& tau_f, int* cnt_t, u64* list_t,
                                               u64* stage, int lane) {
    constexpr int EPL = CAP / 64;
    const int ql = lane & 31;
    const unsigned hrow = 4u * (unsigned)(lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = acc[r];
        const long long row = row0 + (r & 3) + 8 * (r >> 2) + hrow;
        const u64 key = cmr_make_key(v, (unsigned)row);
        if (row < nrows && v == v && key > tau_key) {
            const int slot = atomicAdd(&cnt_t[ql], 1);  // ds_add_rtn_u32; <= 32 pushes per query per panel
            list_t[(size_t)ql * CAP + slot] = key;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // pushes landed (same-CU L1 is coherent)
    const int c = __hip_atomic_load(&cnt_t[ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    u64 need = __ballot(c > CAP - 32) & 0xFFFFFFFFull;
    while (need) {
        const int j = __ffsll((long long)need) - 1;
        need &= need - 1;
        const int n = __builtin_amdgcn_readfirstlane(
            __hip_atomic_load(&cnt_t[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        u64* L = list_t + (size_t)j * CAP;
        u64 e[EPL];
        int rk[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int idx = lane + 64 * i;
            e[i] = idx < n ? L[idx] : 0ull;
            stage[idx] = e[i];
            rk[i] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int jj = 0; jj < n; ++jj) {
            const u64 kj = stage[jj];  // uniform address: LDS broadcast
#pragma unroll
            for (int i = 0; i < EPL; ++i) rk[i] += (kj > e[i]) ? 1 : 0;
        }
        // keys are unique => ranks are a permutation of 0..n-1; keep the k best, in order
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int idx = lane + 64 * i;
            if (idx < n && rk[i] < k) L[rk[i]] = e[i];
            if (idx < n && rk[i] == k - 1) stage[CAP] = e[i];
        }
        if (lane == 0) __hip_atomic_store(&cnt_t[j], n < k ? n : k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (n >= k) {
            const u64 nt = stage[CAP];
            if (ql == j) { tau_key = nt; tau_f = cmr_key_score(nt); }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

template <int DT, int NQT, int CAP, int R, int MODE, int ASMRING>
__global__ __launch_bounds__(CMR_SCAN_THREADS, 2) void scan_kernel(ScanP P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (SGPR)
    const int KS = P.ks;
    constexpr int NQ = NQT * 32;

    v4u* qf = reinterpret_cast<v4u*>(smem);
    int* cnt_all = reinterpret_cast<int*>(smem + (size_t)NQT * KS * 1024);
    u64* stage_all = reinterpret_cast<u64*>(cnt_all + CMR_SCAN_WAVES * NQ);
    int* cnt_w = cnt_all + wave * NQ;
    u64* stage = stage_all + wave * (CAP + 2);

    for (int i = tid; i < NQT * KS * 64; i += CMR_SCAN_THREADS) qf[i] = P.qfrag[i];
    if (MODE == MODE_TOPK)
        for (int i = tid; i < CMR_SCAN_WAVES * NQ; i += CMR_SCAN_THREADS) cnt_all[i] = 0;
    __syncthreads();

    const int W = gridDim.x * CMR_SCAN_WAVES;
    const int gw = blockIdx.x * CMR_SCAN_WAVES + wave;
    int p0, p1;
    if (P.sample_waves > 0) {   // sampling pass: one strided panel per wave
        p0 = gw * P.sample_stride;
        p1 = gw < P.sample_waves ? p0 + 1 : p0;
    } else {
        p0 = (int)(((long long)gw * P.npanels) / W);
        p1 = (int)(((long long)(gw + 1) * P.npanels) / W);
    }

    float rmin[NQT], rmax[NQT], tau_f[NQT];
    u64 tau_key[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        rmin[t] = __builtin_inff();
        rmax[t] = -__builtin_inff();
        tau_f[t] = -__builtin_inff();
        tau_key[t] = 0ull;
        if (MODE == MODE_TOPK) {
            const int q = t * 32 + (lane & 31);
            if (q >= P.nq) {                     // padding query (all-zero operand): nothing may pass
                tau_key[t] = ~0ull;
                tau_f[t] = __builtin_inff();
            } else if (P.tau_init) {             // a valid lower bound on the global k-th best key
                tau_key[t] = P.tau_init[q];
                if (tau_key[t]) tau_f[t] = cmr_key_score(tau_key[t]);
            }
        }
    }
    u64* list_w = (MODE == MODE_TOPK) ? P.lists + (size_t)gw * NQ * CAP : nullptr;

    if (p1 > p0) {
        // The ring always prefetches R blocks ahead, also past the end of this wave's range: the
        // corpus allocation carries CMR_CORPUS_SLACK bytes of tail slack so the over-read is legal.
        //
        // ASMRING = 0: plain loads; hipcc counts them but drains the ring (vmcnt(0)) at the top of
        //              every group of R blocks.
        // ASMRING = 1: the ring loads are inline asm (invisible to hipcc's vmcnt bookkeeping) with a
        //              hand-counted s_waitcnt vmcnt(R-1) in front of each block's MFMAs, so exactly R
        //              blocks stay in flight per wave at all times.  Each slot is a tied "+v" operand
        //              of both the wait and the reload: it never changes register, so no compiler copy
        //              can read it before the data lands (guide §5.7 item 1).  Compiler-issued VMEM ops
        //              (slow path) only make either side's waits stricter: loads return in order.
        const v4u* sbase = P.corpus + (size_t)p0 * KS * 64;   // wave-uniform (SGPR pair)
        v4u buf[R];
        unsigned voff[R / 4];
#pragma unroll
        for (int j = 0; j < R / 4; ++j) voff[j] = (unsigned)lane * 16u + (unsigned)j * 4096u;
        const v4u* src = sbase + lane;

#define CMR_RING_LOAD(u)                                                                                   \
    if constexpr ((u) < R) {                                                                               \
        if constexpr (ASMRING) {                                                                           \
            v4u slot_ = buf[(u)];                                                                        \
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"                                        \
                         : "+v"(slot_) : "v"(voff[(u) >> 2]), "s"(sbase), "n"(((u) & 3) * 1024) : "memory"); \
            buf[(u)] = slot_;                                                                              \
        } else {                                                                                           \
            buf[(u)] = src[(size_t)(u) * 64];                                                              \
        }                                                                                                  \
    }
#define CMR_RING_STEP(u)                                                                                   \
    if constexpr ((u) < R) {                                                                               \
        if constexpr (ASMRING) {                                                                           \
            v4u slot_ = buf[(u)];                                                                        \
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(slot_) : "n"(R - 1));                                \
            buf[(u)] = slot_;                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int t = 0; t < NQT; ++t) {                                                  \
            const v4u b = qg[(t * KS + (u)) * 64];                                                       \
            if (MODE == MODE_TOPK) acc[t] = CmrBlk<DT>::mma(buf[(u)], b, acc[t]); /* D[row][query] */      \
            else                   acc[t] = CmrBlk<DT>::mma(b, buf[(u)], acc[t]); /* D[query][row] */      \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0); /* refill only after the slot's MFMAs were issued */            \
        CMR_RING_LOAD(u)                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
#define CMR_RING_ALL(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

#pragma unroll
        for (int u = 0; u < R; ++u) buf[u] = (v4u){0u, 0u, 0u, 0u};

        f32x16 acc[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

        // ONE loop body holds every ring statement, so a slot's value is only ever a loop-carried
        // register: iteration -1 is the priming pass (waits pass trivially, the MFMAs add 0*q to
        // the zero accumulators, the reloads fetch group 0); iteration `it` computes group `it`.
        const int GPP = KS / R;                              // groups per panel
        const long long ngroups = (long long)(p1 - p0) * GPP;
        int gi = 0;
        int p = p0;
        for (long long it = -1; it < ngroups; ++it) {
            {
                const v4u* qg = qf + (size_t)gi * R * 64 + lane;
                CMR_RING_ALL(CMR_RING_STEP)
                src += (size_t)R * 64;
#pragma unroll
                for (int j = 0; j < R / 4; ++j) voff[j] += (unsigned)R * 1024u;
            }
            if (it < 0) continue;
            if (++gi < GPP) continue;
            gi = 0;

            const long long row0 = (long long)p * CMR_PANEL_ROWS;
            if (MODE == MODE_TOPK) {
                const bool partial = row0 + CMR_PANEL_ROWS > P.nrows;
#pragma unroll
                for (int t = 0; t < NQT; ++t) {
                    float mx, mn;
                    if (!partial) {
                        mx = acc[t][0]; mn = acc[t][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) { mx = fmaxf(mx, acc[t][r]); mn = fminf(mn, acc[t][r]); }
                    } else {
                        mx = -__builtin_inff(); mn = __builtin_inff();
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const bool ok = row0 + cmr_acc_row(r, lane) < P.nrows;
                            mx = fmaxf(mx, ok ? acc[t][r] : -__builtin_inff());
                            mn = fminf(mn, ok ? acc[t][r] : __builtin_inff());
                        }
                    }
                    rmax[t] = fmaxf(rmax[t], mx);
                    rmin[t] = fminf(rmin[t], mn);
                    if (__any(mx >= tau_f[t]))
                        topk_slow_path<CAP>(acc[t], row0, P.nrows, P.k, tau_key[t], tau_f[t], cnt_w + t * 32,
                                            list_w + (size_t)t * 32 * CAP, stage, lane);
                }
            } else {
                const long long row = row0 + (lane & 31);
                if (row < P.nrows) {
#pragma unroll
                    for (int t = 0; t < NQT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int qi = t * 32 + cmr_acc_row(r, lane);
                            if (qi < P.nq) P.scores[(size_t)qi * P.ld + row] = acc[t][r];
                        }
                }
            }
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            ++p;
        }
    }

    if constexpr (MODE == MODE_TOPK) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            const float mn = fminf(rmin[t], __shfl_xor(rmin[t], 32));
            const float mx = fmaxf(rmax[t], __shfl_xor(rmax[t], 32));
            if (lane < 32) {
                const int q = t * 32 + lane;
                P.mm[(size_t)gw * NQ + q] = make_float2(mn, mx);
                P.cnt[(size_t)gw * NQ + q] = __hip_atomic_load(&cnt_w[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ host
static size_t scan_lds_bytes(int nqt, int ks, int cap) {
    return (size_t)nqt * ks * 1024 + (size_t)CMR_SCAN_WAVES * nqt * 32 * 4 + (size_t)CMR_SCAN_WAVES * (cap + 2) * 8;
}
static constexpr size_t kLdsLimit = 160 * 1024;

int cmr_scan_max_nqt(int dtype, int dpad) {
    const int ks = dtype == CMR_DT_F32 ? dpad / 8 : dpad / 16;
    for (int nqt = 2; nqt >= 1; --nqt)
        if (scan_lds_bytes(nqt, ks, 256) <= kLdsLimit) return nqt;
    return 0;
}

bool cmr_scan_geom(CmrScanGeom* g) {
    g->ks = g->dtype == CMR_DT_F32 ? g->dpad / 8 : g->dpad / 16;
    g->lds = scan_lds_bytes(g->nqt, g->ks, g->cap);
    if (g->ring != 8 && g->ring != 16) return false;
    if (g->ks % g->ring) return false;
    return g->lds <= kLdsLimit;
}

template <int DT, int NQT, int CAP, int R, int MODE>
static hipError_t launch_one(const CmrScanGeom& g, const ScanP& p, hipStream_t s) {
    // occupancy follows from LDS: <= 80 KiB of query fragments -> two workgroups per CU
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(g.grid), dim3(CMR_SCAN_THREADS), g.lds, s, p);
        return hipGetLastError();
    };
    if (g.asm_ring) return launch(scan_kernel<DT, NQT, CAP, R, MODE, 1>);
    return launch(scan_kernel<DT, NQT, CAP, R, MODE, 0>);
}

template <int DT, int MODE>
static hipError_t dispatch(const CmrScanGeom& g, const ScanP& p, hipStream_t s) {
#define CASE(NQT, CAP, R) \
    if (g.nqt == NQT && g.cap == CAP && g.ring == R) return launch_one<DT, NQT, CAP, R, MODE>(g, p, s);
    if constexpr (MODE == MODE_TOPK) {
        CASE(1, 128, 8) CASE(1, 128, 16) CASE(1, 256, 8) CASE(1, 256, 16)
        CASE(2, 128, 8) CASE(2, 128, 16) CASE(2, 256, 8) CASE(2, 256, 16)
    } else {
        CASE(1, 128, 8) CASE(1, 128, 16)
        CASE(2, 128, 8) CASE(2, 128, 16)
    }
#undef CASE
    return hipErrorInvalidValue;
}

static ScanP to_p(const CmrScanGeom& g, const CmrScanArgs& a) {
    ScanP p;
    p.corpus = reinterpret_cast<const v4u*>(a.corpus);
    p.qfrag = reinterpret_cast<const v4u*>(a.qfrag);
    p.nrows = a.nrows; p.npanels = a.npanels; p.ks = g.ks; p.k = a.k;
    p.lists = a.lists; p.cnt = a.cnt; p.mm = a.mm;
    p.scores = a.scores; p.ld = a.ld; p.nq = a.nq;
    p.sample_waves = a.sample_waves; p.sample_stride = a.sample_stride; p.tau_init = a.tau_init;
    return p;
}

hipError_t cmr_launch_scan_topk(const CmrScanGeom& g, const CmrScanArgs& a, hipStream_t s) {
    const ScanP p = to_p(g, a);
    switch (g.dtype) {
        case CMR_DT_BF16: return dispatch<CMR_DT_BF16, MODE_TOPK>(g, p, s);
        case CMR_DT_F16: return dispatch<CMR_DT_F16, MODE_TOPK>(g, p, s);
        case CMR_DT_F32: return dispatch<CMR_DT_F32, MODE_TOPK>(g, p, s);
    }
    return hipErrorInvalidValue;
}

hipError_t cmr_launch_scan_scores(const CmrScanGeom& g, const CmrScanArgs& a, hipStream_t s) {
    const ScanP p = to_p(g, a);
    switch (g.dtype) {
        case CMR_DT_BF16: return dispatch<CMR_DT_BF16, MODE_SCORES>(g, p, s);
        case CMR_DT_F16: return dispatch<CMR_DT_F16, MODE_SCORES>(g, p, s);
        case CMR_DT_F32: return dispatch<CMR_DT_F32, MODE_SCORES>(g, p, s);
    }
    return hipErrorInvalidValue;
}
