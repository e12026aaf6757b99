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
#include <cstdlib>

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

// Slow path, part 1 (inline, a handful of registers, no waits on global memory): push the keys of
// one 32x32 score tile that beat the lane's threshold into the (wave, query) lists.  Returns the
// mask of this tile's queries whose list is nearly full.
template <int CAP>
__device__ __forceinline__ u64 topk_push(const f32x16& acc, long long row0, long long nrows, u64 tau_key, int* cnt_t,
                                         u64* list_t, int lane) {
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
    // LDS ops of one wave complete in order, so the counters are current without any fence.  The
    // pushed keys themselves are only read back by a compaction, which fences (s_waitcnt vmcnt(0))
    // itself — a fence on every slow-path entry would drain the load ring / DMA ring each time.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int c = __hip_atomic_load(&cnt_t[ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __ballot(c > CAP - 32) & 0xFFFFFFFFull;
}

// Slow path, part 2 (rare): compact every list in `need` to its k best keys (rank by counting; keys
// are unique so ranks are a permutation) and raise the owning lanes' thresholds.
template <int CAP>
__device__ __forceinline__ void topk_compact(u64 need, int k, u64& tau_key, float& tau_f, int* cnt_t, u64* list_t, u64* stage,
                                             int lane) {
    constexpr int EPL = CAP / 64;
    const int ql = lane & 31;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // pushes landed (same-CU L1 is coherent)
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

template <int CAP>
__device__ __forceinline__ void topk_slow_path(const f32x16& acc, long long row0, long long nrows, int k,
                                               u64& tau_key, float& tau_f, int* cnt_t, u64* list_t,
                                               u64* stage, int lane) {
    const u64 need = topk_push<CAP>(acc, row0, nrows, tau_key, cnt_t, list_t, lane);
    if (need) topk_compact<CAP>(need, k, tau_key, tau_f, cnt_t, list_t, stage, lane);
}

// Out-of-line compaction for the wide kernel: its resident query registers must not be squeezed by
// the compaction's temporaries, and a call's callee-saved spill/reload (with its s_waitcnt vmcnt(0),
// which drains the DMA ring) must stay off the common slow path — so only the rare compaction is a
// call; the pushes are inline.
template <int CAP>
__device__ __attribute__((noinline)) void topk_compact_call(u64 need, int k, u64* tau_key, float* tau_f, int* cnt_t, u64* list_t,
                                                            u64* stage, int lane) {
    topk_compact<CAP>(need, k, *tau_key, *tau_f, cnt_t, list_t, stage, lane);
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

// ------------------------------------------------------------------------------------------
// Wide-batch scan: up to 256 queries in ONE pass over the corpus (BASELINE config 3's batch-256).
// The LDS-resident query tile of scan_kernel tops out at 64 queries (96 KiB); here the queries
// live in REGISTERS: every wave keeps the MFMA B-operands of ITS 32 queries resident (KS blocks x
// 4 VGPRs = 192 registers at 768-d), a workgroup of WAVES waves covers WAVES*32 queries (8 waves,
// two per SIMD, 256 queries at 768-d; 4 waves, 128 queries at 1024-d where a tile needs 256
// registers).  All waves consume the same corpus blocks: the stream goes HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4: one wave moves one 1-KiB block, lane-linear — exactly the block layout)
// into a ring of NST groups of 8 blocks, NST-1 groups (88 KiB) in flight per workgroup and no VGPR
// spent on it; every wave reads each block back (ds_read_b128, one block ahead) and issues one
// MFMA per block.  HBM traffic stays 1x while the MFMA work is 4x that of the 64-query kernel
// (75 % of the MFMA pipe at the HBM rate for 768-d bf16: still HBM-bound on paper).
//   per group g:  first ds_reads of group g        (validated by the previous iteration's barrier)
//                 s_waitcnt vmcnt(PPG*(NST-3))   my DMA pieces of group g+1 have landed
//                 s_barrier                        everybody's have; everybody left group g-1
//                 DMA group g+NST-1 -> stage (g-1) mod NST
//                 8 x (MFMA + ds_read_b128 three blocks ahead) from stage g mod NST
// Counted waits + raw s_barrier (__syncthreads() would drain the DMA queue, guide §5); the DMAs are
// inline asm (§5.7 recipe: M0 = LDS destination, saved/restored inside the statement) so hipcc
// neither counts them nor drains them before LDS reads.  The tail over-reads NST-1 groups past the
// range: CMR_CORPUS_SLACK covers it.  The top-k epilogue, candidate lists and thresholds are the
// per-wave ones of scan_kernel; list / counter rows are laid out [workgroup][WAVES*32 queries] so
// merge_query_kernel consumes them with W = gridDim.x.
#define WIDE_GROUP 8      // blocks per staged group
#define WIDE_STAGES 12    // LDS ring depth in groups (96 KiB)

// GRP blocks per staged group, NSTG groups in the LDS ring (GRP * NSTG = 96 KiB).  Default 8 x 12; 16 x 6 (CMR_WIDE_GROUP=16)
// halves the barriers and DMA issue events per block.
// BURST = 2 (CMR_WIDE_BURST=1, experimental): the MFMAs are issued in adjacent PAIRS with one operand wait in front
// of the pair.  Every MFMA of a panel accumulates into the same 16 registers; an instruction between two such MFMAs
// (here: the s_waitcnt + ds_read_b128 of the operand stream) costs the forwarding window, ~43 cycles per MFMA
// (MI355X_MICROARCH.md constants table) — per pair instead of per MFMA with BURST = 2.  The chain order, hence every
// score bit, is unchanged.
// STAG = 1 (CMR_WIDE_STAGGER=1, experimental): the DMA of a group is issued by ONE wave per SIMD only — waves w and
// w+4 share a SIMD, waves 0-3 load the even groups, 4-7 the odd ones, two pieces each — so that in every group each
// SIMD has a wave that goes from the barrier straight back to its MFMAs.
template <int DT, int KS, int WAVES, int CAP, int ABL = 0, int GRP = WIDE_GROUP, int NSTG = WIDE_STAGES, int STAG = 0, int BURST = 1>   // ABL: developer ablation (1 no MFMA, 2 no DMA, 3 no barrier)
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void scan_wide_kernel(ScanP P) {
    static_assert(KS % GRP == 0 && GRP % WAVES == 0 && GRP / WAVES <= 2, "group/wave geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NQB = WAVES * 32;               // queries per workgroup pass
    constexpr int GPP = KS / GRP;          // groups per panel
    constexpr int NST = NSTG;
    constexpr int PPG = GRP / WAVES;       // DMA pieces per wave per group
    // With two waves per SIMD a wave owns 256 registers: 48 resident fragments (192) + accumulator,
    // read-ahead blocks and epilogue state overflow by a few registers, and hipcc's spill reloads
    // (scratch_load + s_waitcnt vmcnt(0) at the top of every panel) drain the DMA ring — measured
    // 1.3 of 4.7 ms.  The last KLDS k-steps of the tile are therefore served from LDS.
    static_assert(BURST == 1 || (BURST == 2 && WAVES == 8 && GRP % 2 == 0 && ABL == 0), "burst geometry");
    constexpr int KLDS = WAVES == 8 ? 4 : 0;
    constexpr int KREG = KS - KLDS;

    v4u* stage_lds = reinterpret_cast<v4u*>(smem);                                    // [NST][GRP][64]
    int* cnt_all = reinterpret_cast<int*>(smem + NST * GRP * 1024);            // [WAVES][32]
    u64* cstage_all = reinterpret_cast<u64*>(cnt_all + WAVES * 32);                   // [WAVES][CAP+2]
    v4u* qlds = reinterpret_cast<v4u*>(cstage_all + WAVES * (CAP + 2)) + (size_t)wave * KLDS * 64 + lane;   // [WAVES][KLDS][64]
    int* cnt_w = cnt_all + wave * 32;
    u64* cstage = cstage_all + wave * (CAP + 2);
    for (int i = tid; i < WAVES * 32; i += WAVES * 64) cnt_all[i] = 0;

    // this wave's query fragments -> registers (static indices everywhere below)
    v4u qreg[KREG];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const v4u v = P.qfrag[((size_t)wave * KS + ks) * 64 + lane];
        if (ks < KREG) qreg[ks < KREG ? ks : 0] = v;
        else qlds[(ks - KREG) * 64] = v;                 // written and read by the same lane only
    }
    // Make hipcc retire every query-fragment load HERE: left alone it defers each wait to the
    // fragment's first use inside the panel loop, where stale low-count s_waitcnt vmcnt(N) would
    // drain the hand-counted DMA ring on every iteration.
#pragma unroll
    for (int ks = 0; ks < KREG; ++ks) asm volatile("" ::"v"(qreg[ks]));

    const int nb = gridDim.x;
    int p0, p1;
    if (P.sample_waves > 0) {   // sampling pass: workgroup b scans the single strided panel b*stride
        p0 = blockIdx.x * P.sample_stride;
        p1 = (int)blockIdx.x < P.sample_waves ? p0 + 1 : p0;
    } else {
        p0 = (int)(((long long)blockIdx.x * P.npanels) / nb);
        p1 = (int)(((long long)(blockIdx.x + 1) * P.npanels) / nb);
    }

    float rmin = __builtin_inff(), rmax = -__builtin_inff(), tau_f = -__builtin_inff();
    u64 tau_key = 0ull;
    {
        const int q = wave * 32 + (lane & 31);
        if (q >= P.nq) {
            tau_key = ~0ull;
            tau_f = __builtin_inff();
        } else if (P.tau_init) {
            tau_key = P.tau_init[q];
            if (tau_key) tau_f = cmr_key_score(tau_key);
        }
    }
    u64* list_w = P.lists + ((size_t)blockIdx.x * NQB + (size_t)wave * 32) * CAP;
    __syncthreads();

    if (p1 > p0) {
        static_assert(!STAG || (WAVES == 8 && GRP == 8 && (KS / GRP) % 2 == 0 && NSTG % 2 == 0 && ABL == 0), "stagger geometry");
        const int half = wave >> 2;                          // 0: loads even groups, 1: odd groups (STAG only)
        const int wslot = STAG ? (wave & 3) : wave;          // first block of the group this wave moves
        const char* gsrc = reinterpret_cast<const char*>(P.corpus + (size_t)p0 * KS * 64) + (size_t)wslot * 1024 + (size_t)lane * 16;
        const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) + (unsigned)wslot * 1024u;
        auto dma_group = [&](const char* g_src, int stage) {
            const unsigned dst = lds_base + (unsigned)stage * (GRP * 1024u);   // wave-uniform LDS byte address
            unsigned keep;
            if constexpr (STAG) {                            // blocks wslot and wslot + 4 of the group
                const char* g_src2 = g_src + 4096;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                             "s_add_u32 m0, %3, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g_src), "v"(g_src2), "s"(dst) : "memory", "scc");
            } else if constexpr (PPG == 1) {
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g_src), "s"(dst) : "memory");
            } else {
                const char* g_src2 = g_src + WAVES * 1024;          // piece j of wave w is block w + j*WAVES of the group
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                             "s_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g_src), "v"(g_src2), "s"(dst), "n"(WAVES * 1024) : "memory", "scc");
            }
        };
#pragma unroll
        for (int d = 0; d < NST - 1; ++d)
            if (!STAG || half == (d & 1)) dma_group(gsrc + (size_t)d * GRP * 1024, d);
        gsrc += (size_t)(NST - 1) * GRP * 1024;
        int st = 0;                                   // stage holding the current group
        // group 0 must be complete before the first reads; from then on the barrier of iteration g
        // validates group g+1, so the reads of group g are issued BEFORE that barrier and the MFMA
        // chain never drains at a barrier (measured: ~600 of 1100 cycles per group were that bubble)
        if constexpr (STAG) {
            // half 0 issued groups 0, 2, .. NST-2 (two pieces each): group 0 has landed once <= NST-2 pieces are out.
            // In iteration g the loaders of group g+1 (== those of group g+NST-1) have issued, after it, the groups
            // g+3, g+5, .. g+NST-3: NST-4 pieces may stay in flight.
            if (half == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST - 2) : "memory");
            asm volatile("s_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPG * (NST - 2)) : "memory");
        }

        for (int p = p0; p < p1; ++p) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int g = 0; g < GPP; ++g) {
                const v4u* buf = stage_lds + (size_t)st * GRP * 64 + lane;
                constexpr int ADEPTH = BURST == 2 ? 4 : (WAVES == 8 ? 3 : 4);   // blocks read ahead of the barrier
                v4u a[BURST == 2 ? 6 : ADEPTH];
#pragma unroll
                for (int u = 0; u < ADEPTH; ++u) a[u] = buf[u * 64];
                if constexpr (STAG) {
                    const bool loader = half == ((g + 1) & 1);
                    if (loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST - 4) : "memory");
                    asm volatile("s_barrier" ::: "memory");
                    if (loader) dma_group(gsrc + (size_t)g * GRP * 1024, st == 0 ? NST - 1 : st - 1);
                } else
                if constexpr (ABL == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPG * (NST - 3)) : "memory");
                else if constexpr (ABL == 2 || ABL == 5) asm volatile("s_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPG * (NST - 3)) : "memory");
                if constexpr (!STAG && ABL != 2 && ABL != 5) dma_group(gsrc + (size_t)g * GRP * 1024, st == 0 ? NST - 1 : st - 1);
                if constexpr (BURST == 2) {
#pragma unroll
                    for (int pr = 0; pr < GRP / 2; ++pr) {
                        const int ks0 = g * GRP + 2 * pr, ks1 = ks0 + 1;
                        v4u b0 = ks0 < KREG ? qreg[ks0 < KREG ? ks0 : 0] : qlds[(ks0 < KREG ? 0 : ks0 - KREG) * 64];
                        v4u b1 = ks1 < KREG ? qreg[ks1 < KREG ? ks1 : 0] : qlds[(ks1 < KREG ? 0 : ks1 - KREG) * 64];
                        v4u a0 = a[(2 * pr) % 6], a1 = a[(2 * pr + 1) % 6];
                        __builtin_amdgcn_sched_barrier(0);
                        // ONE lgkmcnt wait, in front of the pair: both MFMAs consume the statement's outputs, so it
                        // cannot sink between them (an input-only statement did, and split the wait again)
                        if (ks1 < KREG) asm volatile("" : "+v"(a0), "+v"(a1));
                        else if (ks0 < KREG) asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b1));
                        else asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                        acc = CmrBlk<DT>::mma(a0, b0, acc);
                        acc = CmrBlk<DT>::mma(a1, b1, acc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (2 * pr + 4 < GRP) {                                  // operands of the pair after next
                            a[(2 * pr + 4) % 6] = buf[(2 * pr + 4) * 64];
                            a[(2 * pr + 5) % 6] = buf[(2 * pr + 5) * 64];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    const v4u a_use = a[u % ADEPTH];
                    __builtin_amdgcn_sched_barrier(0);
                    const int ks = g * GRP + u;
                    const v4u b = ks < KREG ? qreg[ks < KREG ? ks : 0] : qlds[(ks < KREG ? 0 : ks - KREG) * 64];
                    if constexpr (ABL == 1 || ABL == 4) asm volatile("" ::"v"(a_use), "v"(b));
                    else acc = CmrBlk<DT>::mma(a_use, b, acc);
                    if (u + ADEPTH < GRP) a[u % ADEPTH] = buf[(u + ADEPTH) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                }
                st = st + 1 == NST ? 0 : st + 1;
            }
            gsrc += (size_t)KS * 1024;

            if constexpr (ABL >= 4) {   // ablation: keep acc alive, skip the epilogue
                asm volatile("" ::"v"(acc));
                continue;
            }
            const long long row0 = (long long)p * CMR_PANEL_ROWS;
            const bool partial = row0 + CMR_PANEL_ROWS > P.nrows;
            float mx, mn;
            if (!partial) {
                mx = acc[0]; mn = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) { mx = fmaxf(mx, acc[r]); mn = fminf(mn, acc[r]); }
            } else {
                mx = -__builtin_inff(); mn = __builtin_inff();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = row0 + cmr_acc_row(r, lane) < P.nrows;
                    mx = fmaxf(mx, ok ? acc[r] : -__builtin_inff());
                    mn = fminf(mn, ok ? acc[r] : __builtin_inff());
                }
            }
            rmax = fmaxf(rmax, mx);
            rmin = fminf(rmin, mn);
            if (__any(mx >= tau_f)) {
                const u64 need = topk_push<CAP>(acc, row0, P.nrows, tau_key, cnt_w, list_w, lane);
                if (need) {
                    // address-taken copies live only inside this branch (passing &tau_f itself would pin
                    // it to scratch and its reload's s_waitcnt vmcnt(0) would drain the DMA ring on
                    // every panel)
                    u64 tk = tau_key;
                    float tf = tau_f;
                    topk_compact_call<CAP>(need, P.k, &tk, &tf, cnt_w, list_w, cstage, lane);
                    tau_key = tk;
                    tau_f = tf;
                    // retire the scratch reloads of tk / tf HERE: otherwise hipcc waits for them at their
                    // next use — an unconditional s_waitcnt vmcnt(0) in front of every panel's threshold
                    // compare, which drains the DMA ring each time
                    asm volatile("" : "+v"(tau_key), "+v"(tau_f));
                }
            }
        }
    }

    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    {
        const float mn = fminf(rmin, __shfl_xor(rmin, 32));
        const float mx = fmaxf(rmax, __shfl_xor(rmax, 32));
        if (lane < 32) {
            const int q = wave * 32 + lane;
            P.mm[(size_t)blockIdx.x * NQB + q] = make_float2(mn, mx);
            P.cnt[(size_t)blockIdx.x * NQB + q] = __hip_atomic_load(&cnt_w[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

static int wide_waves(int ks) { return ks == 48 ? 8 : 4; }

size_t cmr_wide_lds_bytes(int ks, int cap, int variant) {
    const int waves = wide_waves(ks);
    const int klds = waves == 8 ? 4 : 0;
    (void)variant;                        // every variant built so far has the same footprint
    // the corpus ring is 96 KiB for every (group, stages) variant
    return (size_t)WIDE_STAGES * WIDE_GROUP * 1024 + (size_t)waves * 32 * 4 + (size_t)waves * (cap + 2) * 8 +
           (size_t)waves * klds * 1024;
}

// wide kernel availability: 16-bit dtypes at ks = 48 (768-d: 8 waves x 32 = 256 queries per pass),
// ks = 64 (1024-d: a tile needs 256 registers -> 4 waves x 32 = 128 queries per pass)
int cmr_wide_queries(int dtype, int dpad) {
    if (dtype == CMR_DT_F32) return 0;
    if (dpad == 768) return 256;
    if (dpad == 1024) return 128;
    return 0;
}

hipError_t cmr_launch_scan_wide(const CmrScanGeom& gin, const CmrScanArgs& a, hipStream_t s) {
    CmrScanGeom g = gin;
    if (g.wide_group < 0 && g.ks != 48) g.wide_group = 0;      // experimental variants exist for the 8-wave (768-d) shape only
    const ScanP p = to_p(g, a);
    const size_t lds = cmr_wide_lds_bytes(g.ks, g.cap, g.wide_group);
    auto launch = [&](auto kern, int threads) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(g.grid), dim3(threads), lds, s, p);
        return hipGetLastError();
    };
    const int abl = getenv("CMR_WIDE_ABL") ? atoi(getenv("CMR_WIDE_ABL")) : 0;
    if (abl && g.dtype == CMR_DT_BF16 && g.ks == 48 && g.cap == 128) {
        if (abl == 1) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 8, 128, 1>, 512);
        if (abl == 2) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 8, 128, 2>, 512);
        if (abl == 3) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 8, 128, 3>, 512);
        if (abl == 4) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 8, 128, 4>, 512);
        if (abl == 5) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 8, 128, 5>, 512);
        if (abl == 6) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 8, 128, 6>, 512);
    }
    static_assert(WIDE_STAGES * WIDE_GROUP == 6 * 16, "ring variants share one LDS size");
    if (g.wide_group == -2 && g.ks == 48) {     // experimental: paired MFMA issue (CMR_WIDE_BURST=1), 16-block groups
#define WCASEB(DT, CAPV) if (g.dtype == DT && g.cap == CAPV) return launch(scan_wide_kernel<DT, 48, 8, CAPV, 0, 16, 6, 0, 2>, 512);
        WCASEB(CMR_DT_BF16, 128) WCASEB(CMR_DT_BF16, 256) WCASEB(CMR_DT_F16, 128) WCASEB(CMR_DT_F16, 256)
#undef WCASEB
    }
    if (g.wide_group == -1 && g.ks == 48) {     // experimental: staggered DMA issue (CMR_WIDE_STAGGER=1)
#define WCASES(DT, CAPV) if (g.dtype == DT && g.cap == CAPV) return launch(scan_wide_kernel<DT, 48, 8, CAPV, 0, 8, 12, 1>, 512);
        WCASES(CMR_DT_BF16, 128) WCASES(CMR_DT_BF16, 256) WCASES(CMR_DT_F16, 128) WCASES(CMR_DT_F16, 256)
#undef WCASES
    }
    if (g.wide_group == 16 && g.ks == 48) {     // experimental: 16-block groups, 6 stages (8-wave variants only)
#define WCASE16(DT, CAPV) if (g.dtype == DT && g.cap == CAPV) return launch(scan_wide_kernel<DT, 48, 8, CAPV, 0, 16, 6>, 512);
        WCASE16(CMR_DT_BF16, 128) WCASE16(CMR_DT_BF16, 256) WCASE16(CMR_DT_F16, 128) WCASE16(CMR_DT_F16, 256)
#undef WCASE16
    }
#define WCASE(DT, KSV, WV, CAPV) if (g.dtype == DT && g.ks == KSV && g.cap == CAPV) return launch(scan_wide_kernel<DT, KSV, WV, CAPV>, WV * 64);
    WCASE(CMR_DT_BF16, 48, 8, 128) WCASE(CMR_DT_BF16, 48, 8, 256) WCASE(CMR_DT_F16, 48, 8, 128) WCASE(CMR_DT_F16, 48, 8, 256)
    WCASE(CMR_DT_BF16, 64, 4, 128) WCASE(CMR_DT_BF16, 64, 4, 256) WCASE(CMR_DT_F16, 64, 4, 128) WCASE(CMR_DT_F16, 64, 4, 256)
#undef WCASE
    return hipErrorInvalidValue;
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
