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
#include <type_traits>

#include "cmr_device.h"
#include "cmr_kernels.h"
#include "cmr_select.h"

// this translation unit only: the wide kernel's LDS-DMA statements list m0 as a clobber on purpose (cdna_hip_programming.md 5.7) and hipcc
// warns once per inlined copy; asm diagnostics stay on in every other file
#pragma clang diagnostic ignored "-Winline-asm"

#define MODE_TOPK 0
#define MODE_SCORES 1
// MODE_FIN: top-k with the thresholds AND the final selection inside the launch (a synchronous caller's handful of queries on a
// mid-size or large corpus: the sampling scan, its merge and the candidate merge are a chain of dependent launches around a
// short scan there) — see "finishing stage" in scan_kernel
#define MODE_FIN 2
#ifndef CMR_FIN_HINT
#define CMR_FIN_HINT 1     // finishing stage, hand-counted ring: the READY word is asked for inside the ring's own load sequence and only hints at the proper look (below)
#endif

// Cache policy of the corpus stream's loads: non-temporal.  Every byte of the corpus is read once per launch by one CU,
// so keeping the lines in L2 / MALL only evicts what the other kernels of the pipeline use; measured with
// -DCMR_STREAM_NT=0 / 1 builds: 10 M x 768 bf16, B = 64 step 2.567 -> 2.401 ms, 1 M rows 0.294 -> 0.282 ms (the wide
// kernel, which is not HBM-bound, does not move).
#ifndef CMR_STREAM_NT
#define CMR_STREAM_NT 1
#endif
#if CMR_STREAM_NT
#define CMR_STREAM_POLICY " nt"
#define CMR_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define CMR_STREAM_POLICY ""
#define CMR_STREAM_LOAD(p) (*(p))
#endif
// scan_kernel carries the policy as a template parameter (POL = 1: the policy above; 0: default policy — the query-split grid WANTS
// the corpus lines to stay in L2 until the twins of the other query groups have read them)
// wide kernel: 1 = all DMA pieces of a group right after its barrier, 0 = one piece per quad of blocks
#ifndef CMR_WIDE_DMA_BURST
#define CMR_WIDE_DMA_BURST 0
#endif
// Placement of a quad's LDS-DMA piece in the two-tile wide kernel.  0: in front of the quad's eight MFMAs, i.e. directly behind
// the four ds_read_b128 that refill the read-ahead ring (one boundary of the MFMA stream carries 4 LDS reads + the piece + its
// SALU: a dozen issue slots in ONE gap).  1: between tile 0's four MFMAs and tile 1's four — the other boundary between chains
// on DIFFERENT accumulators, 128 cycles behind the ring's reads (MI355X_MICROARCH.md: a gap between MFMAs hides <= 5 single-issue
// instructions, fillers between MFMAs on the SAME accumulator are never free, and a piece issued beside pending LDS reads
// costs its wave 100-185 cycles against 25-60 in a quiet gap).
#ifndef CMR_WIDE_DMA_MID
#define CMR_WIDE_DMA_MID 1
#endif
// M0 around a piece: 0 = saved and restored inside the statement (5 instructions), 1 = written and left (3; hipcc sets M0 itself
// in front of each of its own uses — v_writelane lane selects in the compaction — and expects nothing of it across an asm)
#ifndef CMR_WIDE_M0_CLOBBER
#define CMR_WIDE_M0_CLOBBER 1
#endif
// 1: a piece's displacement inside its group comes from the instruction's immediate + a loop-invariant lane offset, M0 is set inside
// the statement: no compiler-generated SALU per piece (see dma_piece_j)
#ifndef CMR_WIDE_DMA_IMM
#define CMR_WIDE_DMA_IMM 1
#endif
// 1 (two-tile kernel): both tiles' accumulators live in the VGPR half — their readers (the min / max fold, the slow path) take them as
// they are, no v_accvgpr_read: 32 fewer instructions per panel — and WIDE_AMOVE_V k-steps of tile 0's B-operands move to the AGPR half in
// exchange (which then holds 192 + 64 registers of fragments and nothing else)
#ifndef CMR_WIDE_ACC_VGPR
#define CMR_WIDE_ACC_VGPR 1
#endif
// 1: the read-ahead ring's slot j is refilled directly behind the LAST tile's MFMA j of a quad (the slot's final reader), where the
// wait state between two MFMAs of one chain is due anyway, instead of four ds_read_b128 in a row behind the quad
#ifndef CMR_WIDE_READ_INTERLEAVE
#define CMR_WIDE_READ_INTERLEAVE 1
#endif

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
    int sample_waves;      // > 0: sampling pass over sample_waves panels; sampled panel s = (s >> c) * sample_stride + (s & (2^c - 1))
    int sample_stride;     //      (chunks of 2^c consecutive panels: one TLB reach / DRAM page run per chunk instead of per panel)
    int sample_chunk_log2;
    const u64* tau_init;   // per-query initial threshold keys (from the sampling pass) or nullptr
    const u64* slists;     // lists / counters of a preceding sampling pass ([sW][32][CAP]): thresholds are derived in-kernel (nq <= 8, k <= 64)
    const int* scnt;
    int sW;
    int qgroups;           // > 1: query-split grid (see scan_kernel): gridDim.x = qgroups x virtual grid, every group has its own query tile
    // MODE_FIN
    int* fin;              // control words (cmr_kernels.h: CMR_FIN_*), every counter on a 128-byte line of its own
    u64* fin_pmax;         // [32][8 * fin_wgs] per-query maxima of the first panels of the first fin_wgs workgroups' waves
    u64* fin_tau;          // [32] published thresholds
    u64* fin_dense;        // [32][fin_dcap] the candidates that beat their wave's final threshold
    u64* fin_mm;           // [32][grid] (min, max) per workgroup
    int fin_wgs, fin_mul, fin_dcap, fin_spin, fin_first;
    int64_t* out_ids;
    float* out_scores;
    float* out_min;
    float* out_max;
    long long id_base;
    int* fin_done;         // mapped host word (or nullptr): the state once more, stored LAST — what a synchronous caller polls
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
            // (never downwards: with the finishing stage a list still holds what the wave pushed before it adopted the published
            // threshold — the k-th best of THAT is no bound worth having)
            if (ql == j && nt > tau_key) { tau_key = nt; tau_f = cmr_key_score(nt); }
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

// MODE_FIN: the threshold of query q from the published first-panel maxima of the ns supplying waves (one selection chunk:
// <= 1024 keys, 16 per lane) — the k-th largest of them, written through before the query's bit in `ready` says so
__device__ __forceinline__ void fin_threshold(const u64* pmax, int ns, int k, u64* tau, int* ready, u64* stage, int q, int lane) {
    u64 key[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = lane + 64 * j;
        key[j] = i < ns ? __hip_atomic_load(&pmax[(size_t)q * ns + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
    u64* gt = tau + q;
    tiny_select(key, k, stage, lane, [&](int r, u64 kv) {
        if (r == k - 1) __hip_atomic_store(gt, kv ? kv - 1 : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) atomicOr(ready, 1 << q);
#ifdef CMR_FIN_DEBUG
    if (lane == 0) { atomicMin((unsigned*)(ready - CMR_FIN_READY + CMR_FIN_DBG + 2 + 3), (unsigned)wall_clock64()); atomicMax((unsigned*)(ready - CMR_FIN_READY + CMR_FIN_DBG + 2 + 4), (unsigned)wall_clock64()); }      // first / last threshold published
#endif
}

// development builds (-DCMR_FIN_DEBUG): a timeline of the finishing stage in 10 ns ticks of the constant clock, slots behind CMR_FIN_DBG + 2
#ifdef CMR_FIN_DEBUG
#define CMR_FIN_STAMP_MIN(P, slot) atomicMin((unsigned*)&(P).fin[CMR_FIN_DBG + 2 + (slot)], (unsigned)wall_clock64())
#define CMR_FIN_STAMP_MAX(P, slot) atomicMax((unsigned*)&(P).fin[CMR_FIN_DBG + 2 + (slot)], (unsigned)wall_clock64())
#else
#define CMR_FIN_STAMP_MIN(P, slot) ((void)0)
#define CMR_FIN_STAMP_MAX(P, slot) ((void)0)
#endif

template <int DT, int NQT, int CAP, int R, int MODE, int ASMRING, int POL = 1>
__global__ __launch_bounds__(CMR_SCAN_THREADS, 2) void scan_kernel(ScanP P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (SGPR)
    const int KS = P.ks;
    constexpr int NQ = NQT * 32;
    constexpr bool TOPK = MODE != MODE_SCORES;
    constexpr bool FIN = MODE == MODE_FIN;
    static_assert(!FIN || NQT == 1, "the finishing stage handles one query tile");

    v4u* qf = reinterpret_cast<v4u*>(smem);
    int* cnt_all = reinterpret_cast<int*>(smem + (size_t)NQT * KS * 1024);
    u64* stage_all = reinterpret_cast<u64*>(cnt_all + CMR_SCAN_WAVES * NQ);
    int* cnt_w = cnt_all + wave * NQ;
    u64* stage = stage_all + wave * (CAP + 2);
    // FIN only (the launcher adds CMR_FIN_LDS bytes): the waves' min / max per query, reduced per workgroup at the end
    float2* fin_mmw = reinterpret_cast<float2*>(stage_all + CMR_SCAN_WAVES * (CAP + 2) + 2);      // [waves][32]
    int* fin_sh = reinterpret_cast<int*>(stage_all + CMR_SCAN_WAVES * (CAP + 2));     // FIN: [0] the workgroup's ticket, [1] a staging area overflowed, [2] waves past their first panel
    u64* fin_pm = reinterpret_cast<u64*>(fin_mmw + CMR_SCAN_WAVES * 32);             // FIN: [waves][32] per-query maxima of the waves' first panels
    // (dynamic LDS on purpose: the function's dynamic limit is the whole 160 KiB, a static variable on top of it fails the launch)

    // Query-split grid (batches of more than NQ queries in ONE corpus pass): the grid is qgroups x a virtual grid; the
    // workgroups of group g hold query tile g (queries g*NQ ..) in LDS and walk the SAME per-wave panel ranges as their
    // twins of the other groups.  Workgroup ids go round robin over the 8 XCDs, so with a virtual grid that is a multiple of
    // 8 the twins (ids c*8*G + g*8 + x, g = 0..G-1) sit on CUs of ONE XCD and start together: the first of them to ask for a
    // corpus block brings it from HBM into that XCD's L2, the others hit there (a twin that is ahead waits for HBM, one that is
    // behind runs at L2 speed and catches up) — HBM is read once per pass, the matrix work per HBM byte is G-fold.
    int bid = blockIdx.x, grp = 0, vgrid = gridDim.x;
    if (P.qgroups > 1) {
        vgrid = gridDim.x / P.qgroups;
        if ((vgrid & 7) == 0) {
            const int span = 8 * P.qgroups, chunk = bid / span, r = bid - chunk * span;
            grp = r >> 3;
            bid = chunk * 8 + (r & 7);
        } else {
            grp = bid / vgrid;
            bid -= grp * vgrid;
        }
    }
    // FIN: the workgroups that are dispatched first supply the thresholds (their first panels); a multiplicative permutation of the
    // workgroup ids spreads their panel ranges over the whole corpus (rows arrive document by document: a sample of one region
    // would be valid but loose)
    if constexpr (FIN) bid = __builtin_amdgcn_readfirstlane((int)(((unsigned)bid * (unsigned)P.fin_mul) % gridDim.x));     // (the division runs on the VALU: say that it is uniform)
    const int nq_g = P.nq - grp * NQ;          // queries of this group (the last group may be ragged)
    {
        const v4u* qsrc = P.qfrag + (size_t)grp * NQT * KS * 64;
        for (int i = tid; i < NQT * KS * 64; i += CMR_SCAN_THREADS) qf[i] = qsrc[i];
    }
    if (TOPK)
        for (int i = tid; i < CMR_SCAN_WAVES * NQ; i += CMR_SCAN_THREADS) cnt_all[i] = 0;
    if (FIN && tid == 0) { fin_sh[1] = 0; fin_sh[2] = 0; }
    if constexpr (FIN) { if (tid == 0) { CMR_FIN_STAMP_MIN(P, 0); CMR_FIN_STAMP_MAX(P, 1); } }      // first / last workgroup start
    __syncthreads();

    const int W = vgrid * CMR_SCAN_WAVES;
    const int gw = bid * CMR_SCAN_WAVES + wave;
    const size_t gwl = (size_t)grp * W + gw;   // this wave's row of the list / counter / min-max arrays ([group][wave])
    int p0, p1;
    if (P.sample_waves > 0) {   // sampling pass: one strided panel per wave
        p0 = (gw >> P.sample_chunk_log2) * P.sample_stride + (gw & ((1 << P.sample_chunk_log2) - 1));
        p1 = gw < P.sample_waves ? p0 + 1 : p0;
    } else {
        p0 = (int)(((long long)gw * P.npanels) / W);
        p1 = (int)(((long long)(gw + 1) * P.npanels) / W);
    }

    // Thresholds straight from a sampling pass's lists (a handful of queries on a mid-size corpus: a synchronous caller's chain
    // of dependent launches).  Wave w takes query w: every lane keeps the best key of "its" sample lists (lists lane, lane + 64,
    // ...); the k-th largest of those 64 lane maxima is a lower bound of the k-th best key of the whole corpus — k lanes hold
    // k distinct rows at or above it — and nearly as tight as the exact k-th best of the sample (the 20 best of 4096 sample keys
    // sit in ~17 different lanes).  Every workgroup does this for itself (~2 us beside the query-tile fill): the merge launch
    // between the two scans (21 us + its launch gap at 1 M rows) is gone.  Any valid lower bound leaves the results unchanged.
    if constexpr (MODE == MODE_TOPK && NQT == 1) {
        if (P.slists) {                          // wave-uniform
            if (wave < nq_g) {
                u64 best = 0ull;
                for (int w = lane; w < P.sW; w += 64) {
                    // a sampling wave scans ONE panel: <= 32 keys per list, read as 16 independent 16-byte loads (a list has room
                    // for CAP >= 128 keys: in bounds whatever the count; keys beyond 32 would only be left out of the bound)
                    const int c = P.scnt[(size_t)w * NQ + wave];
                    const uint4* L4 = reinterpret_cast<const uint4*>(P.slists + ((size_t)w * NQ + wave) * CAP);
                    uint4 v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = L4[i];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const u64 k0 = ((u64)v[i].y << 32) | v[i].x, k1 = ((u64)v[i].w << 32) | v[i].z;
                        if (2 * i < c) best = k0 > best ? k0 : best;
                        if (2 * i + 1 < c) best = k1 > best ? k1 : best;
                    }
                }
                int rk = 0;
                for (int l = 0; l < 64; ++l) {
                    const u64 o = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(best >> 32), l) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane((int)best, l);
                    rk += o > best ? 1 : 0;
                }
                const u64 has = __ballot(best != 0ull && rk == P.k - 1);      // fewer than k lanes hold a key: no threshold
                if (has && lane == __ffsll((long long)has) - 1) stage_all[(size_t)wave * (CAP + 2) + CAP + 1] = best - 1;
                if (!has && lane == 0) stage_all[(size_t)wave * (CAP + 2) + CAP + 1] = 0ull;
            }
            __syncthreads();
        }
    }

    float rmin[NQT], rmax[NQT], tau_f[NQT];
    u64 tau_key[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        rmin[t] = __builtin_inff();
        rmax[t] = -__builtin_inff();
        tau_f[t] = -__builtin_inff();
        tau_key[t] = 0ull;
        if (TOPK) {
            const int q = t * 32 + (lane & 31);
            if (q >= nq_g) {                     // padding query (all-zero operand): nothing may pass
                tau_key[t] = ~0ull;
                tau_f[t] = __builtin_inff();
            } else if (!FIN && NQT == 1 && P.slists) {   // derived above from the sampling pass's lists (q < nq <= 8 waves)
                tau_key[t] = stage_all[(size_t)q * (CAP + 2) + CAP + 1];
                if (tau_key[t]) tau_f[t] = cmr_key_score(tau_key[t]);
            } else if (P.tau_init) {             // a valid lower bound on the global k-th best key
                tau_key[t] = P.tau_init[grp * NQ + q];
                if (tau_key[t]) tau_f[t] = cmr_key_score(tau_key[t]);
            }
        }
    }
    u64* list_w = TOPK ? P.lists + gwl * NQ * CAP : nullptr;
    int fin_phase = 0;      // FIN: 0 = first panel pending, 1 = waiting for the published thresholds, 2 = adopted
#ifdef CMR_FIN_DEBUG
    unsigned dbg_t_slow = 0, dbg_n_slow = 0, dbg_n_whole = 0, dbg_t_thr = 0, dbg_t_epi = 0;      // ticks (10 ns) / counts of this wave
#define CMR_DBG_T0 const unsigned dbg_t0_ = (unsigned)wall_clock64();
#define CMR_DBG_ADD(V) V += (unsigned)wall_clock64() - dbg_t0_;
#else
#define CMR_DBG_T0
#define CMR_DBG_ADD(V)
#endif
    if constexpr (FIN) {
        // The workgroups that do not supply the thresholds look for them before they start: the second round of workgroups (a CU
        // holds one at a time) finds them published and never scans without.  fin_spin > 0 makes wave 0 look again every ~1.5 us
        // for that many rounds — a BOUNDED wait on workgroups that were dispatched earlier and wait for nobody, so it cannot
        // deadlock — but the head start does not pay: half the CUs idle while the suppliers' first panels stream at half the
        // rate (2 M rows, one query: scan 546 us with 40 rounds, 518 with none).
        if ((int)blockIdx.x >= P.fin_first) {
            if (wave == 0) {
                int rdy = 0;
                for (int spin = 0;; ++spin) {
                    const int full = (int)((1u << nq_g) - 1u);
                    rdy = (__hip_atomic_load(&P.fin[CMR_FIN_READY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & full) == full;
                    if (rdy || spin >= P.fin_spin) break;
                    __builtin_amdgcn_s_sleep(48);
                }
                if (lane == 0) fin_sh[3] = rdy;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            fin_phase = 1;
            if (fin_sh[3]) {
                const int q = lane & 31;
                const u64 gt = q < nq_g ? __hip_atomic_load(&P.fin_tau[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                if (q < nq_g && gt > tau_key[0]) { tau_key[0] = gt; tau_f[0] = cmr_key_score(gt); }
                fin_phase = 2;
            }
        }
    }

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
            /* NQT == 1: the accumulator rides along as a second tied operand — with ONE MFMA per slot LLVM otherwise sinks the   */ \
            /* MFMAs below the reloads and keeps each slot alive in a COPY made before its wait (stale data; build.py's audit)    */ \
            if constexpr (NQT == 1 && POL)                                                                 \
                asm volatile("global_load_dwordx4 %0, %2, %3 offset:%4" CMR_STREAM_POLICY                  \
                             : "+v"(slot_), "+v"(acc[0]) : "v"(voff[(u) >> 2]), "s"(sbase), "n"(((u) & 3) * 1024) : "memory"); \
            else if constexpr (NQT == 1)                                                                   \
                asm volatile("global_load_dwordx4 %0, %2, %3 offset:%4"                                    \
                             : "+v"(slot_), "+v"(acc[0]) : "v"(voff[(u) >> 2]), "s"(sbase), "n"(((u) & 3) * 1024) : "memory"); \
            else if constexpr (POL)                                                                        \
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" CMR_STREAM_POLICY                  \
                             : "+v"(slot_) : "v"(voff[(u) >> 2]), "s"(sbase), "n"(((u) & 3) * 1024) : "memory"); \
            else                                                                                           \
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"                                    \
                             : "+v"(slot_) : "v"(voff[(u) >> 2]), "s"(sbase), "n"(((u) & 3) * 1024) : "memory"); \
            buf[(u)] = slot_;                                                                              \
        } else {                                                                                           \
            if constexpr (POL) buf[(u)] = CMR_STREAM_LOAD(&src[(size_t)(u) * 64]);                         \
            else               buf[(u)] = src[(size_t)(u) * 64];                                           \
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
            if (TOPK) acc[t] = CmrBlk<DT>::mma(buf[(u)], b, acc[t]); /* D[row][query] */                \
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
        [[maybe_unused]] unsigned fin_hint = 0u, fin_zoff = 0u;
        [[maybe_unused]] bool fin_hint_valid = false;
        asm volatile("" : "+v"(fin_zoff));
        for (long long it = -1; it < ngroups; ++it) {
            if constexpr (FIN && ASMRING && CMR_FIN_HINT) {
                // A wave that has not adopted the thresholds yet asks for the READY word at the START of a panel's last group — one more load in
                // the ring's own sequence, nothing waits for it: loads return in order, so the counted waits of the sixteen ring steps that follow
                // retire it on the way.  At the panel end the value is only a HINT: the proper device-scope load (whose s_waitcnt vmcnt(0) drains
                // the whole ring: 3-4 us without a request from this wave, at every panel end until the adoption) is made only when the hint says
                // there is something to fetch.  Whatever the register holds if the compiler ever moved it — an old value, anything — costs one
                // look too many or one panel's delay, never a result.
                // (one or two queries only: with more, the thresholds are taken query by query by the waves that pass a panel end after the
                // publication, and a wave that looks at a third-of-a-panel-old hint joins that later — 8 queries: +10 us per call, measured)
                if (fin_phase == 1 && it >= 0 && gi == GPP - 1 && nq_g <= 2) {
                    // (s_nop 4: with 100 SGPRs spilled the base may have come back through v_readlane in the instruction before — a VALU write of an
                    // SGPR that a VMEM instruction reads as its address needs five wait states, and hipcc pads none for inline asm: the first
                    // version faulted on exactly that in the 8-block-ring variant)
                    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 offset:%3 sc1" : "=v"(fin_hint) : "v"(fin_zoff), "s"(P.fin), "n"(CMR_FIN_READY * 4) : "memory");
                    fin_hint_valid = true;
                }
            }
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
            if (TOPK) {
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
                    if constexpr (FIN) {
                        // no threshold yet (tau_key 0; padding queries ~0): the whole panel goes to the lists — 16 plain stores per
                        // lane at slots that follow from the list length, instead of 16 dependent LDS-atomic + store pairs
                        int c0 = 0;
                        const int ql = lane & 31;
                        const bool valid_q = ql < nq_g;
                        if (valid_q) c0 = __hip_atomic_load(&cnt_w[ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        // (the lists' invariant — at most CAP - 32 keys in front of a push — holds afterwards too; a panel that
                        // ends the corpus or holds a NaN score takes the general path: no empty keys inside a list)
                        bool nan = false;
#pragma unroll
                        for (int r = 0; r < 16; ++r) nan |= acc[0][r] != acc[0][r];
                        if (fin_phase < 2 && !partial && !__any(valid_q && (tau_key[0] != 0ull || c0 > CAP - 64 || nan))) {
                            if (valid_q) {
                                u64* dst = list_w + (size_t)ql * CAP + c0 + 16 * (lane >> 5);
#pragma unroll
                                for (int r = 0; r < 16; ++r) dst[r] = cmr_make_key(acc[0][r], (unsigned)(row0 + cmr_acc_row(r, lane)));
                                if (lane < 32) __hip_atomic_store(&cnt_w[ql], c0 + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
#ifdef CMR_FIN_DEBUG
                            ++dbg_n_whole;
#endif
                            continue;
                        }
                    }
                    if (__any(mx >= tau_f[t])) {
#ifdef CMR_FIN_DEBUG
                        ++dbg_n_slow;
#endif
                        CMR_DBG_T0
                        topk_slow_path<CAP>(acc[t], row0, P.nrows, P.k, tau_key[t], tau_f[t], cnt_w + t * 32,
                                            list_w + (size_t)t * 32 * CAP, stage, lane);
                        CMR_DBG_ADD(dbg_t_slow)
                    }
                }
                if constexpr (FIN) {
                    // Finishing stage, part 1 — thresholds without a sampling launch.  Every wave starts with no threshold (its
                    // first panel goes to the lists whole).  The waves of the fin_wgs workgroups that were dispatched first publish
                    // the per-query maximum of their first panel (slot = workgroup x 8 + wave); the wave that completes the last
                    // of those workgroups takes the k-th largest of the 8 x fin_wgs maxima per query — k distinct rows at or above
                    // it: a valid lower bound of the global k-th best, as tight as a sample of that many panels — and publishes
                    // it.  Nobody waits: a wave looks for the published thresholds at the end of each later panel (normally its
                    // second) and scans on without them until then; what it pushed too generously is dropped when its list is
                    // handed over (part 2).  All of it through device-scope loads / stores of the words concerned (write-through,
                    // cache-bypassing) and counters on lines of their own: a release / acquire FENCE writes back / invalidates the
                    // XCD's whole L2, and 4000 waves doing that — or 9000 atomics on one line — cost more than the launches saved
                    // (measured: 2 M rows, one query, 741 us against 471).
                    // (lane ids opaque in here: the per-lane addresses below would otherwise be hoisted out of the panel loop and
                    // held — or spilled — across it)
                    int lane_o = lane;
                    asm volatile("" : "+v"(lane_o));
                    CMR_DBG_T0
                    const int fin_phase_was_ = fin_phase;
                    (void)fin_phase_was_;
                    if (fin_phase == 0) {
                        // the per-query maxima of this wave's first panel go to LDS; the workgroup's last wave takes a ticket, and the first
                        // fin_wgs workgroups to get one — the FASTEST, whichever they are: the slowest of a fixed set of 128 was through its
                        // first panels at 58-95 us of a 270 us scan, the 64th of all at 31-36 (development build's stamps) — copy their eight
                        // maxima per query to the ticket's slots
                        u64 best = 0ull;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const long long row = row0 + cmr_acc_row(r, lane_o);
                            const float v = acc[0][r];
                            const u64 key = cmr_make_key(v, (unsigned)row);
                            if (row < P.nrows && v == v && key > best) best = key;
                        }
                        const u64 other = __shfl_xor(best, 32);
                        best = other > best ? other : best;
                        if (lane_o < 32) fin_pm[wave * 32 + lane_o] = lane_o < nq_g ? best : 0ull;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane_o == 0) { CMR_FIN_STAMP_MIN(P, 11); }      // the first first panel of a wave is through
                        int o = 0;
                        if (lane_o == 0) o = atomicAdd(&fin_sh[2], 1);
                        o = __builtin_amdgcn_readfirstlane(o);
                        if (o == CMR_SCAN_WAVES - 1) {                            // the workgroup's last wave: one device atomic per workgroup
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            int dn = 0;
                            if (lane_o == 0) dn = atomicAdd(&P.fin[CMR_FIN_DONE], 1);
                            dn = __builtin_amdgcn_readfirstlane(dn);
#ifdef CMR_FIN_DEBUG
                            if (lane_o == 0) { const int sl = dn == 0 ? 12 : dn == 15 ? 13 : dn == 31 ? 14 : dn == 63 ? 15 : dn == 95 ? 16 : -1; if (sl > 0) P.fin[CMR_FIN_DBG + 2 + sl] = (int)(unsigned)wall_clock64(); }
#endif
                            if (dn < P.fin_wgs) {
                                const int ns = P.fin_wgs * CMR_SCAN_WAVES;
                                if (lane_o < nq_g) {
#pragma unroll
                                    for (int w2 = 0; w2 < CMR_SCAN_WAVES; ++w2)
                                        __hip_atomic_store(&P.fin_pmax[(size_t)lane_o * ns + dn * CMR_SCAN_WAVES + w2], fin_pm[w2 * 32 + lane_o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // written through before this workgroup is counted
                                int pb = 0;
                                if (lane_o == 0) pb = atomicAdd(&P.fin[CMR_FIN_PUB], 1);
                                pb = __builtin_amdgcn_readfirstlane(pb);
                                if (pb == P.fin_wgs - 1) {
                                    // every slot is filled: bit 31 says so, and the thresholds are taken query by query by
                                    // whoever claims one — this wave, and every wave that comes past a panel end meanwhile
                                    if (lane_o == 0) { atomicOr(&P.fin[CMR_FIN_READY], (int)0x80000000u); CMR_FIN_STAMP_MAX(P, 2); }      // every slot of first-panel maxima filled
                                    // (a counted loop, every lane in the atomic: whatever the compiler makes of it, it ends)
                                    for (int it = 0; it < nq_g; ++it) {
                                        int cq = __hip_atomic_fetch_add(&P.fin[CMR_FIN_CLAIM], lane_o == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        cq = __builtin_amdgcn_readfirstlane(cq);
                                        if (cq < nq_g)
                                            fin_threshold(P.fin_pmax, P.fin_wgs * CMR_SCAN_WAVES, P.k, P.fin_tau, &P.fin[CMR_FIN_READY], stage, cq, lane_o);
                                    }
                                }
                            }
                        }
                        fin_phase = 1;
                    }
                    // (a wave whose first panel ends after the thresholds were published adopts them here and now, not a panel later)
                    bool fin_look = fin_phase == 1;
                    if constexpr (ASMRING && CMR_FIN_HINT) {
                        if (fin_look && fin_hint_valid) {        // (the hint of this panel's last group: see the top of the ring loop)
                            unsigned h_ = fin_hint;
                            asm volatile("" : "+v"(h_));
                            const int hs = __builtin_amdgcn_readfirstlane((int)h_);
                            const int full_ = (int)((1u << nq_g) - 1u);
                            fin_look = hs < 0 || (hs & full_) == full_;      // published (bit 31: a threshold may be there to claim) or complete
                        }
                        fin_hint_valid = false;
                    }
                    if (fin_look) {
                        const int rd = __hip_atomic_load(&P.fin[CMR_FIN_READY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int full = (int)((1u << nq_g) - 1u);
                        if ((rd & full) == full) {
                            const int q = lane_o & 31;
                            const u64 gt = q < nq_g ? __hip_atomic_load(&P.fin_tau[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                            if (q < nq_g && gt > tau_key[0]) { tau_key[0] = gt; tau_f[0] = cmr_key_score(gt); }
                            fin_phase = 2;
#ifdef CMR_FIN_DEBUG
                            if (lane_o == 0) { atomicAdd(&P.fin[CMR_FIN_DBG], 1); atomicAdd(&P.fin[CMR_FIN_DBG + 1], p - p0); CMR_FIN_STAMP_MIN(P, 5); CMR_FIN_STAMP_MAX(P, 6); }      // first / last adoption
#endif
                        } else if (rd < 0) {                      // published, thresholds incomplete: take one
                            int cq = __hip_atomic_fetch_add(&P.fin[CMR_FIN_CLAIM], lane_o == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            cq = __builtin_amdgcn_readfirstlane(cq);
                            if (cq < nq_g) fin_threshold(P.fin_pmax, P.fin_wgs * CMR_SCAN_WAVES, P.k, P.fin_tau, &P.fin[CMR_FIN_READY], stage, cq, lane_o);
                        }
                    }
                    if constexpr (FIN) { if (fin_phase_was_ != 2) { CMR_DBG_ADD(dbg_t_thr) } }
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
        if constexpr (FIN && ASMRING) {
            // The ring prefetched R blocks past the end of the range and hipcc knows nothing of them: the selection code behind
            // this point reuses the ring's registers, so the loads have to land first (the statement holds every slot — nothing
            // of the tail can be scheduled above it; build.py's audit looks for it)
#define CMR_DRAIN_SLOT(u) v4u d##u##_ = buf[(u) < R ? (u) : 0];
            CMR_RING_ALL(CMR_DRAIN_SLOT)
#undef CMR_DRAIN_SLOT
            if constexpr (R == 16)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(d0_), "+v"(d1_), "+v"(d2_), "+v"(d3_), "+v"(d4_), "+v"(d5_), "+v"(d6_), "+v"(d7_), "+v"(d8_),
                             "+v"(d9_), "+v"(d10_), "+v"(d11_), "+v"(d12_), "+v"(d13_), "+v"(d14_), "+v"(d15_) :: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(d0_), "+v"(d1_), "+v"(d2_), "+v"(d3_), "+v"(d4_), "+v"(d5_), "+v"(d6_), "+v"(d7_) :: "memory");
        }
    }

    if constexpr (TOPK) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            const float mn = fminf(rmin[t], __shfl_xor(rmin[t], 32));
            const float mx = fmaxf(rmax[t], __shfl_xor(rmax[t], 32));
            if (lane < 32) {
                const int q = t * 32 + lane;
                P.mm[gwl * NQ + q] = make_float2(mn, mx);
                if constexpr (FIN) fin_mmw[wave * 32 + q] = make_float2(mn, mx);      // (reduced per workgroup in part 2)
                P.cnt[gwl * NQ + q] = __hip_atomic_load(&cnt_w[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    if constexpr (FIN) {
        if (lane == 0) { CMR_FIN_STAMP_MIN(P, 7); CMR_FIN_STAMP_MAX(P, 8); }      // first / last wave out of its scan loop
#ifdef CMR_FIN_DEBUG
        const unsigned dbg_t_p2_ = (unsigned)wall_clock64();
        if (lane == 0) {
            atomicAdd((unsigned*)&P.fin[CMR_FIN_DBG2 + 0], dbg_t_slow); atomicAdd((unsigned*)&P.fin[CMR_FIN_DBG2 + 1], dbg_n_slow);
            atomicAdd((unsigned*)&P.fin[CMR_FIN_DBG2 + 2], dbg_n_whole); atomicAdd((unsigned*)&P.fin[CMR_FIN_DBG2 + 3], dbg_t_thr);
        }
#endif
        // Finishing stage, part 2 — the final selection without a merge launch.  Every wave stages the keys of its lists that beat
        // its FINAL threshold (any threshold a wave holds is a valid lower bound of the global k-th best: nothing that can win is
        // dropped) in its LDS scratch; the workgroup appends them to one dense list per query with ONE device atomic per query,
        // and the last workgroup to arrive (ticket from a counter that re-arms itself) selects the k best per query, one wave
        // per query, and reduces the waves' min / max.  Words that cross workgroups inside the launch travel by device-scope
        // stores / loads (see part 1).  A dense list or a staging area that overflows (thresholds that came late or loose)
        // leaves state 2: the merge launch behind this kernel then works from the per-wave lists as ever — on state 1 it
        // returns at once.
        constexpr int STAGE_KEYS = 128;                      // of the wave's CAP + 2 scratch keys
        constexpr int EPL = CAP / 64;
        int off = 0;
        for (int q0 = 0; q0 < nq_g; q0 += 8) {                // eight lists at a time: one memory round trip, not eight
            u64 lk[8][EPL];
            int lc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = q0 + j;
                int c = 0;
                if (q < nq_g) c = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&cnt_w[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                lc[j] = c < CAP ? c : CAP;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const int i = lane + 64 * e;
                    lk[j][e] = i < lc[j] ? list_w[(size_t)q * CAP + i] : 0ull;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = q0 + j;
                if (q >= nq_g) break;
                const u64 tq = tiny_readlane(tau_key[0], q);
                int kept = 0;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    // >=, not >: a threshold raised by a compaction of this very list IS its k-th best key (the published ones are
                    // a key - 1); empty slots are 0
                    const bool keep = lk[j][e] != 0ull && lk[j][e] >= tq;
                    const u64 m = __ballot(keep);
                    const int slot = off + kept + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (keep && slot < STAGE_KEYS) stage[slot] = lk[j][e];
                    kept += __popcll(m);
                }
                if (lane == 0) __hip_atomic_store(&cnt_w[q], kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // (the list length went to P.cnt above)
                off += kept;
            }
        }
        if (off > STAGE_KEYS && lane == 0) fin_sh[1] = 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int q = wave; q < nq_g; q += CMR_SCAN_WAVES) {
            int total = 0;
            float wmn = __builtin_inff(), wmx = -__builtin_inff();
            for (int w2 = 0; w2 < CMR_SCAN_WAVES; ++w2) {
                total += cnt_all[w2 * NQ + q];
                const float2 v = fin_mmw[w2 * 32 + q];
                wmn = fminf(wmn, v.x); wmx = fmaxf(wmx, v.y);
            }
            if (lane == 0)
                __hip_atomic_store(&P.fin_mm[(size_t)q * gridDim.x + blockIdx.x], ((u64)__float_as_uint(wmx) << 32) | (u64)__float_as_uint(wmn),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (total == 0) continue;
            int run = 0;
            if (lane == 0) run = atomicAdd(&P.fin[CMR_FIN_DCNT(q)], total);
            run = __builtin_amdgcn_readfirstlane(run);
            for (int w2 = 0; w2 < CMR_SCAN_WAVES; ++w2) {
                const int n = cnt_all[w2 * NQ + q];
                if (n == 0) continue;
                int o = 0;
                for (int q2 = 0; q2 < q; ++q2) o += cnt_all[w2 * NQ + q2];
                const u64* src = stage_all + (size_t)w2 * (CAP + 2) + o;
                for (int i = lane; i < n; i += 64)
                    if (o + i < STAGE_KEYS && run + i < P.fin_dcap)
                        __hip_atomic_store(&P.fin_dense[(size_t)q * P.fin_dcap + run + i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                run += n;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // min / max and dense keys are written through before the ticket
        __syncthreads();
        if (tid == 0) {
            if (fin_sh[1]) __hip_atomic_store(&P.fin[CMR_FIN_OVER], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fin_sh[0] = atomicAdd(&P.fin[CMR_FIN_WGS], 1);
            CMR_FIN_STAMP_MIN(P, 9); CMR_FIN_STAMP_MAX(P, 10);      // first / last ticket
#ifdef CMR_FIN_DEBUG
            atomicAdd((unsigned*)&P.fin[CMR_FIN_DBG2 + 4], (unsigned)wall_clock64() - dbg_t_p2_);      // hand-over of this workgroup (its wave 0: scan end -> ticket)
#endif
        }
        __syncthreads();
        if (fin_sh[0] != (int)gridDim.x - 1) return;
        u64* carry = reinterpret_cast<u64*>(smem) + wave * 64;      // the query tile is no longer needed
        // Everything the selection of a query needs is asked for in ONE batch of independent device-scope loads — the overflow word, the
        // list length, the first 1024 dense keys (speculatively: slots beyond the length hold leftovers of earlier launches and are masked
        // by index afterwards) and the workgroups' min / max: one round trip to L2 (2.2 us) where the dependent sequence took four.
        bool over = false;
        for (int q = wave; q < nq_g; q += CMR_SCAN_WAVES) {
            const u64* D = P.fin_dense + (size_t)q * P.fin_dcap;
            const int over_l = __hip_atomic_load(&P.fin[CMR_FIN_OVER], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int total_l = __hip_atomic_load(&P.fin[CMR_FIN_DCNT(q)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u64 key[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = lane + 64 * j;
                key[j] = i < P.fin_dcap ? __hip_atomic_load(&D[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
            // one value per workgroup (<= 512: eight independent loads per lane — a dependent loop over the W per-wave values
            // costs a memory round trip per iteration, 61 of them at W = 3912)
            u64 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int b = lane + 64 * j;
                v[j] = b < (int)gridDim.x ? __hip_atomic_load(&P.fin_mm[(size_t)q * gridDim.x + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : (((u64)0xFF800000u << 32) | 0x7F800000u);      // (-inf, +inf)
            }
            over |= over_l != 0;
            const int total = __builtin_amdgcn_readfirstlane(total_l);
            if (total > P.fin_dcap) { over = true; continue; }
            const int kk = P.k;
            int64_t* oi = P.out_ids + (size_t)q * kk;
            float* os = P.out_scores + (size_t)q * kk;
            const long long idb = P.id_base;
            // (system-scope stores: the caller's buffer may be mapped host memory that the host polls — a plain store is acknowledged
            // before it is visible there, and the done word behind s_waitcnt vmcnt(0) overtook the results: measured)
            auto emit = [&](int r, u64 kv) {
                __hip_atomic_store(&oi[r], kv ? (int64_t)cmr_key_row(kv) + idb : (int64_t)-1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&os[r], kv ? cmr_key_score(kv) : -__builtin_inff(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            };
            if (total <= 1024) {
#pragma unroll
                for (int j = 0; j < 16; ++j) key[j] = lane + 64 * j < total ? key[j] : 0ull;
                tiny_select(key, kk, stage, lane, emit);
            } else {
                tiny_select_stream(total, kk, stage, carry, lane,
                                   [&](int i) -> u64 { return i < total ? __hip_atomic_load(&D[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull; }, emit);
            }
            if (P.out_min || P.out_max) {
                float mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
                for (int j = 0; j < 8; ++j) { mn = fminf(mn, __uint_as_float((unsigned)v[j])); mx = fmaxf(mx, __uint_as_float((unsigned)(v[j] >> 32))); }
#pragma unroll
                for (int off2 = 32; off2 > 0; off2 >>= 1) { mn = fminf(mn, __shfl_xor(mn, off2)); mx = fmaxf(mx, __shfl_xor(mx, off2)); }
                if (lane == 0) {
                    if (P.out_min) __hip_atomic_store(&P.out_min[q], mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (P.out_max) __hip_atomic_store(&P.out_max[q], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        if (nq_g <= wave) over = __hip_atomic_load(&P.fin[CMR_FIN_OVER], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;      // (a wave without a query still reports the word)
        if (over && lane == 0) fin_sh[1] = 2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's results are written through (the caller's buffer may be mapped host memory) before the state says so
        __syncthreads();
#ifdef CMR_FIN_DEBUG
        if (tid == 0) {
            printf("FIN grid %d W %d wgs %d mul %d: done %d ready %d | adoptions %d, panels before adoption (sum) %d | dense[0] %d tau0 %f over %d\n",
                   (int)gridDim.x, W, P.fin_wgs, P.fin_mul, P.fin[CMR_FIN_DONE], P.fin[CMR_FIN_READY], P.fin[CMR_FIN_DBG], P.fin[CMR_FIN_DBG + 1],
                   P.fin[CMR_FIN_DCNT(0)], cmr_key_score(P.fin_tau[0]), fin_sh[1]);
            {
                const unsigned* T = (const unsigned*)&P.fin[CMR_FIN_DBG + 2];
                const unsigned t0 = T[0], te = (unsigned)wall_clock64();
                auto us = [&](unsigned t) { return (t - t0) * 0.01f; };
                printf("FIN us since the first workgroup started: last start %.1f | maxima published %.1f | thresholds %.1f .. %.1f | adoptions %.1f .. %.1f | "
                       "waves out of the scan %.1f .. %.1f | tickets %.1f .. %.1f | selection written %.1f\n",
                       us(T[1]), us(T[2]), us(T[3]), us(T[4]), us(T[5]), us(T[6]), us(T[7]), us(T[8]), us(T[9]), us(T[10]), us(te));
                printf("FIN    first supplying wave through its first panel %.1f | supplying workgroups complete: 1st %.1f, 16th %.1f, 32nd %.1f, 64th %.1f, 96th %.1f, all %d %.1f\n",
                       us(T[11]), us(T[12]), us(T[13]), us(T[14]), us(T[15]), us(T[16]), P.fin_wgs, us(T[2]));
                const unsigned* T2 = (const unsigned*)&P.fin[CMR_FIN_DBG2];
                printf("FIN    per wave (W = %d): slow-path entries %.2f taking %.2f us, whole panels %.2f, threshold logic %.2f us | hand-over per workgroup %.2f us\n", W,
                       (float)T2[1] / W, T2[0] * 0.01f / W, (float)T2[2] / W, T2[3] * 0.01f / W, T2[4] * 0.01f / (int)gridDim.x);
                for (int i = 0; i < 5; ++i) P.fin[CMR_FIN_DBG2 + i] = 0;
                for (int i = 0; i < 17; ++i) P.fin[CMR_FIN_DBG + 2 + i] = (i == 0 || i == 3 || i == 5 || i == 7 || i == 9 || i == 11) ? -1 : 0;
            }
            P.fin[CMR_FIN_DBG] = 0; P.fin[CMR_FIN_DBG + 1] = 0;
        }
#endif
        if (tid < 32) P.fin[CMR_FIN_DCNT(tid)] = 0;
        if (tid == 0) {
            P.fin[CMR_FIN_STATE] = fin_sh[1] == 2 ? 2 : 1;
            P.fin[CMR_FIN_DONE] = 0; P.fin[CMR_FIN_READY] = 0; P.fin[CMR_FIN_WGS] = 0; P.fin[CMR_FIN_OVER] = 0; P.fin[CMR_FIN_CLAIM] = 0; P.fin[CMR_FIN_PUB] = 0;
            if (P.fin_done) __hip_atomic_store(P.fin_done, fin_sh[1] == 2 ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // the launch's last access to the caller's buffer
        }
    }
}

// ------------------------------------------------------------------------------------------ host
static size_t scan_lds_bytes(int nqt, int ks, int cap) {
    return (size_t)nqt * ks * 1024 + (size_t)CMR_SCAN_WAVES * nqt * 32 * 4 + (size_t)CMR_SCAN_WAVES * (cap + 2) * 8 + 16;
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
        // (the limit, not this launch's size: the attribute is per function and host threads launch concurrently —
        // two indexes of different width would otherwise race each other's value)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(g.grid), dim3(CMR_SCAN_THREADS), g.lds, s, p);
        return hipGetLastError();
    };
    if constexpr (MODE == MODE_TOPK) {
        if (g.stream_default_policy) {
            if (g.asm_ring) return launch(scan_kernel<DT, NQT, CAP, R, MODE, 1, 0>);
            return launch(scan_kernel<DT, NQT, CAP, R, MODE, 0, 0>);
        }
    }
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
    } else if constexpr (MODE == MODE_FIN) {
        CASE(1, 128, 8) CASE(1, 128, 16) CASE(1, 256, 8) CASE(1, 256, 16)
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
    p.sample_waves = a.sample_waves; p.sample_stride = a.sample_stride; p.sample_chunk_log2 = a.sample_chunk_log2; p.tau_init = a.tau_init;
    p.slists = a.sample_lists; p.scnt = a.sample_cnt; p.sW = a.sample_W;
    p.qgroups = a.qgroups > 1 ? a.qgroups : 1;
    p.fin_mm = a.fin_mm;
    p.fin = a.fin; p.fin_pmax = a.fin_pmax; p.fin_tau = a.fin_tau; p.fin_dense = a.fin_dense; p.fin_wgs = a.fin_wgs; p.fin_mul = a.fin_mul; p.fin_dcap = a.fin_dcap; p.fin_spin = a.fin_spin; p.fin_done = a.fin_done; p.fin_first = a.fin_first;
    p.out_ids = a.out_ids; p.out_scores = a.out_scores; p.out_min = a.out_min; p.out_max = a.out_max; p.id_base = a.id_base;
    return p;
}

// ------------------------------------------------------------------------------------------
// Wide-batch scan: up to 256 queries in ONE pass over the corpus (BASELINE config 3's batch-256).
// The LDS-resident query tile of scan_kernel tops out at 64 queries (96 KiB); here the queries live
// in REGISTERS.  A workgroup is 4 waves, ONE per SIMD, so a wave owns the SIMD's whole 512-entry
// register file (VGPR + AGPR, unified on gfx950; MFMA B-operands may be AGPRs): it keeps the MFMA
// B-operands of NT = 2 query tiles resident (2 x 48 blocks x 4 registers = 384 at 768-d; NT = 1 at
// 1024-d) and 4 waves cover 4*NT*32 = 256 (128) queries.
//   * All waves consume the same corpus blocks: HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, one
//     wave-instruction moves one 1-KiB block, lane-linear = exactly the block layout) into a ring of
//     NSTG groups of GRP blocks; NSTG-1 groups stay in flight per workgroup and no VGPR is spent on
//     the stream.  Every wave reads each block back ONCE (ds_read_b128, ADEPTH blocks ahead, the
//     read-ahead runs across group and panel boundaries) and feeds it to NT MFMAs, one per tile:
//     two independent accumulator chains interleave, so no MFMA waits for its predecessor, while
//     each query's own fp32 chain keeps the k order of scan_kernel (bit-identical scores).
//   * per group g:  s_waitcnt vmcnt(PPG*(NSTG-3))   my DMA pieces of group g+1 have landed
//                   s_barrier                        everybody's have; everybody left group g-1
//                   GRP x { NT MFMAs, ds_read_b128 ADEPTH blocks ahead },
//                   with the PPG DMA pieces of group g+NSTG-1 -> stage (g-1) mod NSTG spread between them
//     Counted waits + raw s_barrier (__syncthreads() would drain the DMA queue, guide §5); the DMAs
//     are inline asm (§5.7 recipe: M0 = LDS destination, saved/restored inside the statement) with a
//     wave-uniform SGPR base, so hipcc neither counts them nor drains them before LDS reads.
//   * Budget per CU and 32-row panel at 768-d: 4 SIMDs x 96 MFMAs x 32 cycles = 3072 cycles, 192 KiB
//     of ds_read_b128 = 768 cycles, 48 KiB of HBM = ~4900 cycles at the achievable rate: HBM-bound with
//     the matrix pipe ~63 % busy.
//   * A workgroup scans the (sampled) panels s in [s0, s1): all panels for the main pass; for a sampling pass chunks of
//     consecutive panels spread over the corpus, MANY per workgroup (loading the 384 KiB of query fragments is the fixed
//     cost of a workgroup).  The prefetch cursor is clamped to the workgroup's last group, so nothing outside its
//     panels is ever read.
// The top-k epilogue, candidate lists and thresholds are the per-wave ones of scan_kernel; list /
// counter rows are laid out [workgroup][4*NT*32 queries] so merge_query_kernel consumes them with
// W = gridDim.x.
#define WIDE_WAVES 4
#ifndef WIDE_GROUP
#define WIDE_GROUP 16     // blocks per staged group where 24 does not divide a row's k-steps (1024-d: 64), and for the 8-wave variant
#endif
#ifndef WIDE4_NST
#define WIDE4_NST 8       // stages of the LDS ring at WIDE_GROUP (4-wave kernel): WIDE4_NST x WIDE_GROUP KiB, one stage being refilled
#endif
#ifndef WIDE_GROUP3
#define WIDE_GROUP3 24    // blocks per staged group where 24 divides a row's k-steps (768-d: 48 = two groups, two barriers per panel
#endif                    // instead of three: 3.907 -> 3.853 ms at 10 M rows, profiles/r3_measurements.md; 8-KiB groups cost +6 %)
#ifndef WIDE4_NST3
#define WIDE4_NST3 5      // stages of the ring at WIDE_GROUP3 (120 KiB; 6 stages with half the staging records measured no better)
#endif
constexpr int wide_group(int ks, int nw) { return (nw == 4 && ks % WIDE_GROUP3 == 0) ? WIDE_GROUP3 : WIDE_GROUP; }
constexpr int wide_nst4(int ks) { return ks % WIDE_GROUP3 == 0 ? WIDE4_NST3 : WIDE4_NST; }
#ifndef WIDE_ADEPTH
#define WIDE_ADEPTH 8     // LDS read-ahead ring of a wave, in blocks (two quads: one in use, one landing)
#endif

// Epilogue pieces of the wide kernel.  The wave's register file is full of query fragments, and hipcc's allocator
// spills the values with the longest live range first — the fragments — whenever ANY block of the loop needs more
// registers than are free, so the epilogue is written to need a handful: 32-bit row arithmetic, accumulator values
// taken four at a time, and the pushes kept sequential (sched_barrier) instead of sixteen in flight.
__device__ __forceinline__ float wide_max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));   // no canonicalising v_max(x, x) per asm-produced input
    return d;
}
__device__ __forceinline__ float wide_min3(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// panel max (returned) and running min / max of one tile's 16 scores per lane; gmax[i] = max of accumulator registers
// 4i .. 4i+3 (the slow path only looks into the quarters that beat the threshold)
__device__ __forceinline__ float wide_minmax(const f32x16& acc, float& rmin, float& rmax, float (&gmax)[4]) {
    float mn = rmin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * i;
        gmax[i] = wide_max3(acc[r], acc[r + 1], acc[r + 2]);
        mn = wide_min3(mn, acc[r], acc[r + 1]);
        gmax[i] = wide_max3(gmax[i], acc[r + 3], acc[r + 3]);
        mn = wide_min3(mn, acc[r + 2], acc[r + 3]);
        __builtin_amdgcn_sched_barrier(0);
    }
    rmin = mn;
    const float mx = wide_max3(wide_max3(gmax[0], gmax[1], gmax[2]), gmax[3], gmax[3]);
    rmax = wide_max3(rmax, mx, mx);
    return mx;
}
// One quarter of wide_minmax, issued BETWEEN two MFMAs of the other tile (NT = 2): four accumulator values folded into the
// panel max / min.  With one wave per SIMD nothing else can use the matrix pipe while this wave runs VALU code, so the
// epilogue is software-pipelined into the MFMA stream: an independent MFMA occupies the pipe for 8 issue slots, seven of
// which are free for these eight instructions.
__device__ __forceinline__ void wide_epi_piece(const f32x16& acc, int i, float& gmax, float& mn) {
    // one statement: hipcc would otherwise read every accumulator value twice (one AGPR -> VGPR copy per consumer) and
    // keep all sixteen alive for the (cold) partial-panel branch
    float a0, a1, a2, a3;
    asm volatile("v_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7\n\tv_accvgpr_read_b32 %4, %8\n\tv_accvgpr_read_b32 %5, %9\n\t"
                 "v_max3_f32 %0, %2, %3, %4\n\tv_min3_f32 %1, %1, %2, %3\n\tv_max_f32 %0, %0, %5\n\tv_min3_f32 %1, %1, %4, %5"
                 : "=&v"(gmax), "+v"(mn), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                 : "a"(acc[4 * i]), "a"(acc[4 * i + 1]), "a"(acc[4 * i + 2]), "a"(acc[4 * i + 3]));
    __builtin_amdgcn_sched_barrier(0);
}
// accumulators in VGPRs: the same four-value fold without the reads
__device__ __forceinline__ void wide_epi_piece_v(const f32x16& acc, int i, float& gmax, float& mn) {
    asm volatile("v_max3_f32 %0, %2, %3, %4\n\tv_min3_f32 %1, %1, %2, %3\n\tv_max_f32 %0, %0, %5\n\tv_min3_f32 %1, %1, %4, %5"
                 : "=&v"(gmax), "+v"(mn)
                 : "v"(acc[4 * i]), "v"(acc[4 * i + 1]), "v"(acc[4 * i + 2]), "v"(acc[4 * i + 3]));
    __builtin_amdgcn_sched_barrier(0);
}
// the same for the corpus' last, partial panel: rows >= nvalid are padding
__device__ __forceinline__ float wide_minmax_partial(const f32x16& acc, int nvalid, int lane, float& rmin, float& rmax) {
    float mx = -__builtin_inff(), mn = rmin;
    int lim = nvalid - 4 * (lane >> 5);          // accumulator register r holds panel row crow(r) + 4 * (lane >> 5)
    asm volatile("" : "+v"(lim));                 // not loop-invariant for hipcc: stays inside the (cold) caller branch
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool ok = (r & 3) + 8 * (r >> 2) < lim;
        const float v = acc[r];
        mx = wide_max3(mx, ok ? v : -__builtin_inff(), mx);
        mn = wide_min3(mn, ok ? v : __builtin_inff(), mn);
        __builtin_amdgcn_sched_barrier(0);
    }
    rmin = mn;
    rmax = wide_max3(rmax, mx, mx);
    return mx;
}
// Candidate push of the wide kernel.
//  * The filter is the float compare v >= tau_f alone — a superset of key > tau_key (it also admits rows that tie with
//    the threshold score): any real (score, row) pair may sit in a candidate list, only the keys above the threshold must.
//  * vmcnt counts every vector-memory instruction of the wave IN ORDER, so each global store issued here makes the
//    scan's hand-counted "s_waitcnt vmcnt(N)" wait for one more DMA piece than it needs; a sampling pass (threshold from
//    a 640-row sample: ~1 pass per query and panel, 28 store instructions per wave and panel) drained the whole ring on
//    every panel that way (10-37 us per panel).  So: count first, ONE LDS atomic per lane reserves the lane's list slots
//    and its slots in a per-wave LDS staging area, the keys go to the staging area as (key, destination) records, and the
//    wave then stores the records cooperatively — ceil(T / 64) store instructions for T pushes, usually one.  The number
//    of store instructions is returned in n_stores; the scan adds it to its wait counts (wide_wait_group).
//  * 32-bit rows.  Everything derived from the lane id is made opaque here: hipcc would otherwise hoist sixteen
//    per-register row offsets and the list pointers out of the panel loop, as loop-invariant values that then live
//    across the whole scan (and evict query fragments).
#ifndef WIDE_STG
#define WIDE_STG 256      // staging records (16 B) per wave (4-wave kernel)
#endif
// 8-wave kernel: k-steps of a tile whose B-operand is served from LDS instead of a register (the wave has 256 registers:
// 192 of fragments + 16 accumulators + a 16-register read-ahead ring leave hipcc too few for everything else)
#ifndef WIDE8_KLDS
#define WIDE8_KLDS 8
#endif
#ifndef WIDE8_NST
#define WIDE8_NST 5
#endif
#define WIDE8_STG 64
#define WIDE_AMOVE 8      // NT = 2: k-steps of tile 0 whose B-operand lives in the AGPR half
template <int CAP, int STG>
__device__ __forceinline__ u64 wide_push(const f32x16& acc, unsigned row0, int nvalid, float tau_f, int* cnt_t, u64* list_t, uint4* stg,
                                         int* stg_tail, int lane, int& n_stores) {
    int ql = lane & 31;
    int hrow = 4 * (lane >> 5);
    asm volatile("" : "+v"(ql), "+v"(hrow), "+v"(lane));
    const unsigned rbase = row0 + (unsigned)hrow;
    const int lim = nvalid - hrow;
    int n = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        n += ((r & 3) + 8 * (r >> 2) < lim && acc[r] >= tau_f) ? 1 : 0;
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (lane == 0) *stg_tail = 0;                       // LDS operations of one wave execute in order
    int slot = 0, pos = 0;
    if (n) {
        slot = atomicAdd(&cnt_t[ql], n);                // ds_add_rtn_u32; <= 32 pushes per query per panel (two lanes x 16)
        pos = atomicAdd(stg_tail, n);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int T = __builtin_amdgcn_readfirstlane(*stg_tail);
    if (T <= STG) {
        unsigned dsti = (unsigned)ql * CAP + (unsigned)slot;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r];
            const int cr = (r & 3) + 8 * (r >> 2);
            if (cr < lim && v >= tau_f) {
                const u64 key = cmr_make_key(v, rbase + (unsigned)cr);
                stg[pos++] = make_uint4((unsigned)key, (unsigned)(key >> 32), dsti++, 0u);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma nounroll
        for (int i = lane; i < T; i += 64) {
            const uint4 rec = stg[i];
            list_t[rec.z] = ((u64)rec.y << 32) | (u64)rec.x;
        }
        n_stores += (T + 63) >> 6;
    } else {                                            // more pushes than the staging area holds: direct stores
        u64* dst = list_t + (size_t)ql * CAP;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r];
            const int cr = (r & 3) + 8 * (r >> 2);
            if (cr < lim && v >= tau_f) dst[slot++] = cmr_make_key(v, rbase + (unsigned)cr);
            __builtin_amdgcn_sched_barrier(0);
        }
        n_stores += 16;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // as topk_push: counters current, pushed keys fenced by the compaction
    const int c = __hip_atomic_load(&cnt_t[ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __ballot(c > CAP - 32) & 0xFFFFFFFFull;
}
// The same push for the MAIN pass, where a tile that beats the threshold does so with one or two values (the threshold
// comes from a 300 K-row sample: ~640 of a query's 10 M scores pass, 2-3 per workgroup).  At 256 queries per workgroup
// one panel in two has such an event in SOME wave and the other three wait for it at the next group barrier, so the
// event is kept short: only the accumulator quarters whose maximum (gmax, from the epilogue's fold) reaches the
// threshold are read back and scanned, one wave-uniform test per register, key compare for the ties, one returning LDS
// atomic and one store per pushed value; n_stores counts the store instructions.
template <int CAP, bool ASMREAD>
__device__ __forceinline__ u64 wide_push_sparse(const f32x16& acc, const float (&gmax)[4], unsigned row0, int nvalid, u64 tau_key, float tau_f,
                                                int* cnt_t, u64* list_t, int lane, int& n_stores) {
    int ql = lane & 31;
    int hrow = 4 * (lane >> 5);
    asm volatile("" : "+v"(ql), "+v"(hrow));
    const unsigned rbase = row0 + (unsigned)hrow;
    const int lim = nvalid - hrow;
    int top = 0;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
        if (__any(gmax[gi] >= tau_f)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * gi + j;
                float v;
                // ASMREAD (software-pipelined NT = 2 kernel): the read stays inside this branch; otherwise the caller's fold
                // already holds the sixteen values in VGPRs
                if constexpr (ASMREAD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[r]));
                else v = acc[r];
                if (__any(v >= tau_f)) {
                    const int cr = (r & 3) + 8 * (r >> 2);
                    const u64 key = cmr_make_key(v, rbase + (unsigned)cr);
                    const bool hit = v >= tau_f && cr < lim && key > tau_key;
                    n_stores += __any(hit) ? 1 : 0;
                    if (hit) {
                        const int slot = atomicAdd(&cnt_t[ql], 1);  // ds_add_rtn_u32; <= 32 pushes per query per panel
                        list_t[(size_t)ql * CAP + slot] = key;
                        top = slot + 1;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // the atomics returned each pushing lane's slot: a list is nearly full when some push landed beyond CAP - 32
    const u64 full = __ballot(top > CAP - 32);
    return (full | (full >> 32)) & 0xFFFFFFFFull;
}
// "My DMA pieces of the next group have landed": at most `base` younger DMA pieces plus the x store instructions this
// wave issued after them may still be in flight.  The wait count is an immediate, so x selects among a few (rounded
// down: waiting for fewer outstanding operations than allowed is always safe).
template <int BASE>
__device__ __forceinline__ void wide_wait_group(int x) {
    static_assert(BASE + 32 <= 63, "vmcnt is a 6-bit counter");
    // x == 0 (no slow-path store since the pieces were issued) is the case of nearly every group: ONE compare and an
    // untaken branch in front of its wait.  Left as a plain if / else-if chain hipcc lowers all seven cases into one
    // binary search tree — a dozen SALU instructions and a taken branch on the common path, at the one point of a group
    // where the matrix pipe is about to run dry (the opaque copy keeps the x == 0 test out of that tree).
    if (__builtin_expect(x == 0, 1)) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE) : "memory"); return; }
    asm volatile("" : "+s"(x));
    if (x == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 1) : "memory");
    else if (x < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 2) : "memory");
    else if (x < 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 4) : "memory");
    else if (x < 16) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 8) : "memory");
    else if (x < 32) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 16) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 32) : "memory");
}
// topk_compact for the wide kernel: the same rank-by-counting compaction, but every list element goes through the
// LDS stage instead of being held in registers, and no loop is unrolled — about a dozen registers in all.  (An
// out-of-line call is no way out: the callee's clobber set keeps the caller's long-lived fragments out of v0..v65.)
template <int CAP>
__device__ __forceinline__ void wide_compact(u64 need, int k, u64& tau_key, float& tau_f, int* cnt_t, u64* list_t, u64* stage, int lane) {
    int ql = lane & 31;
    asm volatile("" : "+v"(ql), "+v"(lane));
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // pushes landed (same-CU L1 is coherent)
    while (need) {
        const int j = __ffsll((long long)need) - 1;
        need &= need - 1;
        const int n = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&cnt_t[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        u64* L = list_t + (size_t)j * CAP;
#pragma nounroll
        for (int i = lane; i < n; i += 64) stage[i] = L[i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma nounroll
        for (int i = lane; i < n; i += 64) {
            const u64 e = stage[i];
            int rk = 0;
#pragma nounroll
            for (int jj = 0; jj < n; ++jj) rk += (stage[jj] > e) ? 1 : 0;     // uniform address: LDS broadcast
            if (rk < k) L[rk] = e;                 // keys are unique: ranks are a permutation
            if (rk == k - 1) stage[CAP] = e;
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

// ABL: developer ablations (development builds, -DCMR_DEV_KNOBS, option "wide_abl"; results are wrong by design): 1 no MFMA, 2 no DMA in the loop, 3 no epilogue,
// 4 min/max but no threshold test / slow path, 5 no barrier in the loop, 6 = 2 + no LDS reads, 7 = 2 + no barrier
// NW = waves per workgroup.  4: one wave per SIMD, the whole 512-entry register file, NT = 2 tiles per wave at 768-d.
// 8: TWO waves per SIMD (256 registers each), one tile per wave (192 registers of fragments at 768-d) — the same 256 queries
// per workgroup, but every SIMD now has a partner wave whose MFMAs fill the matrix pipe while the other runs its epilogue
// or sits at a wait / barrier; the price is that every block is read from LDS by eight waves instead of four (128 B/clk of
// the LDS's 256 B/clk ds_read_b128 rate at full MFMA rate, guide §LDS) and a 4-block read-ahead ring instead of 8.
template <int DT, int KS, int NT, int CAP, int NSTG, int KLDS, int ABL = 0, int NW = WIDE_WAVES>
__global__ __launch_bounds__(NW * 64, 1) void scan_wide_kernel(ScanP P) {
    constexpr int GRP = wide_group(KS, NW), NST = NSTG, ADEPTH = NW == 8 ? 4 : (wide_group(KS, NW) % WIDE_ADEPTH == 0 ? WIDE_ADEPTH : 8);
    constexpr int STG = NW == 8 ? WIDE8_STG : WIDE_STG;         // staging records per wave (LDS budget)
    static_assert(NW == 4 || (NW == 8 && NT == 1), "8 waves hold one tile each");
    static_assert(NW == 4 || (size_t)STG * 16 <= (size_t)(CAP + 2) * 8, "8-wave kernel: the staging records live in the compaction stage");
    static_assert(KS % GRP == 0 && GRP % NW == 0 && GRP % ADEPTH == 0 && ADEPTH % 4 == 0 && NST >= 4, "group geometry (a 3-stage ring was measured with a -DWIDE4_NST3=3 build in round 5 and not kept: no shipped or tested configuration has fewer than 4 stages)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NQB = NW * NT * 32;             // queries per workgroup pass
    constexpr int GPP = KS / GRP;                 // groups per panel
    constexpr int PPG = GRP / NW;                 // DMA pieces per wave per group
    constexpr int QPP = (GRP / 4) / PPG;          // quads of blocks per DMA piece of a wave (1 at 4 waves, 2 at 8)
    static_assert(PPG * QPP == GRP / 4, "DMA pieces spread evenly over the quads of a group");
    constexpr int KREG = KS - KLDS;               // k-steps of a tile resident in registers; the last KLDS sit in LDS
    constexpr bool ACCV = NT == 2 && NW == 4 && CMR_WIDE_ACC_VGPR;     // accumulators in the VGPR half (see CMR_WIDE_ACC_VGPR)
    // the LDS-tail branch of the MFMA chain (ks >= KREG) issues mma_asm with an AGPR accumulator: with ACCV the compiler would wrap every
    // such MFMA in accvgpr copies around opaque asm (no wait states padded).  No NT = 2 / NW = 4 configuration has an LDS tail today.
    static_assert(!ACCV || KLDS == 0, "accumulators in VGPRs need every k-step's B-operand in registers (KLDS = 0)");
    constexpr int AMOVE = ACCV ? 16 : WIDE_AMOVE;                       // NT = 2: k-steps of tile 0 whose B-operand lives in the AGPR half
// an empty / nop statement that makes an accumulator opaque at this point, whichever register file holds it
#define CMR_ACC_ASM(TEXT, C) do { if constexpr (ACCV) asm volatile(TEXT : "+v"(C)); else asm volatile(TEXT : "+a"(C)); } while (0)
    constexpr int VK = NW == 8 ? (KS - KLDS) - 28 : KS / 2;     // NT = 1: k-steps whose B-operand lives in the VGPR half (the rest: AGPRs; hipcc splits a 256-register budget 128 / 128: 16 accumulators + 28 k-steps fill the AGPR half exactly)

    v4u* stage_lds = reinterpret_cast<v4u*>(smem);                                    // [NST][GRP][64]
    int* cnt_all = reinterpret_cast<int*>(smem + NST * GRP * 1024);                   // [NQB]
    u64* cstage_all = reinterpret_cast<u64*>(cnt_all + NQB);                          // [WAVES][CAP+2]
    v4u* qlds_all = reinterpret_cast<v4u*>(cstage_all + NW * (CAP + 2));             // [WAVES][NT][KLDS][64]
    // (8-wave kernel: the staging records of a push are dead once its cooperative stores were issued, and a compaction only
    //  ever follows a push: the two per-wave scratch areas share their LDS — WIDE8_STG records = CAP + 2 keys at CAP = 128)
    uint4* stg_all = NW == 8 ? reinterpret_cast<uint4*>(cstage_all) : reinterpret_cast<uint4*>(qlds_all + (size_t)NW * NT * KLDS * 64);   // [WAVES][STG] staging records
    int* tail_all = reinterpret_cast<int*>(NW == 8 ? reinterpret_cast<unsigned char*>(qlds_all + (size_t)NW * NT * KLDS * 64) : reinterpret_cast<unsigned char*>(stg_all + (size_t)NW * STG));   // [WAVES]
    v4u* qlds = qlds_all + (size_t)wave * NT * KLDS * 64 + lane;
    uint4* stg = NW == 8 ? reinterpret_cast<uint4*>(cstage_all + wave * (CAP + 2)) : stg_all + (size_t)wave * STG;
    int* stg_tail = tail_all + wave;
    int* cnt_w = cnt_all + wave * NT * 32;
    u64* cstage = cstage_all + wave * (CAP + 2);
    for (int i = tid; i < NQB; i += NW * 64) cnt_all[i] = 0;

    // this wave's query fragments -> registers (static indices everywhere below)
    v4u qreg[NT][KREG];
    if constexpr (NW == 8) {
        // 256 registers per wave: the fragment loads are inline asm with a wave-uniform SGPR base, one lane offset and the
        // destination's register file fixed by the constraint — compiler-generated loads build a 64-bit VGPR address per
        // few fragments, and that prologue peak alone spills fragments for the whole kernel
        const unsigned loff = (unsigned)lane * 16u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const char* qb = reinterpret_cast<const char*>(P.qfrag) + ((size_t)wave * KS + (size_t)(ks & ~3)) * 1024;   // wave-uniform
            if (ks < KREG) {
                if (ks >= VK) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(qreg[0][ks < KREG ? ks : 0]) : "v"(loff), "s"(qb), "n"((ks & 3) * 1024) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(qreg[0][ks < KREG ? ks : 0]) : "v"(loff), "s"(qb), "n"((ks & 3) * 1024) : "memory");
            } else {
                v4u v;
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(loff), "s"(qb), "n"((ks & 3) * 1024) : "memory");
                qlds[(ks - KREG) * 64] = v;                       // written and read by the same lane only
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4u v = P.qfrag[((size_t)(wave * NT + t) * KS + ks) * 64 + lane];
                if (ks < KREG) qreg[t][ks < KREG ? ks : 0] = v;
                else qlds[(t * KLDS + (ks - KREG)) * 64] = v;        // written and read by the same lane only
            }
    }
    // Make hipcc retire every query-fragment load HERE: left alone it defers each wait to the
    // fragment's first use inside the panel loop, where stale low-count s_waitcnt vmcnt(N) would
    // drain the hand-counted DMA ring on every iteration.
    // (An explicit wait rather than an asm "v" pin of every fragment: a "v" constraint would force the fragments into
    // the VGPR half of the file, and hipcc then shuttles them through v_accvgpr_read in front of every MFMA.)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched

    // panels of this workgroup: s in [s0, s1) of the (sampled) panel sequence; panel_of(s) is the corpus panel
    const int nb = gridDim.x;
    const int clog = P.sample_waves > 0 ? P.sample_chunk_log2 : 0;
    const int cstride = P.sample_waves > 0 ? P.sample_stride : 1;
    const int S = P.sample_waves > 0 ? P.sample_waves : P.npanels;
    const int s0 = (int)(((long long)blockIdx.x * S) / nb);
    const int s1 = (int)(((long long)(blockIdx.x + 1) * S) / nb);
    auto panel_of = [&](int s) -> unsigned { return (unsigned)(s >> clog) * (unsigned)cstride + (unsigned)(s & ((1 << clog) - 1)); };

    float rmin[NT], rmax[NT], tau_f[NT];
    u64 tau_key[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        rmin[t] = __builtin_inff(); rmax[t] = -__builtin_inff(); tau_f[t] = -__builtin_inff();
        tau_key[t] = 0ull;
        const int q = (wave * NT + t) * 32 + (lane & 31);
        if (q >= P.nq) {                         // padding query (all-zero operand): nothing may pass
            tau_key[t] = ~0ull;
            tau_f[t] = __builtin_inff();
        } else if (P.tau_init) {                 // a valid lower bound on the global k-th best key
            tau_key[t] = P.tau_init[q];
            if (tau_key[t]) tau_f[t] = cmr_key_score(tau_key[t]);
        }
    }
    u64* list_w = P.lists + ((size_t)blockIdx.x * NQB + (size_t)wave * NT * 32) * CAP;
    __syncthreads();

    if (s1 > s0) {
        const char* cbase = reinterpret_cast<const char*>(P.corpus);
        auto panel_src = [&](int s) -> const char* { return cbase + (unsigned long long)panel_of(s) * (KS * 1024ull); };   // wave-uniform
        const char* last_group = panel_src(s1 - 1) + (GPP - 1) * GRP * 1024;
        const unsigned voff = (unsigned)lane * 16u + (unsigned)wave * 1024u;                    // piece j of wave w = block j*WAVES + w of its group
        const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) + (unsigned)wave * 1024u;
        // one DMA piece: 1 KiB from (wave-uniform base + voff) to LDS byte address dst
        auto dma_piece = [&](const char* base, unsigned dst) {
#if CMR_WIDE_M0_CLOBBER
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" CMR_STREAM_POLICY
                         :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
#else
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" CMR_STREAM_POLICY "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
#endif
        };
#if CMR_WIDE_DMA_IMM
        // Piece j of a group WITHOUT compiler arithmetic around it (4-wave kernel: pieces are 4 KiB apart): the group's source base and
        // LDS destination are two SGPR operands shared by all its pieces; the piece's own displacement is split between three
        // loop-invariant lane offsets (voff + 4 KiB, + 12 KiB, + 20 KiB) and the instruction's signed 13-bit immediate (-4096 / 0), M0
        // takes destination + j * 4 KiB inside the statement.  Three issue slots per piece, none of them hipcc's to place: left to the
        // compiler, the 64-bit source add and the destination add of every piece land in the gaps of the neighbouring MFMA chain.
        // The instruction's immediate displaces BOTH addresses of an LDS-DMA — the global source and the LDS destination (M0 + offset
        // + lane * 16; measured: without the correction every second piece lands 4 KiB low and the results differ) — so M0 is set to
        // destination + j * 4 KiB MINUS the immediate.
        unsigned voj0 = voff + 4096u, voj1 = voff + 3u * 4096u, voj2 = voff + 5u * 4096u;
        asm volatile("" : "+v"(voj0), "+v"(voj1), "+v"(voj2));      // three registers, not three adds per use
#define CMR_WIDE_PIECE(J, VO)                                                                                                          \
    asm volatile("s_add_i32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%4" CMR_STREAM_POLICY                      \
                 :: "v"(VO), "s"(gb), "s"(gd), "n"((J) * 4096 - (((J) & 1) ? 0 : -4096)), "n"(((J) & 1) ? 0 : -4096) : "memory", "m0", "scc")
        auto dma_piece_j = [&](int j, const char* gb, unsigned gd) __attribute__((always_inline)) {
            static_assert(NW != 4 || PPG <= 6, "six pieces per group at most");
            switch (j) {            // j is a constant after unrolling: one statement survives
                case 0: CMR_WIDE_PIECE(0, voj0); break;
                case 1: CMR_WIDE_PIECE(1, voj0); break;
                case 2: CMR_WIDE_PIECE(2, voj1); break;
                case 3: CMR_WIDE_PIECE(3, voj1); break;
                case 4: CMR_WIDE_PIECE(4, voj2); break;
                default: CMR_WIDE_PIECE(5, voj2); break;
            }
        };
#undef CMR_WIDE_PIECE
#endif
        // prologue: groups 0 .. NST-2 of this workgroup's stream (clamped to its last group)
#pragma unroll
        for (int d = 0; d < NST - 1; ++d) {
            const int dp = s0 + d / GPP;
            const char* src = dp < s1 ? panel_src(dp) + (d % GPP) * GRP * 1024 : last_group;
#pragma unroll
            for (int j = 0; j < PPG; ++j) dma_piece(src + j * NW * 1024, lds_base + (unsigned)(d * GRP + j * NW) * 1024u);
        }
        // group 0 must be complete before the first reads; from then on the barrier of group g validates
        // group g+1, so the LDS read-ahead never has to stop at a group or panel boundary
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPG * (NST - 2)) : "memory");
        int st = 0;                                   // stage holding the current group
        int st_new = 0, st_old = 0;                   // store instructions this wave issued in the last / second-to-last epilogue
        const v4u* buf = stage_lds + lane;            // current stage, this lane's slot
        v4u a[ADEPTH];
#pragma unroll
        for (int u = 0; u < ADEPTH; ++u) a[u] = buf[u * 64];

        // Epilogue schedule.  NT = 1: after the panel's MFMAs (drain, min / max, threshold test).  NT = 2: software-pipelined
        // into the MFMA stream, because with one wave per SIMD the matrix pipe idles whenever this wave runs VALU code —
        // tile 0's min / max sits between tile 1's last four MFMAs of the panel, tile 1's between the first four tile-0
        // MFMAs of the NEXT panel (tile 1's accumulators stay untouched until that quad's own tile-1 MFMAs), the last
        // panel's after the loop.  No accumulator is duplicated; the threshold test and the (rare) slow path follow the
        // fold at once, so a tile's pushes still precede the next panel's compare for that tile.
        f32x16 acc[NT];
        if constexpr (ACCV) asm volatile("" : "=v"(acc[1]));
        else if constexpr (NT == 2) asm volatile("" : "=a"(acc[1]));   // the very first quad folds (and discards) whatever is there
        float eg[4] = {0.0f, 0.0f, 0.0f, 0.0f}, emn = 0.0f;   // quarter maxima / panel min under construction by the interleaved pieces
        unsigned prow0 = 0;                           // the previous panel: tile 1's epilogue is still pending
        int pnvalid = CMR_PANEL_ROWS;
        int st_mid = 0;                               // NT = 2: store instructions of tile 0's epilogue of the previous panel
        // fold + threshold test + slow path of tile tc; FOLDED: emx / emn already hold the panel's max / min
        auto epi_finish = [&](auto tc, auto folded, unsigned row0, int nvalid, int& n_stores, bool& compacted) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            float mx;
            if (__builtin_expect(nvalid < CMR_PANEL_ROWS, 0)) {        // only the corpus' last panel: masked re-computation
                asm volatile("" : "+s"(nvalid));
                CMR_ACC_ASM("", acc[t]);                               // keeps the masked variant's arithmetic and accumulator reads inside this branch
                mx = wide_minmax_partial(acc[t], nvalid, lane, rmin[t], rmax[t]);
                eg[0] = eg[1] = eg[2] = eg[3] = mx;                    // the slow path looks everywhere
            } else if (decltype(folded)::value) {
                mx = wide_max3(wide_max3(eg[0], eg[1], eg[2]), eg[3], eg[3]);
                rmin[t] = wide_min3(rmin[t], emn, emn);
                rmax[t] = wide_max3(rmax[t], mx, mx);
            } else {
                mx = wide_minmax(acc[t], rmin[t], rmax[t], eg);
            }
            if (ABL != 4 && __any(mx >= tau_f[t])) {
                // (opaque: hipcc must not hoist the slow path's sixteen accumulator reads in front of the test)
                if constexpr (decltype(folded)::value) CMR_ACC_ASM("", acc[t]);
                // sampling pass: dense hits (threshold from a small sample) -> staged push; main pass: sparse hits
                const u64 need = P.sample_waves > 0
                    ? wide_push<CAP, STG>(acc[t], row0, nvalid, tau_f[t], cnt_w + t * 32, list_w + (size_t)t * 32 * CAP, stg, stg_tail, lane, n_stores)
                    : wide_push_sparse<CAP, decltype(folded)::value && !ACCV>(acc[t], eg, row0, nvalid, tau_key[t], tau_f[t], cnt_w + t * 32, list_w + (size_t)t * 32 * CAP, lane, n_stores);
                if (need) {
                    wide_compact<CAP>(need, P.k, tau_key[t], tau_f[t], cnt_w + t * 32, list_w + (size_t)t * 32 * CAP, cstage, lane);
                    compacted = true;
                }
            }
        };
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, NT - 1>;

        for (int s = s0; s < s1; ++s) {
            const unsigned row0 = panel_of(s) * CMR_PANEL_ROWS;                         // < 2^32 rows per shard (cmr_index_append)
            int nvalid = CMR_PANEL_ROWS;
            if (__builtin_expect((long long)row0 + CMR_PANEL_ROWS > P.nrows, 0)) nvalid = (int)(P.nrows - (long long)row0);
#pragma unroll
            for (int g = 0; g < GPP; ++g) {
                // The DMA pieces of group g+1 were issued NST-2 groups (two panels) ago; the store instructions issued since
                // are younger than they are: NT = 1 the last two epilogues'; NT = 2 the same, except that at g = 0 tile 1's
                // epilogue of the previous panel has not run yet (it sits in this group's first quad).
                if constexpr (ABL != 2 && ABL != 6 && ABL != 7) wide_wait_group<PPG * (NST - 3)>(st_old + ((NT == 2 && g == 0) ? st_mid : st_new));
                if constexpr (ABL != 5 && ABL != 7) asm volatile("s_barrier" ::: "memory");
                // prefetch cursor: group g + NST-1 of the stream, into the stage everybody just left
                const int dps = s + (g + NST - 1) / GPP;
                const char* dsrc = dps < s1 ? panel_src(dps) + ((g + NST - 1) % GPP) * GRP * 1024 : last_group;
                const unsigned ddst = lds_base + (unsigned)(st == 0 ? NST - 1 : st - 1) * (GRP * 1024u);
                const int stn = st + 1 == NST ? 0 : st + 1;
                const v4u* bufn = stage_lds + (size_t)stn * GRP * 64 + lane;
                // Issue order inside a quad of four blocks: the four MFMAs of tile 0, then the four of tile 1 (each tile's
                // own chain stays in k order).  Alternating the two accumulators MFMA by MFMA is the slowest pattern the
                // matrix pipe has (tools/probe/mfma_probe.hip: 1.45 PFLOP/s chip-wide against 1.83 for runs of four).
#pragma unroll
                for (int qd = 0; qd < GRP / 4; ++qd) {
                    constexpr bool DMA_MID = NT == 2 && CMR_WIDE_DMA_MID && !CMR_WIDE_DMA_BURST;
                    auto quad_dma = [&]() __attribute__((always_inline)) {
                        if (ABL != 2 && ABL != 6 && ABL != 7) {
                            if (CMR_WIDE_DMA_BURST) {
                                if (qd == 0)
                                    for (int j = 0; j < PPG; ++j) dma_piece(dsrc + j * NW * 1024, ddst + (unsigned)j * NW * 1024u);
                            } else if (qd % QPP == 0) {
#if CMR_WIDE_DMA_IMM
                                if constexpr (NW == 4) dma_piece_j(qd / QPP, dsrc, ddst);
                                else
#endif
                                dma_piece(dsrc + (qd / QPP) * NW * 1024, ddst + (unsigned)(qd / QPP) * NW * 1024u);
                            }
                        }
                    };
                    if constexpr (!DMA_MID) quad_dma();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if constexpr (DMA_MID) {
                            if (t == 1) {           // the boundary between the two tiles' chains of this quad
                                __builtin_amdgcn_sched_barrier(0);
                                quad_dma();
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int u = qd * 4 + j;
                            const int ks = g * GRP + u;
                            const v4u a_use = a[u % ADEPTH];
                            // register file of the resident B-operand: tile 0 in VGPRs, tile 1 in AGPRs (NT = 2); first /
                            // second half of the k-steps (NT = 1).  The accumulators are AGPRs.
                            // (WIDE_AMOVE of tile 0's operands also sit in AGPRs: the AGPR half has registers to spare)
                            const int ab = NT == 2 ? (t == 1 || ks >= KS - AMOVE ? 1 : 0) : (ks >= VK ? 1 : 0);
                            if constexpr (ABL == 1) {
                                if (ab) asm volatile("" ::"v"(a_use), "a"(qreg[t][ks < KREG ? ks : 0]));
                                else asm volatile("" ::"v"(a_use), "v"(qreg[t][ks < KREG ? ks : 0]));
                                if (ks == 0) { if constexpr (ACCV) asm volatile("" : "=v"(acc[t])); else asm volatile("" : "=a"(acc[t])); }
                            } else if (ks < KREG) {
                                if constexpr (ACCV) CmrBlk<DT>::mma_asm_cv(ab, ks == 0, acc[t], a_use, qreg[t][ks < KREG ? ks : 0]);
                                else CmrBlk<DT>::mma_asm(ab, ks == 0, acc[t], a_use, qreg[t][ks < KREG ? ks : 0]);
                            } else {
                                const v4u b = qlds[(t * KLDS + (ks < KREG ? 0 : ks - KREG)) * 64];
                                CmrBlk<DT>::mma_asm(0, ks == 0, acc[t], a_use, b);
                            }
                            if constexpr (CMR_WIDE_READ_INTERLEAVE && ABL != 6) {
                                if (t == NT - 1) {
                                    const int blk = u + ADEPTH;
                                    a[u % ADEPTH] = blk < GRP ? buf[blk * 64] : bufn[(blk - GRP) * 64];
                                }
                            }
                            if constexpr (NT == 2 && ABL != 3) {
                                // (g, qd, t, j are constants after unrolling: at most one of these survives per MFMA)
                                if (g == 0 && qd == 0 && t == 0) {                  // tile 1 of the PREVIOUS panel, under this panel's first tile-0 MFMAs
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (j == 0) {
                                        // MFMA result -> VALU reader needs 12 wait states after the producer's issue, which hipcc does
                                        // not pad for asm: the producer (tile 1's last MFMA of the previous panel) is tile 0's tail
                                        // pieces, four LDS reads, a counted wait, a barrier, a DMA statement and an MFMA back
                                        CMR_ACC_ASM("", acc[NT - 1]);
                                        emn = __builtin_inff();
                                    }
                                    if constexpr (ACCV) wide_epi_piece_v(acc[NT - 1], j, eg[j], emn);
                                    else wide_epi_piece(acc[NT - 1], j, eg[j], emn);
                                    if (j == 3) {
                                        int n1 = 0;
                                        bool comp = false;
                                        if (s > s0) epi_finish(T1{}, std::true_type{}, prow0, pnvalid, n1, comp);
                                        st_new = st_mid + __builtin_amdgcn_readfirstlane(n1);
                                        if (comp) {     // its loads already drained the ring; retire its stores too and restart the counts
                                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                            st_old = st_new = 0;
                                        }
                                        __builtin_amdgcn_sched_barrier(0);
                                    }
                                }
                                if (g == GPP - 1 && qd == GRP / 4 - 1 && t == NT - 1) {   // tile 0 of THIS panel, under tile 1's last MFMAs
                                    __builtin_amdgcn_sched_barrier(0);
                                    // The producer is tile 0's last MFMA, directly in front of tile 1's four.  Tile 1's second MFMA
                                    // depends on its first, so it issues >= 8 states after it whatever the pipe does with independent
                                    // MFMAs: the first piece goes behind the SECOND MFMA, two more states make 12 by construction.
                                    if (j == 1) {
                                        CMR_ACC_ASM("s_nop 1", acc[0]);
                                        emn = __builtin_inff();
                                    }
                                    if constexpr (ACCV) {
                                        if (j >= 1) wide_epi_piece_v(acc[0], j - 1, eg[j - 1 < 0 ? 0 : j - 1], emn);
                                        if (j == 3) wide_epi_piece_v(acc[0], 3, eg[3], emn);
                                    } else {
                                        if (j >= 1) wide_epi_piece(acc[0], j - 1, eg[j - 1 < 0 ? 0 : j - 1], emn);
                                        if (j == 3) wide_epi_piece(acc[0], 3, eg[3], emn);
                                    }
                                }
                            }
                        }
                    }
                    // the quad's four ring slots take the blocks ADEPTH ahead (they land under the next quad's MFMAs)
                    if constexpr (ABL != 6 && !CMR_WIDE_READ_INTERLEAVE) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int blk = qd * 4 + j + ADEPTH;
                            a[(qd * 4 + j) % ADEPTH] = blk < GRP ? buf[blk * 64] : bufn[(blk - GRP) * 64];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                st = stn;
                buf = bufn;
            }
            if constexpr (ABL == 3) {
                if constexpr (NT == 1) cmr_mfma_drain<NT>(acc);
                continue;
            }
            int n_stores = 0;
            bool compacted = false;
            if constexpr (NT == 1) {
                cmr_mfma_drain<NT>(acc);          // MFMA results -> VALU readers: wait states hipcc does not insert for asm
                if constexpr (NW == 8) {
                    // 256 registers per wave: the fold takes the accumulators four at a time through asm reads (wide_epi_piece)
                    // and the slow path re-reads what it needs — sixteen scores held in VGPRs from the fold to the push do not fit
                    emn = __builtin_inff();
#pragma unroll
                    for (int j = 0; j < 4; ++j) wide_epi_piece(acc[0], j, eg[j], emn);
                    epi_finish(T0{}, std::true_type{}, row0, nvalid, n_stores, compacted);
                } else {
                    epi_finish(T0{}, std::false_type{}, row0, nvalid, n_stores, compacted);
                }
                st_old = st_new;
                st_new = __builtin_amdgcn_readfirstlane(n_stores);
                if (compacted) {        // its loads already drained the ring; retire its stores too and restart the count
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    st_old = st_new = 0;
                }
            } else {
                epi_finish(T0{}, std::true_type{}, row0, nvalid, n_stores, compacted);
                st_old = st_new;
                st_mid = __builtin_amdgcn_readfirstlane(n_stores);
                if (compacted) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    st_old = st_mid = 0;
                }
                prow0 = row0;
                pnvalid = nvalid;
            }
        }
        if constexpr (NT == 2 && ABL != 3) {      // tile 1 of the last panel
            CMR_ACC_ASM("s_nop 15", acc[NT - 1]);
            emn = __builtin_inff();
#pragma unroll
            for (int j = 0; j < 4; ++j) { if constexpr (ACCV) wide_epi_piece_v(acc[NT - 1], j, eg[j], emn); else wide_epi_piece(acc[NT - 1], j, eg[j], emn); }
            int n1 = 0;
            bool comp = false;
            epi_finish(T1{}, std::true_type{}, prow0, pnvalid, n1, comp);
        }
    }

    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float mn = fminf(rmin[t], __shfl_xor(rmin[t], 32));
        const float mx = fmaxf(rmax[t], __shfl_xor(rmax[t], 32));
        if (lane < 32) {
            const int q = (wave * NT + t) * 32 + lane;
            P.mm[(size_t)blockIdx.x * NQB + q] = make_float2(mn, mx);
            P.cnt[(size_t)blockIdx.x * NQB + q] = __hip_atomic_load(&cnt_w[t * 32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

#undef CMR_ACC_ASM

// geometry of the wide kernel per shape: waves per workgroup, query tiles per wave, LDS ring depth
struct WideCfg { int nw, nt, nst, stg, klds; };
// Default waves per workgroup at 768-d (ks = 48): measured A/B on MI355X, 10 M x 768 bf16, B = 256 (profiles/r3_wide_ab.txt)
#ifndef CMR_WIDE_DEFAULT_WAVES
#define CMR_WIDE_DEFAULT_WAVES 4
#endif
static WideCfg wide_cfg(int ks, int cap, int waves) {
    if (ks == 48) {
        if (waves == 0) waves = CMR_WIDE_DEFAULT_WAVES;
        // 8 waves: 16 KiB of staging + 8 compaction stages next to the ring -> 7 stages of 16 KiB for the 256-entry lists
#ifdef CMR_WIDE8
        if (waves == 8) return {8, 1, cap > 128 ? WIDE8_NST - 1 : WIDE8_NST, WIDE8_STG, WIDE8_KLDS};
#endif
        return {4, 2, wide_nst4(48), WIDE_STG, 0};
    }
    return {4, 1, wide_nst4(ks), WIDE_STG, 0};
}

size_t cmr_wide_lds_bytes(int ks, int cap, int waves) {
    const WideCfg c = wide_cfg(ks, cap, waves);
    return (size_t)c.nst * wide_group(ks, c.nw) * 1024 + (size_t)c.nw * c.nt * 32 * 4 + (size_t)c.nw * (cap + 2) * 8 + (size_t)c.nw * c.nt * c.klds * 1024 +
           (c.nw == 8 ? 0 : (size_t)c.nw * c.stg * 16) + c.nw * 4;
}

// wide kernel availability: 16-bit dtypes at ks = 48 (768-d: 256 queries per pass — 4 waves x 2 tiles or 8 waves x 1 tile of 32),
// ks = 64 (1024-d: a tile needs 256 registers -> 4 waves x 1 tile x 32 = 128 queries per pass).  Any other padded dim, and
// fp32 indexes, run batches of more than 64 queries as ceil(B / 64) passes of the narrow kernel.
int cmr_wide_queries(int dtype, int dpad) {
    if (dtype == CMR_DT_F32) return 0;
    if (dpad == 768) return 256;
    if (dpad == 1024) return 128;
    return 0;
}

hipError_t cmr_launch_scan_wide(const CmrScanGeom& g, const CmrScanArgs& a, hipStream_t s) {
    const ScanP p = to_p(g, a);
    const WideCfg c = wide_cfg(g.ks, g.cap, g.wide_waves);
    const size_t lds = cmr_wide_lds_bytes(g.ks, g.cap, g.wide_waves);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(g.grid), dim3(c.nw * 64), lds, s, p);
        return hipGetLastError();
    };
#ifdef CMR_DEV_KNOBS
    // development builds: ablation kernels (results are wrong by design) behind the "wide_abl" option
    if (g.wide_abl && g.dtype == CMR_DT_BF16 && g.ks == 48 && g.cap == 128 && c.nw == 4) {
        if (g.wide_abl == 1) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 2, 128, wide_nst4(48), 0, 1>);
        if (g.wide_abl == 2) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 2, 128, wide_nst4(48), 0, 2>);
        if (g.wide_abl == 3) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 2, 128, wide_nst4(48), 0, 3>);
        if (g.wide_abl == 4) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 2, 128, wide_nst4(48), 0, 4>);
        if (g.wide_abl == 5) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 2, 128, wide_nst4(48), 0, 5>);
        if (g.wide_abl == 6) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 2, 128, wide_nst4(48), 0, 6>);
        if (g.wide_abl == 7) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 2, 128, wide_nst4(48), 0, 7>);
    }
#endif
#if defined(CMR_DEV_KNOBS) && defined(CMR_WIDE8)
    if (g.wide_abl && g.dtype == CMR_DT_BF16 && g.ks == 48 && g.cap == 128 && c.nw == 8) {
        if (g.wide_abl == 1) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 1, 128, WIDE8_NST, WIDE8_KLDS, 1, 8>);
        if (g.wide_abl == 2) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 1, 128, WIDE8_NST, WIDE8_KLDS, 2, 8>);
        if (g.wide_abl == 3) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 1, 128, WIDE8_NST, WIDE8_KLDS, 3, 8>);
        if (g.wide_abl == 5) return launch(scan_wide_kernel<CMR_DT_BF16, 48, 1, 128, WIDE8_NST, WIDE8_KLDS, 5, 8>);
    }
#endif
#define WCASE(DT, KSV, NTV, CAPV, NSTV, NWV) if (g.dtype == DT && g.ks == KSV && g.cap == CAPV && c.nw == NWV) return launch(scan_wide_kernel<DT, KSV, NTV, CAPV, NSTV, (NWV == 8 ? WIDE8_KLDS : 0), 0, NWV>);
    WCASE(CMR_DT_BF16, 48, 2, 128, wide_nst4(48), 4) WCASE(CMR_DT_BF16, 48, 2, 256, wide_nst4(48), 4)
    WCASE(CMR_DT_F16, 48, 2, 128, wide_nst4(48), 4) WCASE(CMR_DT_F16, 48, 2, 256, wide_nst4(48), 4)
#ifdef CMR_WIDE8     // experimental: two waves per SIMD, one tile each (hipcc does not yet fit it into 256 registers without spills)
    WCASE(CMR_DT_BF16, 48, 1, 128, WIDE8_NST, 8) WCASE(CMR_DT_BF16, 48, 1, 256, WIDE8_NST - 1, 8)
    WCASE(CMR_DT_F16, 48, 1, 128, WIDE8_NST, 8) WCASE(CMR_DT_F16, 48, 1, 256, WIDE8_NST - 1, 8)
#endif
    WCASE(CMR_DT_BF16, 64, 1, 128, wide_nst4(64), 4) WCASE(CMR_DT_BF16, 64, 1, 256, wide_nst4(64), 4)
    WCASE(CMR_DT_F16, 64, 1, 128, wide_nst4(64), 4) WCASE(CMR_DT_F16, 64, 1, 256, wide_nst4(64), 4)
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

// top-k scan with thresholds and final selection inside the launch (MODE_FIN: one query tile, <= fin_ns <= 1024 first-panel slots)
hipError_t cmr_launch_scan_fin(const CmrScanGeom& g, const CmrScanArgs& a, hipStream_t s) {
    if (g.nqt != 1 || a.qgroups > 1 || !a.fin || !a.fin_pmax || !a.fin_tau || !a.fin_dense || a.fin_wgs < 1 || a.fin_wgs * CMR_SCAN_WAVES > CMR_FIN_SLOTS ||
        a.fin_wgs > g.grid || a.fin_first < a.fin_wgs || a.fin_mul < 1 || a.fin_dcap < 1 || !a.fin_mm || g.grid > 512 || !tiny_select_stream_ok((long long)a.fin_dcap + 1025, a.k) || a.nq > CMR_FIN_MAX_QUERIES || !a.out_ids || !a.out_scores || a.sample_waves > 0)
        return hipErrorInvalidValue;
    const ScanP p = to_p(g, a);
    CmrScanGeom gf = g;
    gf.lds += CMR_FIN_LDS;
    if (gf.lds > kLdsLimit) return hipErrorInvalidValue;
    switch (g.dtype) {
        case CMR_DT_BF16: return dispatch<CMR_DT_BF16, MODE_FIN>(gf, p, s);
        case CMR_DT_F16: return dispatch<CMR_DT_F16, MODE_FIN>(gf, p, s);
        case CMR_DT_F32: return dispatch<CMR_DT_F32, MODE_FIN>(gf, p, s);
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
