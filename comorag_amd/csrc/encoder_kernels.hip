// Encoder-side gfx950 kernels of libcomorag_hip.so: the two memory-/latency-shaped pieces of a BERT layer that are not GEMMs.
//
//   attn_fwd_kernel   softmax(Q K^T / sqrt(64)) V of one (sequence, head, 128-query block) per workgroup, 16-bit in / out,
//                     fp32 scores and accumulators, keys beyond the sequence's length masked (right-padded mini-batches).
//                     Replaces the scaled-dot-product-attention call inside transformers' BertSelfAttention, i.e. part of
//                     `self.embedding_model(**inputs)` at embedding_model/BGEEmbedding.py:119.
//   add_ln_kernel     LayerNorm(y + bias + residual) * gamma + beta, one wave per token: BertSelfOutput / BertOutput minus
//                     their GEMM (dense bias add, residual add and LayerNorm are three kernels and five passes over the
//                     activations in PyTorch; here one read of y and the residual, one write).
//   embed_ln_kernel   BertEmbeddings: the word / position / token-type gathers, their sum and the LayerNorm in one pass.
//
// Attention layout.  Q, K, V are column slices of ONE packed projection output qkv[b*L, 3*hidden] (the host side multiplies by
// the concatenated query/key/value weights once): head h reads columns h*64 (Q), hidden + h*64 (K), 2*hidden + h*64 (V).
// A wave owns 32 query rows.  Scores are computed TRANSPOSED, S^T = K Q^T (A operand = a 32-key tile read from LDS, B operand
// = the wave's Q rows, loaded once from global memory in MFMA lane order), so that in the 32x32 accumulator layout
// (cmr_acc_row) lane l holds column q = l & 31: every softmax statistic of a query row is lane-local except one cross-half
// max.  The same registers, converted to 16 bits, ARE the B operand of O^T = V^T P^T: k-slot e of lane half g is key
// acc_row(8*s + e, g) — a permutation of the 16 keys of a k-step, applied identically to the A operand by reading V^T from
// LDS at [d][16*s + 4*g + {0..3}] and [d][16*s + 8 + 4*g + {0..3}].  V is transposed on its way into LDS (two keys per thread,
// eight packed 4-byte writes); K goes in row-major.  Both are double-buffered 64-key chunks: the global loads of chunk c + 1
// are in flight while chunk c is multiplied, one barrier per chunk.  35 KiB of LDS per workgroup: several workgroups share a
// CU, so one's exponentials overlap another's MFMAs (the 64-wide head makes the softmax, not the matrix pipe, the longer leg).
#include "cmr_device.h"
#include "cmr_kernels.h"

typedef __attribute__((ext_vector_type(2))) __bf16 enc_bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 enc_f16x2;

template <int DT> __device__ __forceinline__ unsigned enc_pack2(float a, float b) {   // round-to-nearest-even, a in the low half
    if constexpr (DT == CMR_DT_BF16) {
        enc_bf16x2 t;
        t[0] = (__bf16)a;
        t[1] = (__bf16)b;
        return __builtin_bit_cast(unsigned, t);
    } else {
        enc_f16x2 t;
        t[0] = (_Float16)a;
        t[1] = (_Float16)b;
        return __builtin_bit_cast(unsigned, t);
    }
}
template <int DT> __device__ __forceinline__ void enc_unpack2(unsigned u, float& a, float& b) {
    if constexpr (DT == CMR_DT_BF16) {
        a = __uint_as_float(u << 16);
        b = __uint_as_float(u & 0xFFFF0000u);
    } else {
        enc_f16x2 t = __builtin_bit_cast(enc_f16x2, u);
        a = (float)t[0];
        b = (float)t[1];
    }
}

// ------------------------------------------------------------------------------------------------ attention
#define ATT_CHUNK 64          // keys per LDS chunk
#define ATT_QBLOCK 128        // query rows per workgroup of the 4-wave kernel (4 waves x 32); the 8-wave kernel takes 256
#define ATT_KSTR 72           // K rows in LDS: 64 elements + 8 of padding (144 B: ds_read_b128 of 16 consecutive rows hit 16 different 16-B slots)
#ifndef ATT_VTR
#define ATT_VTR 1             // 1: V goes into LDS ROW-MAJOR (two 16-byte writes per thread, as K) and the PV step reads its A operands with
#endif                        //    ds_read_b64_tr_b16 (gfx950's transposing LDS read); 0: V transposed on its way in (eight 4-byte writes per thread)
#if ATT_VTR && defined(ATT_VSTR_OVERRIDE)
#define ATT_VSTR ATT_VSTR_OVERRIDE
#elif ATT_VTR
#define ATT_VSTR 88           // V rows in LDS: 64 d + 24 of padding (176 B).  Measured, us per layer at 32 x 512 x 12 heads bf16 / 16 heads fp16 (gpurun_out/r5x):
                              // 72: 46.8 / 59.7   80: 48.3 / 61.8   88: 46.2 / 59.6   96: 47.4 / 60.9   104: 48.7 / 63.7   112: 49.0 / 62.1   136: 48.5 / 62.3
#else
#define ATT_VSTR 68           // V^T rows in LDS: 64 keys + 4 (136 B: ds_read_b64 of 32 consecutive d rows hit 32 different bank pairs)
#endif
// Round 6, measured with the kernel's own duration (rocprofv3 --kernel-trace, tools/attn_variants.sh; 32 x 512 x 12 heads bf16, full and ragged
// mini-batches averaged, two runs on one box): round 5's kernel 41.6 / 40.9 us; + packed softmax arithmetic (ATT_PK) 39.5 / 40.3 — kept;
// + unpadded XOR-swizzled K / V rows written through registers (ATT_SWZL: SQ_LDS_BANK_CONFLICT 37 K -> 0, LDS-array cycles -41 %) 41.1 / 41.4;
// + the chunks by LDS-DMA instead of registers (ATT_DMA: VGPRs 120 -> 112, no ds_write) 42.0 / 41.6.  Neither the bank conflicts nor the
// register staging is what bounds this kernel: the swizzle's address arithmetic costs a VALU-bound loop more than the conflicts did.
#ifndef ATT_PK
#define ATT_PK 1              // 1: the softmax's multiply-adds and sums two scores per instruction (v_pk_fma_f32 / v_pk_add_f32)
#endif
#ifndef ATT_DMA
#define ATT_DMA 0             // 1 (needs ATT_VTR): K / V chunks go HBM / L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write),
#endif                        //    rows unpadded (128 B) and XOR-swizzled by 16-byte segment so that both read patterns stay conflict-free; 0: through registers
#ifndef ATT_SWZL
#define ATT_SWZL 0            // 1 (needs ATT_VTR): K / V rows in LDS unpadded and XOR-swizzled by 16-byte segment (ATT_SWZ below) — no bank conflict on either read;
#endif                        //    0: padded rows (ATT_KSTR / ATT_VSTR).  ATT_DMA implies it (a DMA instruction lands 1 KiB lane-linear: no padding possible)
#if ATT_DMA && !ATT_SWZL
#undef ATT_SWZL
#define ATT_SWZL 1
#endif
#if ATT_SWZL && !ATT_VTR
#error "the swizzled layout stages V row-major: it needs ATT_VTR"
#endif
typedef short att_v4s __attribute__((ext_vector_type(4)));
#define ATT_NEG (-1.0e30f)
#ifndef ATT_MIN_WG
#define ATT_MIN_WG 2          // workgroups per CU the register budget is held to (__launch_bounds__)
#endif
#ifndef ATT_ABL
#define ATT_ABL 0             // development ablations (wrong results): 1 no v_exp, 2 no QK^T MFMAs, 3 no PV MFMAs, 4 no K / V staging after chunk 0 and no barrier, 5 no barrier, 6 no V^T stores, 7 no K / V global loads
#endif
#ifndef ATT_SKIP_RESCALE
#define ATT_SKIP_RESCALE 0    // 1: a chunk that raises no query's running maximum (wave-uniform test) skips the 32 accumulator multiplies by 1.0 (measured SLOWER: 54 vs 50 us)
#endif

// NW = waves per workgroup.  4: 128 query rows share a staged K / V chunk.  8: 256 rows do — the chunk's staging (global -> registers
// -> LDS, one barrier) is the longest leg of this kernel (26 % at 4 waves: DESIGN 4.10) and costs the same per chunk however many
// query rows consume it; every thread then stages ONE 16-byte segment of K and of V instead of two.  Used for sequences of >= 256
// (padded) tokens; shorter mini-batches keep 4 waves (half of an 8-wave workgroup would hold padding rows only).
template <int DT, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : ATT_MIN_WG) void attn_fwd_kernel(const unsigned short* __restrict__ qkv, const int* __restrict__ lens, int L, int hidden,
                                                           int n_heads, int n_qblocks, int total, float sc /* log2(e) / sqrt(64) */,
                                                           unsigned short* __restrict__ out) {
#if ATT_SWZL
    // [key][64 d] unpadded (two rows per 256-byte bank row), 16-byte segment sg of row r at slot sg ^ f(r), f = ATT_SWZ: a DMA instruction lands
    // 8 rows x 128 B lane-linear (the lane picks the SOURCE segment that belongs in its slot).  f permutes the bits of u = (r >> 1) & 7 —
    // (u & 1) << 2 | u >> 1 — so that (MI355X_MICROARCH.md, LDS: 64 banks, lane groups per instruction)
    //   * ds_read_b128 of QK^T: a 16-lane group {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} reads one segment of 16 rows; its eight even rows
    //     have eight different u, its eight odd rows (the other half of the bank row) too: 16 different slots;
    //   * ds_read_b64_tr_b16 of PV: a 32-lane group reads four segments of four consecutive rows; rows r and r + 2 share a half of the bank
    //     row and differ in bit 2 of f: their segment sets {4 dt .. 4 dt + 3} ^ f are disjoint.
    // (first version: f = r & 7 — SQ_LDS_BANK_CONFLICT twice the padded layout's, rows r and r + 2 of a transposing read on the same slots)
#define ATT_SWZ(r) (((((r) >> 1) & 1) << 2) | ((((r) >> 1) & 7) >> 1))
    __shared__ __attribute__((aligned(1024))) unsigned short k_lds[2][ATT_CHUNK * 64];
    __shared__ __attribute__((aligned(1024))) unsigned short v_lds[2][ATT_CHUNK * 64];
#else
    __shared__ __attribute__((aligned(16))) unsigned short k_lds[2][ATT_CHUNK * ATT_KSTR];
    __shared__ __attribute__((aligned(16))) unsigned short v_lds[2][64 * ATT_VSTR];      // ATT_VTR: [key][d], else [d][key]
#endif
    // workgroup id -> work item.  Hardware deals workgroup ids round-robin over the 8 XCDs: the query blocks of one (sequence, head)
    // pair stay on ONE XCD, so the pair's K / V come out of that XCD's L2 after the first block; the PAIRS go round the XCDs, so every
    // XCD holds a share of every sequence.  (Round 4 kept a whole sequence — all its heads — on one XCD: in a mini-batch of mixed
    // lengths the XCD that drew the long sequences set the kernel's time — 32 sequences of 17 .. 512 tokens took as long as 32 of 512.)
    const int xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
    const int pair = (j / n_qblocks) * 8 + xcd;
    if (pair >= total / n_qblocks) return;
    // (the host sorts a mini-batch's rows by ascending length: taking the sequences from the back starts the long ones first and
    //  leaves the short ones to fill the tail)
    const int qb = j % n_qblocks, head = pair % n_heads, seq = total / (n_qblocks * n_heads) - 1 - pair / n_heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, c = lane & 31;
    constexpr int QBLOCK = NW * 32, PASSES = 8 / NW, PROWS = NW * 8;      // rows per workgroup; staging passes of PROWS K / V rows each
    const int q0 = qb * QBLOCK;
    int len = lens[seq];
    len = len < L ? len : L;
    const size_t rs = (size_t)3 * hidden;
    const unsigned short* qbase = qkv + (size_t)seq * L * rs + (size_t)head * 64;
    const unsigned short* kbase = qbase + hidden;
    const unsigned short* vbase = qbase + 2 * (size_t)hidden;
    unsigned short* obase = out + (size_t)seq * L * hidden + (size_t)head * 64;

    if (q0 >= len) {                                   // a block of padding rows only: zeros, nothing reads them but LayerNorm
        const int row = q0 + (tid >> 1);
        if (row < L) {
            uint4* dst = reinterpret_cast<uint4*>(obase + (size_t)row * hidden + (tid & 1) * 32);
            const uint4 z = make_uint4(0, 0, 0, 0);
            dst[0] = z; dst[1] = z; dst[2] = z; dst[3] = z;
        }
        return;
    }

    // the wave's 32 query rows as B operands of the four k-steps over d (lane: row c, d = 16*ks + 8*g + [0,8))
    const int qrow = q0 + wave * 32 + c;
    v4u qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = v4u{0, 0, 0, 0};
        if (qrow < L) qf[ks] = *reinterpret_cast<const v4u*>(qbase + (size_t)qrow * rs + ks * 16 + g * 8);
    }

#if ATT_DMA
    // chunk staging by LDS-DMA: wave w brings rows 8w + PROWS*h .. + 7 of K and of V (one instruction each: 64 lanes x 16 B = 8 rows), lane l the
    // segment (l & 7) ^ f(row) of row l >> 3 into slot l & 7.  Rows beyond the sequence re-read its last row: finite values under p = 0
    // (a lane switched off would leave whatever the buffer held, and 0 x Inf is NaN).  Inline asm (M0 = LDS destination): hipcc neither
    // counts these loads nor knows that the LDS reads depend on them — the wait in front of the chunk's barrier is written out below.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // (M0 takes an SGPR)
    const int dr = lane >> 3, dsg = (lane & 7) ^ ATT_SWZ(dr + 8 * (wave_u & 1));      // (a wave's rows start at a multiple of 8: bit 3 of the row = wave & 1)
    const unsigned k_dst = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned short*)&k_lds[0][0]) + (unsigned)wave_u * 1024u;
    const unsigned v_dst = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned short*)&v_lds[0][0]) + (unsigned)wave_u * 1024u;
    auto dma_chunk = [&](int ch, int buf) {
        const int k0 = ch * ATT_CHUNK;
#pragma unroll
        for (int h = 0; h < PASSES; ++h) {
            int key = k0 + wave * 8 + PROWS * h + dr;
            key = key < len ? key : len - 1;
            const unsigned voff = ((unsigned)key * (unsigned)rs + (unsigned)dsg * 8u) * 2u;
            const unsigned dk = k_dst + (unsigned)buf * (ATT_CHUNK * 128u) + (unsigned)h * (PROWS * 128u);
            const unsigned dv = v_dst + (unsigned)buf * (ATT_CHUNK * 128u) + (unsigned)h * (PROWS * 128u);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(kbase), "s"(dk) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(vbase), "s"(dv) : "memory", "m0");
        }
    };
#endif
    // chunk staging.  K: thread -> rows r and r + 32, 16-byte segment seg.  V: thread -> the key pair pi, d segment dseg.
    const int kr = tid >> 3, kseg = tid & 7;
    const int pi = wave * 8 + (lane & 7), dseg = lane >> 3;
    v4u kreg[PASSES], vreg[PASSES];
    auto load_chunk = [&](int ch) {
        const int k0 = ch * ATT_CHUNK;
#pragma unroll
        for (int h = 0; h < PASSES; ++h) {
            const int key = k0 + kr + PROWS * h;
            kreg[h] = v4u{0, 0, 0, 0};
            if (key < len) kreg[h] = *reinterpret_cast<const v4u*>(kbase + (size_t)key * rs + kseg * 8);
            const int vkey = ATT_VTR ? key : k0 + 2 * pi + h;                  // ATT_VTR: the same (row, segment) as K
            vreg[h] = v4u{0, 0, 0, 0};
            if (vkey < len) vreg[h] = *reinterpret_cast<const v4u*>(vbase + (size_t)vkey * rs + (ATT_VTR ? kseg : dseg) * 8);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
#if ATT_SWZL
        for (int h = 0; h < PASSES; ++h) *reinterpret_cast<v4u*>(&k_lds[buf][(kr + PROWS * h) * 64 + ((kseg ^ ATT_SWZ(kr + PROWS * h)) * 8)]) = kreg[h];
#pragma unroll
        for (int h = 0; h < (ATT_ABL == 6 ? 0 : PASSES); ++h) *reinterpret_cast<v4u*>(&v_lds[buf][(kr + PROWS * h) * 64 + ((kseg ^ ATT_SWZ(kr + PROWS * h)) * 8)]) = vreg[h];
#elif ATT_VTR
        for (int h = 0; h < PASSES; ++h) *reinterpret_cast<v4u*>(&k_lds[buf][(kr + PROWS * h) * ATT_KSTR + kseg * 8]) = kreg[h];
#pragma unroll
        for (int h = 0; h < (ATT_ABL == 6 ? 0 : PASSES); ++h) *reinterpret_cast<v4u*>(&v_lds[buf][(kr + PROWS * h) * ATT_VSTR + kseg * 8]) = vreg[h];
#else
        for (int h = 0; h < PASSES; ++h) *reinterpret_cast<v4u*>(&k_lds[buf][(kr + PROWS * h) * ATT_KSTR + kseg * 8]) = kreg[h];
#pragma unroll
        for (int w = 0; w < (ATT_ABL == 6 ? 0 : 4); ++w) {                  // element 2w and 2w + 1 of both keys -> V^T[d][key0], V^T[d][key0 + 1]
            const unsigned a = vreg[0][w], b = vreg[1][w];
            *reinterpret_cast<unsigned*>(&v_lds[buf][(dseg * 8 + 2 * w) * ATT_VSTR + 2 * pi]) = (a & 0xFFFFu) | (b << 16);
            *reinterpret_cast<unsigned*>(&v_lds[buf][(dseg * 8 + 2 * w + 1) * ATT_VSTR + 2 * pi]) = (a >> 16) | (b & 0xFFFF0000u);
        }
#endif
    };

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.0f; o[1][r] = 0.0f; }
    float m = ATT_NEG, lsum = 0.0f;
    const int nch = (len + ATT_CHUNK - 1) / ATT_CHUNK;
#if ATT_DMA
    dma_chunk(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    load_chunk(0);
    store_chunk(0);
#endif
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int cur = ch & 1;
#if ATT_DMA
        if (ATT_ABL != 4 && ATT_ABL != 7 && ch + 1 < nch) dma_chunk(ch + 1, cur ^ 1);      // (everybody left that buffer at the previous chunk's barrier)
#else
        if (ATT_ABL != 4 && ATT_ABL != 7 && ch + 1 < nch) load_chunk(ch + 1);
#endif
        // S^T tiles: keys t*32 + [0,32) of the chunk x the wave's 32 queries
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#if ATT_SWZL
                const v4u kf = *reinterpret_cast<const v4u*>(&k_lds[ATT_ABL == 4 ? 0 : cur][(t * 32 + c) * 64 + (((ks * 2 + g) ^ ATT_SWZ(c)) * 8)]);
#else
                const v4u kf = *reinterpret_cast<const v4u*>(&k_lds[ATT_ABL == 4 ? 0 : cur][(t * 32 + c) * ATT_KSTR + ks * 16 + g * 8]);
#endif
                if (ATT_ABL != 2) s[t] = CmrBlk<DT>::mma(kf, qf[ks], s[t]);
                else s[t][ks] += __uint_as_float(kf[0] & 0x3F800000u);
            }
        }
        const int k0 = ch * ATT_CHUNK;
        if (k0 + ATT_CHUNK > len) {                    // the chunk holding the end of the sequence (wave-uniform)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + t * 32 + cmr_acc_row(r, lane) >= len) s[t][r] = ATT_NEG;
        }
        float cm = ATT_NEG;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) cm = fmaxf(cm, s[t][r]);
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * sc);
        const float msc = mn * sc;
        // two scores per instruction where the ISA has a packed form (v_pk_fma_f32, v_pk_add_f32: the kernel is bound by its VALU work —
        // 155 instructions per 16 MFMAs and chunk, 64 of them these multiply-adds and sums)
#if ATT_PK
        typedef float att_f2 __attribute__((ext_vector_type(2)));
        att_f2 ps2 = {0.0f, 0.0f};
        const att_f2 sc2 = {sc, sc}, nm2 = {-msc, -msc};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                att_f2 x = {s[t][r], s[t][r + 1]};
                x = __builtin_elementwise_fma(x, sc2, nm2);
                if (ATT_ABL != 1) { x[0] = __builtin_amdgcn_exp2f(x[0]); x[1] = __builtin_amdgcn_exp2f(x[1]); }
                s[t][r] = x[0];
                s[t][r + 1] = x[1];
                ps2 += x;
            }
        const float ps = ps2[0] + ps2[1];
#else
        float ps = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = ATT_ABL == 1 ? fmaf(s[t][r], sc, -msc) : __builtin_amdgcn_exp2f(fmaf(s[t][r], sc, -msc));
                s[t][r] = p;
                ps += p;
            }
#endif
        // (after the first chunks the running maxima rarely move: alpha is exactly 1.0 in every lane then, and x * 1.0 == x)
        if (!ATT_SKIP_RESCALE || __any(mn != m)) {
            lsum = fmaf(lsum, alpha, ps);
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        } else {
            lsum += ps;
        }
        // O^T += V^T P^T: four k-steps of 16 keys; the P registers 8*sp .. 8*sp + 7 of tile t are the B operand as they are
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                v4u pb;
                pb[0] = enc_pack2<DT>(s[t][8 * sp + 0], s[t][8 * sp + 1]);
                pb[1] = enc_pack2<DT>(s[t][8 * sp + 2], s[t][8 * sp + 3]);
                pb[2] = enc_pack2<DT>(s[t][8 * sp + 4], s[t][8 * sp + 5]);
                pb[3] = enc_pack2<DT>(s[t][8 * sp + 6], s[t][8 * sp + 7]);
                const int kb = t * 32 + sp * 16 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
#if ATT_VTR
                    // the A operand of lane l: V[kb + {0..3, 8..11}][dt*32 + c].  A transposing read hands lane i of a 16-lane group column i of the
                    // 4 x 16 block whose row (i >> 2), columns 4 * (i & 3) .. + 3 that lane addresses (tools/probe/tr_probe.hip): rows = keys
                    // kb .. kb + 3, columns = d of this group's sixteen lanes
#if ATT_SWZL
                    // (row + 8 flips bit 2 of u = bit 1 of f: the second read's slot is the first one's ^ 2)
                    const int vrow_ = kb + ((lane & 15) >> 2);
                    const int vslot_ = (dt * 4 + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) ^ ATT_SWZ(vrow_);
                    const unsigned short* vblk = &v_lds[cur][vrow_ * 64 + vslot_ * 8 + 4 * (lane & 1)];
                    const unsigned short* vblk8 = &v_lds[cur][(vrow_ + 8) * 64 + (vslot_ ^ 2) * 8 + 4 * (lane & 1)];
#else
                    const unsigned short* vblk = &v_lds[cur][(kb + ((lane & 15) >> 2)) * ATT_VSTR + dt * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)];
                    const unsigned short* vblk8 = vblk + 8 * ATT_VSTR;
#endif
                    typedef __attribute__((address_space(3))) att_v4s* lds_v4s;
                    const att_v4s lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)vblk);
                    const att_v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)vblk8);
                    const uint2 lo = __builtin_bit_cast(uint2, lo4), hi = __builtin_bit_cast(uint2, hi4);
#else
                    const unsigned short* vrow = &v_lds[cur][(dt * 32 + c) * ATT_VSTR + kb];
                    const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
                    const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 8);
#endif
                    if (ATT_ABL != 3) o[dt] = CmrBlk<DT>::mma(v4u{lo.x, lo.y, hi.x, hi.y}, pb, o[dt]);
                    else o[dt][0] += __uint_as_float((lo.x ^ pb[0]) & 0x3F800000u);
                }
            }
        if (ATT_ABL != 4) {
#if ATT_DMA
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of the next chunk have landed; the barrier says everybody's have
#else
            if (ch + 1 < nch) store_chunk(cur ^ 1);
#endif
            if (ATT_ABL != 5) __syncthreads();
        }
    }
    const float inv = 1.0f / (lsum + __shfl_xor(lsum, 32));
    if (qrow < L) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {           // registers 4*rr .. 4*rr + 3 = d columns dt*32 + 8*rr + 4*g + [0,4)
                uint2 w;
                w.x = enc_pack2<DT>(o[dt][4 * rr + 0] * inv, o[dt][4 * rr + 1] * inv);
                w.y = enc_pack2<DT>(o[dt][4 * rr + 2] * inv, o[dt][4 * rr + 3] * inv);
                *reinterpret_cast<uint2*>(obase + (size_t)qrow * hidden + dt * 32 + 8 * rr + 4 * g) = w;
            }
    }
}

#ifndef ATT_WAVES_MODE
#define ATT_WAVES_MODE 0      // 0: eight waves per workgroup for mini-batches of >= 256 (padded) tokens, four below; 4 / 8: always
#endif
template <int DT, int NW>
static hipError_t launch_attention(const void* qkv, const int* lens, int b, int L, int n_heads, void* out, hipStream_t s) {
    static_assert(ATT_VTR || NW == 4, "the transposed-store layout of V is a 4-wave mapping");
    const int hidden = n_heads * 64;
    const int n_qblocks = (L + NW * 32 - 1) / (NW * 32);
    const long long total = (long long)n_qblocks * n_heads * b;
    if (total > (1LL << 28)) return hipErrorInvalidValue;
    const long long pairs = (long long)n_heads * b;
    const int per = (int)((pairs + 7) / 8) * n_qblocks;        // workgroups per XCD: its share of the (sequence, head) pairs x their query blocks
    const float sc = 1.4426950408889634f * 0.125f;
    hipLaunchKernelGGL((attn_fwd_kernel<DT, NW>), dim3((unsigned)(per * 8)), dim3(NW * 64), 0, s, reinterpret_cast<const unsigned short*>(qkv), lens, L, hidden,
                       n_heads, n_qblocks, (int)total, sc, reinterpret_cast<unsigned short*>(out));
    return hipGetLastError();
}

hipError_t cmr_launch_attention(const void* qkv, int dtype, const int* lens, int b, int L, int n_heads, void* out, hipStream_t s) {
    const bool eight = ATT_VTR && (ATT_WAVES_MODE == 8 || (ATT_WAVES_MODE == 0 && L >= 256));
    if (dtype == CMR_DT_BF16) return eight ? launch_attention<CMR_DT_BF16, (ATT_VTR ? 8 : 4)>(qkv, lens, b, L, n_heads, out, s) : launch_attention<CMR_DT_BF16, 4>(qkv, lens, b, L, n_heads, out, s);
    return eight ? launch_attention<CMR_DT_F16, (ATT_VTR ? 8 : 4)>(qkv, lens, b, L, n_heads, out, s) : launch_attention<CMR_DT_F16, 4>(qkv, lens, b, L, n_heads, out, s);
}

// ------------------------------------------------------------------------------------------------ bias + residual + LayerNorm
// One wave per token row, d = 4 * d4 elements, lane holds vectors lane, lane + 64, ... (8-byte loads: a wave instruction reads
// 512 contiguous bytes).  Sum, mean and variance in fp32 over the UNROUNDED sum y + bias + residual (PyTorch rounds to 16 bits
// after the bias and again after the residual add).
// v[VPL][4] = the lane's elements of one row (vector i = j*64 + lane, valid while i < d4): mean / variance over the wave, then
// (v - mean) * rstd * gamma + beta, rounded once, written as 8-byte vectors
template <int DT, int VPL>
__device__ __forceinline__ void enc_ln_store(float (&v)[VPL][4], float sum, int lane, int d4, const uint2* __restrict__ gamma,
                                             const uint2* __restrict__ beta, float eps, uint2* __restrict__ out_row) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv_d = 1.0f / (float)(4 * d4);
    const float mean = sum * inv_d;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < VPL; ++j)
        if (j * 64 + lane < d4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = v[j][e] - mean; sq = fmaf(t, t, sq); }
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq * inv_d + eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int i = j * 64 + lane;
        if (i < d4) {
            const uint2 gg = gamma[i], be = beta[i];
            float gm[4], bt[4];
            enc_unpack2<DT>(gg.x, gm[0], gm[1]);
            enc_unpack2<DT>(gg.y, gm[2], gm[3]);
            enc_unpack2<DT>(be.x, bt[0], bt[1]);
            enc_unpack2<DT>(be.y, bt[2], bt[3]);
            uint2 w;
            w.x = enc_pack2<DT>(fmaf((v[j][0] - mean) * rstd, gm[0], bt[0]), fmaf((v[j][1] - mean) * rstd, gm[1], bt[1]));
            w.y = enc_pack2<DT>(fmaf((v[j][2] - mean) * rstd, gm[2], bt[2]), fmaf((v[j][3] - mean) * rstd, gm[3], bt[3]));
            out_row[i] = w;
        }
    }
}
template <int DT> __device__ __forceinline__ void enc_acc4(float (&f)[4], uint2 a) {
    float t[4];
    enc_unpack2<DT>(a.x, t[0], t[1]);
    enc_unpack2<DT>(a.y, t[2], t[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] += t[e];
}

template <int DT, int VPL>
__global__ __launch_bounds__(256) void add_ln_kernel(const uint2* __restrict__ y, const uint2* __restrict__ bias, const uint2* __restrict__ res,
                                                     const uint2* __restrict__ gamma, const uint2* __restrict__ beta, float eps, long long rows,
                                                     int d4, uint2* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    float v[VPL][4];
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int i = j * 64 + lane;
        v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.0f;
        if (i < d4) {
            enc_acc4<DT>(v[j], y[row * d4 + i]);
            if (bias) enc_acc4<DT>(v[j], bias[i]);
            if (res) enc_acc4<DT>(v[j], res[row * d4 + i]);
            sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    enc_ln_store<DT, VPL>(v, sum, lane, d4, gamma, beta, eps, out + row * d4);
}

// The same for d % 8 == 0 with 16-byte vectors: sixteen lanes per token row, four rows per wave (a wave instruction still reads
// contiguous 256-byte pieces, and each lane has 2 * VPL 16-byte loads in flight instead of 8-byte ones: the one-row-per-wave
// form above ran at 4.5 TB/s of algorithmic traffic at 768 columns).
template <int DT> __device__ __forceinline__ void enc_acc8(float (&f)[8], uint4 a) {
    float t[8];
    enc_unpack2<DT>(a.x, t[0], t[1]);
    enc_unpack2<DT>(a.y, t[2], t[3]);
    enc_unpack2<DT>(a.z, t[4], t[5]);
    enc_unpack2<DT>(a.w, t[6], t[7]);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += t[e];
}
template <int DT, int VPL>
__global__ __launch_bounds__(256) void add_ln16_kernel(const uint4* __restrict__ y, const uint4* __restrict__ bias, const uint4* __restrict__ res,
                                                       const uint4* __restrict__ gamma, const uint4* __restrict__ beta, float eps, long long rows,
                                                       int d8, uint4* __restrict__ out) {
    const int sub = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;                           // whole 16-lane groups leave together: the shuffles below stay inside a group
    float v[VPL][8];
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int i = j * 16 + sub;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = 0.0f;
        if (i < d8) {
            enc_acc8<DT>(v[j], y[row * d8 + i]);
            if (bias) enc_acc8<DT>(v[j], bias[i]);
            if (res) enc_acc8<DT>(v[j], res[row * d8 + i]);
            sum += ((v[j][0] + v[j][1]) + (v[j][2] + v[j][3])) + ((v[j][4] + v[j][5]) + (v[j][6] + v[j][7]));
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv_d = 1.0f / (float)(8 * d8);
    const float mean = sum * inv_d;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < VPL; ++j)
        if (j * 16 + sub < d8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float t = v[j][e] - mean; sq = fmaf(t, t, sq); }
        }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq * inv_d + eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int i = j * 16 + sub;
        if (i < d8) {
            float gm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            enc_acc8<DT>(gm, gamma[i]);
            enc_acc8<DT>(bt, beta[i]);
            uint4 w;
            w.x = enc_pack2<DT>(fmaf((v[j][0] - mean) * rstd, gm[0], bt[0]), fmaf((v[j][1] - mean) * rstd, gm[1], bt[1]));
            w.y = enc_pack2<DT>(fmaf((v[j][2] - mean) * rstd, gm[2], bt[2]), fmaf((v[j][3] - mean) * rstd, gm[3], bt[3]));
            w.z = enc_pack2<DT>(fmaf((v[j][4] - mean) * rstd, gm[4], bt[4]), fmaf((v[j][5] - mean) * rstd, gm[5], bt[5]));
            w.w = enc_pack2<DT>(fmaf((v[j][6] - mean) * rstd, gm[6], bt[6]), fmaf((v[j][7] - mean) * rstd, gm[7], bt[7]));
            out[row * d8 + i] = w;
        }
    }
}

// The LAST layer's bias + residual + LayerNorm with the encoder tail folded in (embedding_model/BGEEmbedding.py:15-28 mean_pooling,
// :126-127 F.normalize): the layer's output [b, l, d] is never written.  A block takes 16 consecutive tokens of ONE sequence (l is a
// multiple of 16), normalises them as add_ln16_kernel does, rounds to 16 bits (what the hidden state would have held), zeroes the
// tokens at or behind the sequence's length and adds the 16 tokens up in a fixed order (xor tree inside a wave, waves 0..3 in
// LDS): one fp32 partial row [d] per block, 1/16 of the bytes of the hidden state at 4 bytes instead of 2 -> an eighth of the
// write, and nothing is read back but these partials.  Blocks that hold only padding leave at once (no loads, no partial).
// pool_finish_kernel then adds a sequence's partials in block order, divides by the token count and L2-normalises (eps 1e-12).
template <int DT, int VPL>
__global__ __launch_bounds__(256) void add_ln16_pool_kernel(const uint4* __restrict__ y, const uint4* __restrict__ bias, const uint4* __restrict__ res,
                                                            const uint4* __restrict__ gamma, const uint4* __restrict__ beta, float eps, int l, int d8,
                                                            const int* __restrict__ lens, float* __restrict__ partial) {
    __shared__ float red[4][16 * VPL * 8];
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const int seq = blockIdx.y, t0 = blockIdx.x * 16;
    const int len = lens[seq];
    if (t0 >= len) return;                                          // a block of padding only
    const long long row = (long long)seq * l + t0 + grp;
    float v[VPL][8];
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int i = j * 16 + sub;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = 0.0f;
        if (i < d8) {
            enc_acc8<DT>(v[j], y[row * d8 + i]);
            if (bias) enc_acc8<DT>(v[j], bias[i]);
            if (res) enc_acc8<DT>(v[j], res[row * d8 + i]);
            sum += ((v[j][0] + v[j][1]) + (v[j][2] + v[j][3])) + ((v[j][4] + v[j][5]) + (v[j][6] + v[j][7]));
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv_d = 1.0f / (float)(8 * d8);
    const float mean = sum * inv_d;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < VPL; ++j)
        if (j * 16 + sub < d8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float t = v[j][e] - mean; sq = fmaf(t, t, sq); }
        }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq * inv_d + eps);
    const bool live = t0 + grp < len;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int i = j * 16 + sub;
        if (i < d8) {
            float gm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            enc_acc8<DT>(gm, gamma[i]);
            enc_acc8<DT>(bt, beta[i]);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {      // round to 16 bits as the stored hidden state would be, then back
                const unsigned w = enc_pack2<DT>(fmaf((v[j][e] - mean) * rstd, gm[e], bt[e]), fmaf((v[j][e + 1] - mean) * rstd, gm[e + 1], bt[e + 1]));
                enc_unpack2<DT>(w, v[j][e], v[j][e + 1]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = live ? v[j][e] : 0.0f;
            x += __shfl_xor(x, 16);               // the wave's four tokens: a fixed tree
            x += __shfl_xor(x, 32);
            v[j][e] = x;
        }
    }
    if ((threadIdx.x & 63) < 16) {
#pragma unroll
        for (int j = 0; j < VPL; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][(j * 16 + sub) * 8 + e] = v[j][e];
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float* dst = partial + ((size_t)seq * gridDim.x + blockIdx.x) * (size_t)(8 * d8);
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = j * 16 + sub;
            if (i < d8) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = ((red[0][i * 8 + e] + red[1][i * 8 + e]) + red[2][i * 8 + e]) + red[3][i * 8 + e];
                reinterpret_cast<float4*>(dst + i * 8)[0] = make_float4(o[0], o[1], o[2], o[3]);
                reinterpret_cast<float4*>(dst + i * 8)[1] = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
    }
}

// out[seq] = normalise(sum over the sequence's live blocks of partial / len): one block per sequence
__global__ __launch_bounds__(256) void pool_finish_kernel(const float* __restrict__ partial, const int* __restrict__ lens, int nblk, int d, int normalize,
                                                          float* __restrict__ out) {
    __shared__ float wsum[4];
    const int seq = blockIdx.x, len = lens[seq];
    const int live = len > 0 ? (len + 15) / 16 : 0;
    const float inv = len > 0 ? 1.0f / (float)len : 0.0f;
    float ss = 0.0f;
    float keep[8];                                                  // d <= 2048 = 256 threads x 8
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int f = r * 256 + threadIdx.x;
        float s = 0.0f;
        if (f < d)
            for (int b = 0; b < live && b < nblk; ++b) s += partial[((size_t)seq * nblk + b) * d + f];
        keep[r] = s * inv;
        ss = fmaf(keep[r], keep[r], ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float norm = sqrtf((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    const float scale = normalize ? 1.0f / fmaxf(norm, 1e-12f) : 1.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int f = r * 256 + threadIdx.x;
        if (f < d) out[(size_t)seq * d + f] = keep[r] * scale;
    }
}

// BertEmbeddings: LayerNorm(word[ids[t]] + position[t mod L] + token_type[tt[t]]), one wave per token (tt == NULL: type 0).
// Ids outside the tables are clamped into them (PyTorch's gather would fault the device instead).
template <int DT, int VPL>
__global__ __launch_bounds__(256) void embed_ln_kernel(const long long* __restrict__ ids, const long long* __restrict__ tt,
                                                       const uint2* __restrict__ word, const uint2* __restrict__ pos, const uint2* __restrict__ type,
                                                       const uint2* __restrict__ gamma, const uint2* __restrict__ beta, float eps, long long rows,
                                                       int L, int d4, int vocab, int n_pos, int n_types, int pos_off, uint2* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    long long id = ids[row], ty = tt ? tt[row] : 0;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    ty = ty < 0 ? 0 : (ty >= n_types ? n_types - 1 : ty);
    int p = (int)(row % L) + pos_off;      // RoBERTa-style tables start at padding_idx + 1 (right-padded rows: token t is position t + offset)
    p = p >= n_pos ? n_pos - 1 : p;
    float v[VPL][4];
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int i = j * 64 + lane;
        v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.0f;
        if (i < d4) {
            enc_acc4<DT>(v[j], word[id * d4 + i]);
            enc_acc4<DT>(v[j], pos[(long long)p * d4 + i]);
            enc_acc4<DT>(v[j], type[ty * d4 + i]);
            sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    enc_ln_store<DT, VPL>(v, sum, lane, d4, gamma, beta, eps, out + row * d4);
}

// The same from RAGGED token ids: ids32 holds the sequences' tokens back to back (int32), sequence s = ids32[off[s] .. off[s + 1]);
// row (s, t) of the [b, L] mini-batch embeds token t of sequence s, or token 0 behind the sequence's end (those rows are padding:
// attention masks them as keys, the pooling leaves them out).  No padded id / mask / token-type tensors exist on either side of
// the link: the host concatenates the tokenizer's output and ships lens | offsets | ids in ONE copy.  Token type 0 (one segment).
template <int DT, int VPL>
__global__ __launch_bounds__(256) void embed_ln_ragged_kernel(const int* __restrict__ ids32, const int* __restrict__ off, const uint2* __restrict__ word,
                                                              const uint2* __restrict__ pos, const uint2* __restrict__ type, const uint2* __restrict__ gamma,
                                                              const uint2* __restrict__ beta, float eps, long long rows, int L, int d4, int vocab, int n_pos,
                                                              int pos_off, uint2* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int seq = (int)(row / L), t = (int)(row % L);
    const int o0 = off[seq], len = off[seq + 1] - o0;
    long long id = t < len ? ids32[o0 + t] : 0;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int p = t + pos_off;
    p = p >= n_pos ? n_pos - 1 : p;
    float v[VPL][4];
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int i = j * 64 + lane;
        v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.0f;
        if (i < d4) {
            enc_acc4<DT>(v[j], word[id * d4 + i]);
            enc_acc4<DT>(v[j], pos[(long long)p * d4 + i]);
            enc_acc4<DT>(v[j], type[i]);
            sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    enc_ln_store<DT, VPL>(v, sum, lane, d4, gamma, beta, eps, out + row * d4);
}

template <int DT>
static hipError_t launch_embed_ln_ragged(const int* ids32, const int* off, const void* word, const void* pos, const void* type, const void* gamma,
                                         const void* beta, float eps, long long rows, int L, int d, int vocab, int n_pos, int pos_off, void* out, hipStream_t s) {
    const int d4 = d / 4, vpl = (d4 + 63) / 64;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define ENC_EMBR(V)                                                                                                                       \
    hipLaunchKernelGGL((embed_ln_ragged_kernel<DT, V>), grid, block, 0, s, ids32, off, reinterpret_cast<const uint2*>(word), reinterpret_cast<const uint2*>(pos), \
                       reinterpret_cast<const uint2*>(type), reinterpret_cast<const uint2*>(gamma), reinterpret_cast<const uint2*>(beta), eps, rows, \
                       L, d4, vocab, n_pos, pos_off, reinterpret_cast<uint2*>(out))
    switch (vpl) {
        case 1: ENC_EMBR(1); break;
        case 2: ENC_EMBR(2); break;
        case 3: ENC_EMBR(3); break;
        case 4: ENC_EMBR(4); break;
        case 5: case 6: ENC_EMBR(6); break;
        case 7: case 8: ENC_EMBR(8); break;
        default: return hipErrorInvalidValue;
    }
#undef ENC_EMBR
    return hipGetLastError();
}

hipError_t cmr_launch_embed_layernorm_ragged(const int* ids32, const int* off, const void* word, const void* pos, const void* type, const void* gamma,
                                             const void* beta, float eps, long long rows, int L, int d, int vocab, int n_pos, int pos_off, int dtype,
                                             void* out, hipStream_t s) {
    if (dtype == CMR_DT_BF16) return launch_embed_ln_ragged<CMR_DT_BF16>(ids32, off, word, pos, type, gamma, beta, eps, rows, L, d, vocab, n_pos, pos_off, out, s);
    return launch_embed_ln_ragged<CMR_DT_F16>(ids32, off, word, pos, type, gamma, beta, eps, rows, L, d, vocab, n_pos, pos_off, out, s);
}

#ifndef ENC_LN_FEW_ROWS
#define ENC_LN_FEW_ROWS 256
#endif
template <int DT>
static hipError_t launch_add_ln(const void* y, const void* bias, const void* res, const void* gamma, const void* beta, float eps, long long rows,
                                int d, void* out, hipStream_t s) {
    const bool al16 = (((uintptr_t)y | (uintptr_t)bias | (uintptr_t)res | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15) == 0;
    // A handful of rows (one short query: 16 .. 128 token rows, 24 / 48 of these launches per encode) is all latency: a WAVE per row (8-byte
    // vectors, three or four per lane and array at 768 / 1024-d) instead of sixteen lanes per row (six / eight 16-byte vectors per lane) —
    // a quarter of the dependent loads per lane, four times the waves.  Above, the 16-byte kernel's bandwidth wins (14.6 us at 16 K rows).
    const bool few = rows <= ENC_LN_FEW_ROWS && d % 4 == 0 && (d / 4 + 63) / 64 <= 8;
    if (d % 8 == 0 && al16 && !few) {
        const int d8 = d / 8, vpl16 = (d8 + 15) / 16;
        const dim3 grid16((unsigned)((rows + 15) / 16)), block16(256);
#define ENC_LN16(V)                                                                                                                       \
    hipLaunchKernelGGL((add_ln16_kernel<DT, V>), grid16, block16, 0, s, reinterpret_cast<const uint4*>(y), reinterpret_cast<const uint4*>(bias), \
                       reinterpret_cast<const uint4*>(res), reinterpret_cast<const uint4*>(gamma), reinterpret_cast<const uint4*>(beta), eps, \
                       rows, d8, reinterpret_cast<uint4*>(out))
        switch (vpl16) {
            case 1: ENC_LN16(1); break;
            case 2: ENC_LN16(2); break;
            case 3: case 4: ENC_LN16(4); break;
            case 5: case 6: ENC_LN16(6); break;
            case 7: case 8: ENC_LN16(8); break;
            case 9: case 10: case 11: case 12: ENC_LN16(12); break;
            case 13: case 14: case 15: case 16: ENC_LN16(16); break;
            default: return hipErrorInvalidValue;
        }
#undef ENC_LN16
        return hipGetLastError();
    }
    const int d4 = d / 4, vpl = (d4 + 63) / 64;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define ENC_LN(V)                                                                                                                     \
    hipLaunchKernelGGL((add_ln_kernel<DT, V>), grid, block, 0, s, reinterpret_cast<const uint2*>(y), reinterpret_cast<const uint2*>(bias), \
                       reinterpret_cast<const uint2*>(res), reinterpret_cast<const uint2*>(gamma), reinterpret_cast<const uint2*>(beta), eps, \
                       rows, d4, reinterpret_cast<uint2*>(out))
    switch (vpl) {
        case 1: ENC_LN(1); break;
        case 2: ENC_LN(2); break;
        case 3: ENC_LN(3); break;
        case 4: ENC_LN(4); break;
        case 5: case 6: ENC_LN(6); break;
        case 7: case 8: ENC_LN(8); break;
        default: return hipErrorInvalidValue;
    }
#undef ENC_LN
    return hipGetLastError();
}

template <int DT>
static hipError_t launch_embed_ln(const long long* ids, const long long* tt, const void* word, const void* pos, const void* type, const void* gamma,
                                  const void* beta, float eps, long long rows, int L, int d, int vocab, int n_pos, int n_types, int pos_off, void* out,
                                  hipStream_t s) {
    const int d4 = d / 4, vpl = (d4 + 63) / 64;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define ENC_EMB(V)                                                                                                                       \
    hipLaunchKernelGGL((embed_ln_kernel<DT, V>), grid, block, 0, s, ids, tt, reinterpret_cast<const uint2*>(word), reinterpret_cast<const uint2*>(pos), \
                       reinterpret_cast<const uint2*>(type), reinterpret_cast<const uint2*>(gamma), reinterpret_cast<const uint2*>(beta), eps, rows, \
                       L, d4, vocab, n_pos, n_types, pos_off, reinterpret_cast<uint2*>(out))
    switch (vpl) {
        case 1: ENC_EMB(1); break;
        case 2: ENC_EMB(2); break;
        case 3: ENC_EMB(3); break;
        case 4: ENC_EMB(4); break;
        case 5: case 6: ENC_EMB(6); break;
        case 7: case 8: ENC_EMB(8); break;
        default: return hipErrorInvalidValue;
    }
#undef ENC_EMB
    return hipGetLastError();
}

template <int DT>
static hipError_t launch_add_ln_pool(const void* y, const void* bias, const void* res, const void* gamma, const void* beta, float eps, int b, int l,
                                     int d, const int* lens, int normalize, float* partial, float* out, hipStream_t s) {
    const int d8 = d / 8, vpl16 = (d8 + 15) / 16;
    const dim3 grid((unsigned)(l / 16), (unsigned)b), block(256);
#define ENC_LNP(V)                                                                                                                        \
    hipLaunchKernelGGL((add_ln16_pool_kernel<DT, V>), grid, block, 0, s, reinterpret_cast<const uint4*>(y), reinterpret_cast<const uint4*>(bias), \
                       reinterpret_cast<const uint4*>(res), reinterpret_cast<const uint4*>(gamma), reinterpret_cast<const uint4*>(beta), eps, l, \
                       d8, lens, partial)
    switch (vpl16) {
        case 1: ENC_LNP(1); break;
        case 2: ENC_LNP(2); break;
        case 3: case 4: ENC_LNP(4); break;
        case 5: case 6: ENC_LNP(6); break;
        case 7: case 8: ENC_LNP(8); break;
        case 9: case 10: case 11: case 12: ENC_LNP(12); break;
        case 13: case 14: case 15: case 16: ENC_LNP(16); break;
        default: return hipErrorInvalidValue;
    }
#undef ENC_LNP
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pool_finish_kernel, dim3((unsigned)b), dim3(256), 0, s, partial, lens, l / 16, d, normalize, out);
    return hipGetLastError();
}

// LayerNorm(y + bias + residual) of a [b, l, d] mini-batch folded into the masked mean-pool + L2-normalise of its rows:
// l % 16 == 0, d % 8 == 0, d <= 2048, 16-byte aligned buffers; partial: b * (l / 16) * d floats of scratch
hipError_t cmr_launch_add_layernorm_pool(const void* y, const void* bias, const void* res, const void* gamma, const void* beta, float eps, int b, int l,
                                         int d, int dtype, const int* lens, int normalize, float* partial, float* out, hipStream_t s) {
    if (dtype == CMR_DT_BF16) return launch_add_ln_pool<CMR_DT_BF16>(y, bias, res, gamma, beta, eps, b, l, d, lens, normalize, partial, out, s);
    return launch_add_ln_pool<CMR_DT_F16>(y, bias, res, gamma, beta, eps, b, l, d, lens, normalize, partial, out, s);
}

hipError_t cmr_launch_embed_layernorm(const long long* ids, const long long* tt, const void* word, const void* pos, const void* type,
                                      const void* gamma, const void* beta, float eps, long long rows, int L, int d, int vocab, int n_pos,
                                      int n_types, int pos_off, int dtype, void* out, hipStream_t s) {
    if (dtype == CMR_DT_BF16) return launch_embed_ln<CMR_DT_BF16>(ids, tt, word, pos, type, gamma, beta, eps, rows, L, d, vocab, n_pos, n_types, pos_off, out, s);
    return launch_embed_ln<CMR_DT_F16>(ids, tt, word, pos, type, gamma, beta, eps, rows, L, d, vocab, n_pos, n_types, pos_off, out, s);
}

hipError_t cmr_launch_add_layernorm(const void* y, const void* bias, const void* res, const void* gamma, const void* beta, float eps, long long rows,
                                    int d, int dtype, void* out, hipStream_t s) {
    if (dtype == CMR_DT_BF16) return launch_add_ln<CMR_DT_BF16>(y, bias, res, gamma, beta, eps, rows, d, out, s);
    return launch_add_ln<CMR_DT_F16>(y, bias, res, gamma, beta, eps, rows, d, out, s);
}
