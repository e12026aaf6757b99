// Device-side helpers shared by the gfx950 kernels of libcomorag_hip.so.
//
// Candidate keys.  Every (score, row) pair on the device is one u64 whose unsigned order IS the
// exported result order (include/comorag_hip.h): score descending, then row ascending.
//   hi 32 bits: order-preserving map of the fp32 score (-0.0 canonicalised to +0.0)
//   lo 32 bits: 0xFFFFFFFF - row                         (smaller row => larger key)
// key 0 never occurs for a real (non-NaN) score and marks an empty slot.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned v4u;  // native 128-bit vector (asm-constraint friendly)

#define CMR_DT_F32 0
#define CMR_DT_BF16 1
#define CMR_DT_F16 2

#define CMR_PANEL_ROWS 32   // rows per panel == MFMA M (32x32 tiles)
#define CMR_SCAN_THREADS 512
#define CMR_SCAN_WAVES 8
#define CMR_CORPUS_SLACK (128 * 1024)  // bytes readable past the last panel (load rings over-read up to 88 KiB)

__device__ __forceinline__ u64 cmr_make_key(float v, unsigned row) {
    unsigned u = __float_as_uint(v + 0.0f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((u64)u << 32) | (u64)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ float cmr_key_score(u64 key) {
    unsigned u = (unsigned)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ unsigned cmr_key_row(u64 key) { return 0xFFFFFFFFu - (unsigned)key; }

// fp32 -> bf16 bits, round-to-nearest-even (finite inputs; matches torch / numpy bf16 casts)
__device__ __forceinline__ unsigned short cmr_f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float cmr_bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short cmr_f2h(float f) {
    _Float16 h = (_Float16)f;  // RNE
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float cmr_h2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }

// One "block" = 1 KiB = what one wave loads with one global_load_dwordx4: the A- (or B-)
// operand of the panel's MFMAs for one k-step, already in lane order.
//   16-bit dtypes: 32 rows x 16 k.  lane l holds row (l & 31), k = 16*ks + 8*(l >> 5) + [0,8)
//   fp32         : 32 rows x  8 k.  lane l holds row (l & 31), k =  8*ks + 2*s + (l >> 5), s = 0..3
// (any k permutation is fine for a dot product as long as corpus and query blocks agree).
// The wide kernel issues its MFMAs as inline asm so that the register FILE of every operand is fixed by the
// constraint (v = VGPR half, a = AGPR half of the unified 512-entry file): left to hipcc, B-operands that live in
// AGPRs are shuttled through v_accvgpr_read in front of every MFMA as soon as the loop body contains a branch.
// hipcc pads no hazards around these statements (guide §5.7 item 2): the first MFMA of a chain takes the literal 0
// as C (no VALU-written accumulator), chains accumulate in place (0 wait states), and the caller ends a chain
// with cmr_mfma_drain() before any VALU / v_accvgpr_read touches the accumulators.
//   AB: 0 = B in VGPRs, 1 = B in AGPRs;  FIRST: C = 0
#define CMR_MFMA_ASM(MNEMONIC)                                                                                      \
    static __device__ __forceinline__ void mma_asm(int ab, bool first, f32x16& c, const v4u& a, const v4u& b) {      \
        /* ab / first are constants after unrolling: exactly one statement survives */                               \
        if (first) {                                                                                                 \
            if (ab) asm volatile(MNEMONIC " %0, %1, %2, 0" : "=&a"(c) : "v"(a), "a"(b));                              \
            else asm volatile(MNEMONIC " %0, %1, %2, 0" : "=&a"(c) : "v"(a), "v"(b));                                 \
        } else {                                                                                                     \
            if (ab) asm volatile(MNEMONIC " %0, %1, %2, %0" : "+a"(c) : "v"(a), "a"(b));                              \
            else asm volatile(MNEMONIC " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));                                 \
        }                                                                                                            \
    }                                                                                                                \
    /* the same with the accumulator in the VGPR half (its readers then need no v_accvgpr_read) */                    \
    static __device__ __forceinline__ void mma_asm_cv(int ab, bool first, f32x16& c, const v4u& a, const v4u& b) {   \
        if (first) {                                                                                                 \
            if (ab) asm volatile(MNEMONIC " %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));                              \
            else asm volatile(MNEMONIC " %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));                                 \
        } else {                                                                                                     \
            if (ab) asm volatile(MNEMONIC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));                              \
            else asm volatile(MNEMONIC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));                                 \
        }                                                                                                            \
    }
// 8-pass XDL result -> any non-MFMA reader: 12 wait states (guide §5.7 item 2); s_nop 15 = 16
template <int NT> __device__ __forceinline__ void cmr_mfma_drain(f32x16 (&acc)[NT]) {
    if constexpr (NT == 1) asm volatile("s_nop 15" : "+a"(acc[0]));
    else asm volatile("s_nop 15" : "+a"(acc[0]), "+a"(acc[1]));
}

template <int DT> struct CmrBlk;
template <> struct CmrBlk<CMR_DT_BF16> {
    static constexpr int K = 16;
    static __device__ __forceinline__ f32x16 mma(v4u a, v4u b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    CMR_MFMA_ASM("v_mfma_f32_32x32x16_bf16")
};
template <> struct CmrBlk<CMR_DT_F16> {
    static constexpr int K = 16;
    static __device__ __forceinline__ f32x16 mma(v4u a, v4u b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    CMR_MFMA_ASM("v_mfma_f32_32x32x16_f16")
};
template <> struct CmrBlk<CMR_DT_F32> {
    static constexpr int K = 8;
    static __device__ __forceinline__ f32x16 mma(v4u a, v4u b, f32x16 c) {
        // exact fp32: each MFMA is a k-ordered fmaf chain (guide §3 "FP32-input MFMA")
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
        return c;
    }
};

// k index (within the padded row) of element e (0..7 for 16-bit, 0..3 for fp32) of lane l in block ks
template <int DT> __device__ __forceinline__ int cmr_blk_k(int ks, int lane, int e) {
    if (DT == CMR_DT_F32) return 8 * ks + 2 * e + (lane >> 5);
    return 16 * ks + 8 * (lane >> 5) + e;
}

// C/D fragment of the 32x32 MFMAs: lane l, register r -> (row i, col j)
__device__ __forceinline__ int cmr_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// 256-thread block: pick the k largest keys of pool[0..n) (LDS, destroyed) in descending order.
// Keys are unique (distinct rows) so exactly one thread owns each round's winner.
__device__ __forceinline__ void cmr_block_select(u64* pool, int n, int k, u64* out, u64* wbest) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int round = 0; round < k; ++round) {
        u64 best = 0;
        int bi = 0;
        for (int i = tid; i < n; i += 256) {
            u64 v = pool[i];
            if (v > best) { best = v; bi = i; }
        }
        u64 wb = best;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            u64 o = __shfl_xor(wb, off);
            wb = o > wb ? o : wb;
        }
        if (lane == 0) wbest[wave] = wb;
        __syncthreads();
        u64 fb = wbest[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) fb = wbest[w] > fb ? wbest[w] : fb;
        if (fb != 0 && best == fb) pool[bi] = 0;
        if (tid == 0) out[round] = fb;
        __syncthreads();
    }
}
