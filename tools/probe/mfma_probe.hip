// Micro-probe (developer tool): cycles per v_mfma_f32_32x32x16_bf16 for a single wave per SIMD with 1 / 2 / 3 / 4
// independent accumulator chains, B operand in VGPRs or AGPRs, with and without an instruction between the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned v4u;

template <int CH, int AB, int FILL>
__global__ __launch_bounds__(256, 1) void probe(const v4u* in, float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    v4u a = in[lane], b0 = in[64 + lane], b1 = in[128 + lane], b2 = in[192 + lane], b3 = in[256 + lane];
    f32x16 c0, c1, c2, c3;
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(c0) : "v"(a), "v"(b0));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(c1) : "v"(a), "v"(b1));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(c2) : "v"(a), "v"(b2));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(c3) : "v"(a), "v"(b3));
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
#define MF(c, b)                                                                                        \
    if (AB) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "a"(b));          \
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));            \
    if (FILL == 1) asm volatile("s_nop 0");                                                              \
    if (FILL == 2) asm volatile("v_mov_b32 %0, %0" : "+v"(a.x));
            if (CH == 22) { MF(c0, b0) MF(c0, b2) MF(c1, b1) MF(c1, b3) }                 // two chains, pairs back to back
            else if (CH == 24) { MF(c0, b0) MF(c0, b2) MF(c0, b0) MF(c0, b2) MF(c1, b1) MF(c1, b3) MF(c1, b1) MF(c1, b3) }
            else {
            MF(c0, b0)
            if (CH >= 2) { MF(c1, b1) }
            if (CH >= 3) { MF(c2, b2) }
            if (CH >= 4) { MF(c3, b3) }
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH, int AB, int FILL>
void run(const char* name, const v4u* in, float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<CH, AB, FILL>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<CH, AB, FILL>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 12 * (CH == 22 ? 4 : CH == 24 ? 8 : CH);
    printf("%-44s chains %d B-in-%s fill %d: %.1f s_memtime ticks / MFMA, %.2f ns / MFMA (wall), %.0f TFLOP/s chip\n", name, CH, AB ? "AGPR" : "VGPR", FILL,
           (double)h / n, ms * 1e6 / n, 2.0 * 32 * 32 * 16 * n * 1024 / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    v4u* in; float* out; long long* cyc;
    hipMalloc(&in, 320 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    unsigned short h[320 * 8];
    // operand values: argv[1] = "unit" -> bf16 of N(0, 1/768) components (what a normalised 768-d corpus holds: random sign,
    // exponent and mantissa bits); default -> values near 1.0 with random low mantissa bits (little toggling)
    const bool unit = argc > 1 && argv[1][0] == 'u';
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < 320 * 8; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        if (unit) {
            double u1 = ((st >> 11) + 1) / 9007199254740993.0; st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            double u2 = (st >> 11) / 9007199254740992.0;
            float f = (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2) / sqrt(768.0));
            unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
        } else h[i] = 0x3c00 + (unsigned short)(st & 0x3ff);
    }
    printf("operands: %s\n", unit ? "N(0, 1/768) bf16" : "~1.0 bf16");
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<1, 0, 0>("1 chain back-to-back", in, out, cyc);
    run<1, 0, 1>("1 chain + s_nop between", in, out, cyc);
    run<2, 0, 0>("2 chains", in, out, cyc);
    run<2, 1, 0>("2 chains, B in AGPR", in, out, cyc);
    run<2, 0, 1>("2 chains + s_nop after each", in, out, cyc);
    run<2, 0, 2>("2 chains + VALU after each", in, out, cyc);
    run<22, 0, 0>("2 chains, AABB order", in, out, cyc);
    run<22, 1, 0>("2 chains, AABB order, B in AGPR", in, out, cyc);
    run<24, 0, 0>("2 chains, AAAABBBB order", in, out, cyc);
    run<3, 0, 0>("3 chains", in, out, cyc);
    run<3, 0, 1>("3 chains + s_nop after each", in, out, cyc);
    run<4, 0, 0>("4 chains", in, out, cyc);
    run<4, 1, 0>("4 chains, B in AGPR", in, out, cyc);
    run<4, 0, 1>("4 chains + s_nop after each", in, out, cyc);
    run<4, 0, 2>("4 chains + VALU after each", in, out, cyc);
    return 0;
}
