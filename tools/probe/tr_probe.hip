// What does ds_read_b64_tr_b16 return?  LDS holds lds[i] = i (16-bit); lane l of a wave supplies the address of four consecutive elements
// of "row" key = kb + (i >> 2) at column d0 + 4 * (i & 3) (i = l & 15, kb = 4 * (l >> 5), d0 = 16 * ((l >> 4) & 1)) of a row-major [keys][STRIDE]
// tile; the attention kernel's PV step wants lane l to receive V[kb + 0..3][d0 + (l & 15)].   hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
#define STRIDE 96
__global__ void k(v4s* out) {
    __shared__ unsigned short lds[64 * STRIDE];
    for (int i = threadIdx.x; i < 64 * STRIDE; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, kb = 4 * (l >> 5), d0 = 16 * ((l >> 4) & 1);
    out[l] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(&lds[(kb + (i >> 2)) * STRIDE + d0 + 4 * (i & 3)]));
}
int main() {
    v4s* d; hipMalloc(&d, 64 * sizeof(v4s));
    k<<<1, 64>>>(d);
    v4s h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int kb = 4 * (l >> 5), d0 = 16 * ((l >> 4) & 1);
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int want = (kb + j) * STRIDE + d0 + (l & 15);
            printf(" %5d%s", (int)(unsigned short)h[l][j], (int)(unsigned short)h[l][j] == want ? "" : "!");
            bad += (int)(unsigned short)h[l][j] != want;
        }
        printf("   (want V[%d..%d][%d])\n", kb, kb + 3, d0 + (l & 15));
    }
    printf("mismatches: %d\n", bad);
    return 0;
}
