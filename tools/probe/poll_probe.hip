// Completion of a short synchronous launch: hipStreamSynchronize against polling a word the kernel's last act stores into pinned,
// device-mapped host memory (results first, s_waitcnt vmcnt(0), then the word).  hipcc --offload-arch=gfx950 -O2 poll_probe.hip -o poll_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include <immintrin.h>
__global__ void work(const float* in, float* out, int* done, int spin) {
    float v = in[threadIdx.x & 63];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    out[threadIdx.x] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (done && threadIdx.x == 0) *(volatile int*)done = 1;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    char* h; hipHostMalloc((void**)&h, 1 << 16, hipHostMallocDefault);
    char* d; hipHostGetDevicePointer((void**)&d, h, 0);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    memset(h, 0, 1 << 16);
    for (int spin : {0, 2000, 20000}) {
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<double> t;
            for (int it = 0; it < 300; ++it) {
                *(volatile int*)h = 0;
                const double t0 = now();
                hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, s, (const float*)(d + 4096), (float*)(d + 8192), mode ? (int*)d : nullptr, spin);
                if (mode) { while (!*(volatile int*)h) _mm_pause(); }
                else hipStreamSynchronize(s);
                t.push_back(now() - t0);
                if (mode) hipStreamSynchronize(s);
            }
            std::sort(t.begin(), t.end());
            printf("kernel spin %6d  %s: median %.1f us  min %.1f  p90 %.1f\n", spin, mode ? "poll mapped word " : "hipStreamSynchronize", t[150], t[0], t[270]);
        }
    }
    return 0;
}
