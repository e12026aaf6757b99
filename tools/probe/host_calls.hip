// Host-side cost of the HIP calls an enqueued pass is made of (hipcc --offload-arch=gfx950 -O2 host_calls.hip -o host_calls)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void nop(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t e[64]; for (auto& x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < N; ++i) hipFuncSetAttribute(reinterpret_cast<const void*>(nop), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        double t1 = now(); printf("hipFuncSetAttribute        %.2f us\n", (t1 - t0) / N);
        t0 = now(); for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a, nullptr); if ((i & 63) == 63) hipStreamSynchronize(a); } t1 = now();
        printf("launch (one stream)        %.2f us\n", (t1 - t0) / N); hipStreamSynchronize(a);
        t0 = now(); for (int i = 0; i < N; ++i) { hipEventRecord(e[i & 63], a); if ((i & 63) == 63) hipStreamSynchronize(a); } t1 = now();
        printf("hipEventRecord             %.2f us\n", (t1 - t0) / N); hipStreamSynchronize(a);
        t0 = now(); for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a, nullptr); hipEventRecord(e[i & 63], a); hipStreamWaitEvent(b, e[i & 63], 0); hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, b, nullptr);
            if ((i & 31) == 31) { hipStreamSynchronize(a); hipStreamSynchronize(b); } } t1 = now();
        printf("launch a + record + wait on b + launch b   %.2f us\n", (t1 - t0) / N);
        hipDeviceSynchronize();
    }
    return 0;
}
