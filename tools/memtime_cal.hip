// calibrate s_memtime: ticks per microsecond of wall time, idle chip vs busy chip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long* out, int iters, float seed)
{
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    float f = seed + threadIdx.x;
    for (int i = 0; i < iters; i++) f = __builtin_fmaf(f, 1.0001f, 0.5f);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; }
    if (f == 12345.f) out[1] = 1;
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 256, 2048, 8192}) {
        spin<<<blocks, 256>>>(d, 1000, 1.f); hipDeviceSynchronize();
        hipEventRecord(e0); spin<<<blocks, 256>>>(d, 400000, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("blocks %5d: wall %8.1f us, s_memtime delta %10llu ticks -> %7.1f ticks/us; %.2f ticks per fma\n", blocks, ms * 1e3, h[0], h[0] / (ms * 1e3), (double)h[0] / 400000);
    }
    return 0;
}
