// tools/mallbench.hip -- does a buffer that was just WRITTEN come back from the Infinity Cache when it is read (or
// written again) right after, as a function of its size?  (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void wr(dbl2* __restrict__ d, long n, double v)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = dbl2{v, v + i};
}
__global__ __launch_bounds__(256) void rd(const dbl2* __restrict__ s, long n, double* out)
{
    dbl2 a = {0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a += s[i];
    if (a.x + a.y == 12345.678) out[0] = a.x;
}
int main()
{
    dbl2 *buf, *other; double* out;
    const long maxb = 1024l << 20;
    hipMalloc(&buf, maxb); hipMalloc(&other, maxb); hipMalloc(&out, 64);
    hipMemset(buf, 0, maxb); hipMemset(other, 0, maxb);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    for (int mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
        const long n = (long)mb * (1 << 20) / 16;
        float tw = 0, tr = 0, tw2 = 0, trc = 0;
        for (int rep = 0; rep < 4; rep++) {
            rd<<<2048, 256>>>(other, maxb / 16, out);              // flush: stream 1 GiB of something else
            hipEventRecord(e0); wr<<<2048, 256>>>(buf, n, 1.0 + rep); hipEventRecord(e1);
            rd<<<2048, 256>>>(buf, n, out); hipEventRecord(e2); hipEventSynchronize(e2);
            float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
            if (rep) { tw += a; tr += b; }
            hipEventRecord(e0); wr<<<2048, 256>>>(buf, n, 2.0 + rep); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&a, e0, e1); if (rep) tw2 += a;                 // write again right after (lines resident?)
            rd<<<2048, 256>>>(other, maxb / 16, out);
            hipEventRecord(e0); rd<<<2048, 256>>>(buf, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&a, e0, e1); if (rep) trc += a;                 // cold read (after 1 GiB of other traffic)
        }
        const double gb = mb * 1.048576e-3;
        printf("%5d MB: write (cold) %6.2f TB/s | read right after the write %6.2f TB/s | write again %6.2f TB/s | cold read %6.2f TB/s\n",
               mb, gb / (tw / 3), gb / (tr / 3), gb / (tw2 / 3), gb / (trc / 3));
    }
    return 0;
}
