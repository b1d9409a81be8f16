// tools/occ_probe.hip -- how many workgroups are really co-resident, as a function of LDS bytes and waves per
// workgroup: every workgroup records its start (s_memrealtime, 100 MHz) and spins ~30 us; workgroups that start
// within the first 5 us are the first residency round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void probe(unsigned long long* start, int spin_ticks)
{
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; lds[0] = 1; }
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[0] == 77) start[0] = 0;
}
int main()
{
    const int nblocks = 4096;
    unsigned long long* d; hipMalloc(&d, nblocks * 8);
    std::vector<unsigned long long> h(nblocks);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int threads : {128, 256, 384}) {
        for (int kb : {1, 16, 30, 32, 40, 50, 60, 64, 80}) {
            probe<<<nblocks, threads, kb * 1024>>>(d, 3000);
            if (hipDeviceSynchronize() != hipSuccess) { printf("threads %d lds %d KB: launch failed\n", threads, kb); (void)hipGetLastError(); continue; }
            hipMemcpy(h.data(), d, nblocks * 8, hipMemcpyDeviceToHost);
            const unsigned long long t0 = *std::min_element(h.begin(), h.end());
            int first = 0;
            for (auto t : h) first += (t - t0) < 500;          // started within 5 us
            printf("threads %3d  lds %3d KB: %4d workgroups in the first round = %.2f per CU\n", threads, kb, first, first / 256.0);
        }
    }
    return 0;
}
