// ldsdma_bench.hip -- how fast can a CU fill LDS with global_load_lds_dwordx4 (LDS-DMA)?  The ceiling the matrix-core
// kernels' operand staging runs against.  One 512-thread workgroup per CU, every wave issues 1-KiB pieces (16 B per lane)
// into a ring of LDS slots and waits with a counted vmcnt so that `depth` pieces per wave stay in flight; nothing reads
// the LDS.  Sources: (a) a small buffer every workgroup reads (L2 hits after the first touch: the weight operand),
// (b) a large buffer streamed once (HBM / Infinity Cache: the activation operand), (c) 1 part of (b) to 2 parts of (a)
// per piece pair, roughly the patch-embed GEMM's mix (A is fetched from HBM by one of three column tiles).
//   hipcc --offload-arch=gfx950 -O3 tools/ldsdma_bench.hip -o /tmp/ldsdma_bench && /tmp/ldsdma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(512) void k_fill(const char* __restrict__ small, size_t small_bytes, const char* __restrict__ big,
                                             size_t big_bytes, int pieces, int mix)
{
    extern __shared__ __align__(16) char lds[];            // 8 waves x DEPTH slots x 1 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* my = lds + wave * DEPTH * 1024;
    // each wave walks its own sequence of 1-KiB pieces
    const size_t wid = (size_t)blockIdx.x * 8 + wave, nw = (size_t)gridDim.x * 8;
    for (int i = 0; i < pieces; i++) {
        const size_t seq = (size_t)i * nw + wid;
        const char* src;
        const bool from_big = mix == 1 || mix == 4 || (mix == 2 && (i % 3) == 0);
        if (from_big) src = big + (seq * 1024) % big_bytes;
        else src = small + (((size_t)i * 8 + wave + (size_t)blockIdx.x * (mix == 3 ? 0 : 37)) * 1024) % small_bytes;   // every workgroup reads the same buffer, at its own phase (mix 3: all in step)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 16),
                                         (__attribute__((address_space(3))) void*)(my + (i % DEPTH) * 1024), 16, 0, 0);
        if (i >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the same traffic through ordinary loads: 16 bytes per lane into registers (global_load_dwordx4), DEPTH loads in flight
template <int DEPTH>
__global__ __launch_bounds__(512) void k_vgpr(const char* __restrict__ small, size_t small_bytes, const char* __restrict__ big,
                                             size_t big_bytes, int pieces, int mix)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wid = (size_t)blockIdx.x * 8 + wave, nw = (size_t)gridDim.x * 8;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < pieces; i0 += DEPTH) {
        f4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const int i = i0 + d;
            const size_t seq = (size_t)i * nw + wid;
            const bool from_big = mix == 1 || mix == 4 || (mix == 2 && (i % 3) == 0);
            const char* src = from_big ? big + (seq * 1024) % big_bytes
                                       : small + (((size_t)i * 8 + wave + (size_t)blockIdx.x * 37) * 1024) % small_bytes;
            v[d] = *reinterpret_cast<const f4*>(src + lane * 16);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) acc += v[d];
    }
    if (acc[0] == 123.456f) reinterpret_cast<f4*>(const_cast<char*>(big))[threadIdx.x] = acc;    // never true: keeps the loads
}

template <int DEPTH>
void run_vgpr(const char* name, int mix, const char* small, size_t sb, const char* big, size_t bb, int cus)
{
    const int pieces = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_vgpr<DEPTH>, dim3(cus), dim3(512), 0, 0, small, sb, big, bb, pieces, mix);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_vgpr<DEPTH>, dim3(cus), dim3(512), 0, 0, small, sb, big, bb, pieces, mix);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 3.0 * cus * 8 * pieces * 1024.0;
    printf("%-34s depth %2d: %7.2f TB/s  (%5.1f GB/s per CU)   [loads into registers]\n", name, DEPTH, bytes / (ms * 1e-3) / 1e12,
           bytes / (ms * 1e-3) / 1e9 / cus);
}

template <int DEPTH>
void run(const char* name, int mix, const char* small, size_t sb, const char* big, size_t bb, int cus)
{
    const int pieces = 4096;                                // per wave: 4 MiB
    const size_t lds = 8 * DEPTH * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_fill<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_fill<DEPTH>, dim3(cus), dim3(512), lds, 0, small, sb, big, bb, pieces, mix);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_fill<DEPTH>, dim3(cus), dim3(512), lds, 0, small, sb, big, bb, pieces, mix);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 3.0 * cus * 8 * pieces * 1024.0;
    printf("%-34s depth %2d: %7.2f TB/s  (%5.1f GB/s per CU)\n", name, DEPTH, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / cus);
}

int main(int argc, char** argv)
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = argc > 1 ? atoi(argv[1]) : p.multiProcessorCount;     // workgroups (one per CU); fewer = part of the chip
    const size_t sb = 1179648, bb = (size_t)768 << 20;      // 1.1 MiB (a 768 x 768 bf16 weight), 768 MiB
    char *small, *big;
    CHECK(hipMalloc(&small, sb)); CHECK(hipMalloc(&big, bb));
    CHECK(hipMemset(small, 1, sb)); CHECK(hipMemset(big, 2, bb));
    run<4>("all workgroups, one small buffer", 0, small, sb, big, bb, cus);
    run<12>("all workgroups, one small buffer", 0, small, sb, big, bb, cus);
    run<16>("all workgroups, one small buffer", 0, small, sb, big, bb, cus);
    run<4>("large buffer streamed once", 1, small, sb, big, bb, cus);
    run<12>("large buffer streamed once", 1, small, sb, big, bb, cus);
    run<16>("large buffer streamed once", 1, small, sb, big, bb, cus);
    run<12>("small buffer, all workgroups in step", 3, small, sb, big, bb, cus);
    run<12>("64 MiB buffer (Infinity Cache)", 4, small, sb, big, (size_t)64 << 20, cus);
    run<12>("1 : 2 mix (GEMM-like)", 2, small, sb, big, bb, cus);
    run<16>("1 : 2 mix (GEMM-like)", 2, small, sb, big, bb, cus);
    run_vgpr<8>("all workgroups, one small buffer", 0, small, sb, big, bb, cus);
    run_vgpr<16>("all workgroups, one small buffer", 0, small, sb, big, bb, cus);
    run_vgpr<16>("large buffer streamed once", 1, small, sb, big, bb, cus);
    run_vgpr<16>("1 : 2 mix (GEMM-like)", 2, small, sb, big, bb, cus);
    return 0;
}
