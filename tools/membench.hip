// tools/membench.hip -- HBM access-pattern microbenchmark (not part of the product).
// Calibrates what the level-0 Farneback kernels could reach: linear read / write / copy of a D-sized buffer
// (487 MB) against the k_hscan loader pattern (595 workgroups, five 4 KiB tiles per chunk, 40 chunks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef double dbl2 __attribute__((ext_vector_type(2)));
constexpr int kPairs = 119, kXch = 40, kNyb = 5, kPairTiles = 5 * kXch * kNyb + 1;   // as d16_pair_tiles(320)
constexpr int64_t kDoubles = (int64_t)kPairs * kPairTiles * 512;

__global__ __launch_bounds__(256) void read_linear(const dbl2* __restrict__ src, int64_t n, double* out)
{
    dbl2 acc = {0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += src[i];
    if (acc.x + acc.y == 12345.678) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void write_linear(dbl2* __restrict__ dst, int64_t n)
{
    const dbl2 v = {1.0, 2.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = v;
}
__global__ __launch_bounds__(256) void copy_linear(const dbl2* __restrict__ src, dbl2* __restrict__ dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

// k_hscan loader pattern.  LAYOUT 0: tiles [c][xc] (as shipped: a chunk = 5 pieces of 4 KiB, 160 KiB apart);
// LAYOUT 1: tiles [xc][c] (a chunk = 20 KiB contiguous).  DEPTH chunks in flight per wave, WAVES loader waves.
template <int LAYOUT, int DEPTH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void read_tiles(const double* __restrict__ D, double* out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x / kNyb, ybk = blockIdx.x - p * kNyb;
    const double* tiles = D + ((int64_t)p * kPairTiles + (int64_t)ybk * 5 * kXch) * 512;
    dbl2 acc = {0, 0};
    for (int xc0 = wave * DEPTH; xc0 < kXch; xc0 += WAVES * DEPTH) {
        dbl2 r[DEPTH][5][4];
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int c = 0; c < 5; c++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int xc = min(xc0 + d, kXch - 1);
                    const int64_t t = LAYOUT == 0 ? (int64_t)c * kXch + xc : (int64_t)xc * 5 + c;
                    r[d][c][i] = *reinterpret_cast<const dbl2*>(tiles + t * 512 + i * 128 + lane * 2);
                }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int c = 0; c < 5; c++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc += r[d][c][i];
    }
    if (acc.x + acc.y == 12345.678) out[0] = acc.x;
}

template <typename F>
static void timeit(const char* name, double bytes, F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-52s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
    double *a, *b, *out;
    const int64_t bytes = kDoubles * 8;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 64);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    printf("buffer %.1f MB\n", bytes / 1e6);
    const int64_t n2 = kDoubles / 2;
    for (int blocks : {1024, 2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "read_linear  %d blocks", blocks);
        timeit(nm, bytes, [&] { read_linear<<<blocks, 256>>>((const dbl2*)a, n2, out); });
        snprintf(nm, 64, "write_linear %d blocks", blocks);
        timeit(nm, bytes, [&] { write_linear<<<blocks, 256>>>((dbl2*)b, n2); });
        snprintf(nm, 64, "copy_linear  %d blocks (read+write bytes)", blocks);
        timeit(nm, 2.0 * bytes, [&] { copy_linear<<<blocks, 256>>>((const dbl2*)a, (dbl2*)b, n2); });
    }
    const int nb = kPairs * kNyb;
    timeit("tiles [c][xc] depth1 1 wave", bytes, [&] { read_tiles<0, 1, 1><<<nb, 64>>>(a, out); });
    timeit("tiles [c][xc] depth2 1 wave", bytes, [&] { read_tiles<0, 2, 1><<<nb, 64>>>(a, out); });
    timeit("tiles [c][xc] depth4 1 wave", bytes, [&] { read_tiles<0, 4, 1><<<nb, 64>>>(a, out); });
    timeit("tiles [c][xc] depth2 2 waves", bytes, [&] { read_tiles<0, 2, 2><<<nb, 128>>>(a, out); });
    timeit("tiles [c][xc] depth2 4 waves", bytes, [&] { read_tiles<0, 2, 4><<<nb, 256>>>(a, out); });
    timeit("tiles [xc][c] depth2 1 wave", bytes, [&] { read_tiles<1, 2, 1><<<nb, 64>>>(a, out); });
    timeit("tiles [xc][c] depth4 1 wave", bytes, [&] { read_tiles<1, 4, 1><<<nb, 64>>>(a, out); });
    timeit("tiles [xc][c] depth2 2 waves", bytes, [&] { read_tiles<1, 2, 2><<<nb, 128>>>(a, out); });
    timeit("tiles [xc][c] depth2 4 waves", bytes, [&] { read_tiles<1, 2, 4><<<nb, 256>>>(a, out); });
    return 0;
}
