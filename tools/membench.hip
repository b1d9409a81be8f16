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

// k_uv's D-write pattern.  MODE 0 (as shipped): per 2 rows, 10 stores of 8 B per lane, 8 lanes = one 64-B half line,
// every 8-lane group in another tile.  MODE 1: per 8 rows, 15 stores of 16 B per lane, 32 lanes = 512 contiguous bytes.
template <int MODE>
__global__ __launch_bounds__(256) void write_uv(double* __restrict__ D)
{
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= kPairs * 7) return;
    const int p = wid / 7, strip = wid - p * 7;
    const int64_t pairbase = (int64_t)p * kPairTiles * 512;
    if (MODE == 0) {
        const int xl = strip * 48 - 8 + lane;
        const bool writer = lane >= 8 && lane < 56 && xl < 320;
        const int x = min(max(xl, 0), 319);
        for (int y0 = 0; y0 < 320; y0 += 2)
            if (writer)
#pragma unroll
                for (int c = 0; c < 5; c++)
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const int y = y0 + r;
                        D[pairbase + ((int64_t)(y >> 6) * 5 * kXch + c * kXch + (x >> 3)) * 512 + (y & 63) * 8 + ((x & 7) ^ (y & 7))] = 1.0 + y;
                    }
    } else {
        const int ntile = strip == 6 ? 4 : 6;
        for (int yb = 0; yb < 320; yb += 8)
#pragma unroll
            for (int i = 0; i < 15; i++) {
                const int e = i * 128 + lane * 2, c = e / 384, rem = e - c * 384, tile = rem >> 6, within = rem & 63;
                if (tile < ntile)
                    *reinterpret_cast<dbl2*>(D + pairbase + ((int64_t)(yb >> 6) * 5 * kXch + c * kXch + strip * 6 + tile) * 512 +
                                             (yb & 63) * 8 + within) = dbl2{1.0 + yb, 2.0};
            }
    }
}

// k_uv's whole traffic without its arithmetic: per row and lane R0 (5 floats), the two R1 gather rows (10 floats
// each, at the unwarped position), flow (2 floats) in; D (5 doubles, tile pattern) out.  UNR rows in flight.
struct __attribute__((packed, aligned(4))) F4 { float a, b, c, d; };
struct __attribute__((packed, aligned(4))) F2 { float a, b; };
template <int UNR, bool STORE>
__global__ __launch_bounds__(256) void uv_traffic(const float* __restrict__ R, const float* __restrict__ flow, double* __restrict__ D)
{
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= kPairs * 7) return;
    const int p = wid / 7, strip = wid - p * 7;
    const int xl = strip * 48 - 8 + lane;
    const bool writer = lane >= 8 && lane < 56 && xl < 320;
    const int x = min(max(xl, 0), 318);
    const int64_t pairbase = (int64_t)p * kPairTiles * 512;
    const float* R0 = R + (int64_t)p * 5 * 102400; const float* R1 = R0 + 5 * 102400;
    const float* fl = flow + (int64_t)p * 2 * 102400;
    float acc = 0.f;
    for (int y0 = 0; y0 < 320; y0 += UNR) {
        F4 a[UNR], t0[UNR], t1[UNR], b0[UNR], b1[UNR]; F2 t2[UNR], b2[UNR]; float a4[UNR], dx[UNR], dy[UNR];
#pragma unroll
        for (int r = 0; r < UNR; r++) {
            const int y = min(y0 + r, 318), o = y * 320 + x;
            dx[r] = fl[o]; dy[r] = fl[102400 + o];
            const float* q = R0 + o * 5; a[r] = *(const F4*)q; a4[r] = q[4];
            const float* g = R1 + o * 5;
            t0[r] = *(const F4*)g; t1[r] = *(const F4*)(g + 4); t2[r] = *(const F2*)(g + 8);
            b0[r] = *(const F4*)(g + 1600); b1[r] = *(const F4*)(g + 1604); b2[r] = *(const F2*)(g + 1608);
        }
#pragma unroll
        for (int r = 0; r < UNR; r++) {
            const float v = dx[r] + dy[r] + a[r].a + a[r].d + a4[r] + t0[r].a + t1[r].b + t2[r].a + b0[r].c + b1[r].d + b2[r].b;
            acc += v;
            if (STORE && writer) {
                const int y = y0 + r;
#pragma unroll
                for (int c = 0; c < 5; c++)
                    D[pairbase + ((int64_t)(y >> 6) * 5 * kXch + c * kXch + (x >> 3)) * 512 + (y & 63) * 8 + ((x & 7) ^ (y & 7))] = v + c;
            }
        }
    }
    if (acc == 12345.678f) D[0] = acc;
}

// loads two rows at a time (as uv_traffic<2>), but the D stores of SB rows leave together in one burst
template <int SB>
__global__ __launch_bounds__(256) void uv_burst(const float* __restrict__ R, const float* __restrict__ flow, double* __restrict__ D)
{
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= kPairs * 7) return;
    const int p = wid / 7, strip = wid - p * 7;
    const int xl = strip * 48 - 8 + lane;
    const bool writer = lane >= 8 && lane < 56 && xl < 320;
    const int x = min(max(xl, 0), 318);
    const int64_t pairbase = (int64_t)p * kPairTiles * 512;
    const float* R0 = R + (int64_t)p * 5 * 102400; const float* R1 = R0 + 5 * 102400;
    const float* fl = flow + (int64_t)p * 2 * 102400;
    for (int yb = 0; yb < 320; yb += SB) {
        float vrow[SB];
#pragma unroll
        for (int y0 = 0; y0 < SB; y0 += 2) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int y = min(yb + y0 + r, 318), o = y * 320 + x;
                const float* q = R0 + o * 5; const float* g = R1 + o * 5;
                const F4 a = *(const F4*)q, t0 = *(const F4*)g, t1 = *(const F4*)(g + 4), b0 = *(const F4*)(g + 1600), b1 = *(const F4*)(g + 1604);
                const F2 t2 = *(const F2*)(g + 8), b2 = *(const F2*)(g + 1608);
                vrow[y0 + r] = fl[o] + fl[102400 + o] + a.a + a.d + q[4] + t0.a + t1.b + t2.a + b0.c + b1.d + b2.b;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (writer)
#pragma unroll
            for (int r = 0; r < SB; r++) {
                const int y = yb + r;
#pragma unroll
                for (int c = 0; c < 5; c++)
                    D[pairbase + ((int64_t)(y >> 6) * 5 * kXch + c * kXch + (x >> 3)) * 512 + (y & 63) * 8 + ((x & 7) ^ (y & 7))] = vrow[r] + c;
            }
    }
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
    timeit("k_uv D writes: 8 B/lane, 128 B per tile per step", bytes, [&] { write_uv<0><<<(kPairs * 7 + 3) / 4, 256>>>(b); });
    timeit("k_uv D writes: 16 B/lane, 512 B per tile per flush", bytes, [&] { write_uv<1><<<(kPairs * 7 + 3) / 4, 256>>>(b); });
    {
        float *R, *fl;
        hipMalloc(&R, (size_t)(kPairs + 1) * 5 * 102400 * 4); hipMalloc(&fl, (size_t)kPairs * 2 * 102400 * 4);
        hipMemset(R, 0, (size_t)(kPairs + 1) * 5 * 102400 * 4); hipMemset(fl, 0, (size_t)kPairs * 2 * 102400 * 4);
        const double rd = (double)kPairs * 102400 * (20 + 40 + 8), wr = (double)kPairs * 102400 * 40;
        const int nb = (kPairs * 7 + 3) / 4;
        timeit("k_uv traffic, loads only, 2 rows in flight", rd, [&] { uv_traffic<2, false><<<nb, 256>>>(R, fl, b); });
        timeit("k_uv traffic, loads only, 4 rows in flight", rd, [&] { uv_traffic<4, false><<<nb, 256>>>(R, fl, b); });
        timeit("k_uv traffic, loads only, 8 rows in flight", rd, [&] { uv_traffic<8, false><<<nb, 256>>>(R, fl, b); });
        {
            hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
            hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
            for (int rep = 0; rep < 3; rep++) {
                hipDeviceSynchronize();
                hipEventRecord(e0, s1); hipStreamWaitEvent(s2, e0, 0);
                uv_traffic<4, false><<<nb, 256, 0, s1>>>(R, fl, b);
                write_uv<0><<<nb, 256, 0, s2>>>(a);
                hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0);
                hipEventRecord(e1, s1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("loads (4 rows in flight) and D stores as two concurrent kernels: %8.1f us  %6.2f TB/s\n", ms * 1e3, (rd + wr) / (ms * 1e-3) / 1e12);
            }
        }
        timeit("2-row loads, D stores in bursts of 2 rows", rd + wr, [&] { uv_burst<2><<<nb, 256>>>(R, fl, b); });
        timeit("2-row loads, D stores in bursts of 8 rows", rd + wr, [&] { uv_burst<8><<<nb, 256>>>(R, fl, b); });
        timeit("2-row loads, D stores in bursts of 16 rows", rd + wr, [&] { uv_burst<16><<<nb, 256>>>(R, fl, b); });
        timeit("2-row loads, D stores in bursts of 32 rows", rd + wr, [&] { uv_burst<32><<<nb, 256>>>(R, fl, b); });
        timeit("k_uv traffic, loads + D stores, 2 rows in flight", rd + wr, [&] { uv_traffic<2, true><<<nb, 256>>>(R, fl, b); });
        timeit("k_uv traffic, loads + D stores, 4 rows in flight", rd + wr, [&] { uv_traffic<4, true><<<nb, 256>>>(R, fl, b); });
        timeit("k_uv traffic, loads + D stores, 8 rows in flight", rd + wr, [&] { uv_traffic<8, true><<<nb, 256>>>(R, fl, b); });
    }
    return 0;
}
