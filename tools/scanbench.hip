// tools/scanbench.hip -- what does one column step of the horizontal scanner cost?  (not part of the product)
// One wave per workgroup, 40 active lanes with the LDS address pattern of avd_fbfused.hip's scanner (K = 4 segments,
// G = 2 rows, 5 channels, line pitch 322 doubles, segment k in group buffer (j - k) mod 6).  s_memtime around 80 steps.
//   mode 0: dependent v_add_f64 chain only            mode 1: + the independent subtraction per step
//   mode 2: + ds_read_b128 of the next body           mode 3: + ds_write_b128 of g (the full step)
//   mode 4: as 3 with a conflict-free lane->address map (lane * 16 B)
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
typedef double dbl2 __attribute__((ext_vector_type(2)));
constexpr int P = 322, GROUP = 2 * 5 * P, NBUF = 6;

template <int MODE>
__global__ __launch_bounds__(64) void k(long long* out, double seed, int reps)
{
    __shared__ __align__(16) double lds[NBUF * GROUP];
    const int lane = threadIdx.x;
    for (int i = lane; i < NBUF * GROUP; i += 64) lds[i] = seed + i * 1e-3;
    __syncthreads();
    const int seg = lane / 10, line = lane % 10;
    double* q = MODE == 4 ? lds + lane * 2 : lds + ((6 - seg) % 6) * GROUP + line * P + seg * 80;
    dbl2* Q = reinterpret_cast<dbl2*>(q);
    double gs = seed, dl[16], d[16];
    for (int i = 0; i < 16; i++) { dl[i] = seed * i; d[i] = seed + i; }
    long long t0 = 0, t1 = 0;
    if (lane < 40) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int b = 0; b < 5; b++) {
                dbl2 nx[8];
                if (MODE >= 2) {
#pragma unroll
                    for (int t = 0; t < 8; t++) nx[t] = Q[(MODE == 4 ? 64 : 1) * (4 + 8 * b + t)];
                } else {
#pragma unroll
                    for (int t = 0; t < 8; t++) nx[t] = dbl2{gs + t, gs - t};
                }
                double o[16];
#pragma unroll
                for (int kk = 0; kk < 16; kk++) {
                    gs += d[kk];
                    o[kk] = gs;
                    if (MODE >= 1) {
                        const double e = (kk & 1) ? nx[kk >> 1].y : nx[kk >> 1].x;
                        const double l = dl[(kk + 1) & 15];
                        dl[kk] = e;
                        d[kk] = e - l;
                    }
                    if (MODE >= 3 && (kk & 1)) Q[(MODE == 4 ? 64 : 1) * (8 * b + (kk >> 1))] = dbl2{o[kk - 1], o[kk]};
                }
                if (MODE < 3) asm volatile("" ::"v"(o[15]), "v"(o[7]));
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (long long)gs; }
}

template <int MODE>
void run(const char* name, int blocks)
{
    long long* out; hipMalloc(&out, blocks * 16);
    const int reps = 50;
    k<MODE><<<blocks, 64>>>(out, 1.25, reps);
    k<MODE><<<blocks, 64>>>(out, 1.25, reps);
    hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%-44s %d workgroup(s): %6.1f cycles per column step\n", name, blocks, (double)h[0] / (reps * 80));
    hipFree(out);
}

int main()
{
    for (int blocks : {1, 256}) {
        run<0>("dependent v_add_f64 chain", blocks);
        run<1>("+ independent subtraction", blocks);
        run<2>("+ ds_read_b128 (scanner's addresses)", blocks);
        run<3>("+ ds_write_b128 (full step)", blocks);
        run<4>("full step, conflict-free addresses", blocks);
    }
    return 0;
}
