// tools/ubench.hip -- instruction-throughput microbenchmark (not part of the product).
// Every lane runs 8 independent dependency chains of one operation; 256 CUs x 32 waves.
// Reports wave-instructions per ns per SIMD, and the ratio to v_fma_f32 (= 2 cycles/wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)

#define ITER 2000
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed)
{
    unsigned a[8]; float f[8]; double d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 7 + i; f[i] = 1.0f + i + (seed & 3); d[i] = 1.0 + i + (seed & 3); }
    const unsigned c1 = seed | 0x01010101u; const float fc = 1.0001f + seed; const double dc = 1.0001 + seed;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) f[i] = __builtin_fmaf(f[i], fc, fc);
            if (OP == 1) a[i] = __builtin_amdgcn_udot4(a[i], c1, a[i], false);
            if (OP == 2) a[i] = __builtin_amdgcn_alignbyte(a[i], c1, 3);
            if (OP == 3) a[i] = (a[i] << 8) + c1;
            if (OP == 4) f[i] = (float)(a[i] >> 24) , a[i] += 1u << 24;      // cvt_f32_ubyte3 + add
            if (OP == 5) f[i] = f[i] * fc;
            if (OP == 6) f[i] = f[i] + fc;
            if (OP == 7) d[i] = d[i] + dc;
            if (OP == 8) d[i] = d[i] * dc;
            if (OP == 9) d[i] = __builtin_fma(d[i], dc, dc);
            if (OP == 10) d[i] = 1.0 / d[i];
            if (OP == 11) a[i] = __builtin_amdgcn_ds_bpermute((threadIdx.x * 4 + 28) & 255, a[i]);
            if (OP == 12) a[i] = __builtin_amdgcn_perm(a[i], c1, 0x0c020c00u);
            if (OP == 13) a[i] = a[i] * c1;                                     // v_mul_lo_u32
            if (OP == 14) a[i] = __builtin_amdgcn_sad_u8(a[i], c1, a[i]);
            if (OP == 15) a[i] = (a[i] >> 8) & 0xFF00FFu;                       // 2 ops (or bfe)
            if (OP == 16) f[i] = sqrtf(f[i]);
            if (OP == 17) d[i] = (double)(float)d[i] + dc;                      // cvt pair + add
            if (OP == 18) { f[i] = f[i] + fc; d[i] = d[i] + (double)f[i]; }     // add_f32 + cvt_f64_f32 + add_f64
            if (OP == 19) { f[i] = f[i] + fc; d[i] = d[i] + dc; }               // add_f32 + add_f64 (the same without the conversion)
        }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; i++) r += a[i] + __float_as_uint(f[i]) + (unsigned)__double_as_longlong(d[i]);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
void run(const char* name, int ops_per, double* base)
{
    const int blocks = 256 * 8;                 // 8 workgroups of 4 waves per CU = 8 waves/SIMD
    unsigned* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k<OP><<<blocks, 256>>>(out, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double winst = (double)blocks * 4 * ITER * 8 * ops_per;      // wave-instructions
    const double per_simd_ns = winst / 1024.0 / (ms * 1e6);
    if (*base == 0) *base = per_simd_ns;
    printf("%-28s %8.3f ms  %7.4f winst/ns/SIMD  cycles(rel fma=2): %6.2f\n", name, ms, per_simd_ns, 2.0 * *base / per_simd_ns);
    hipFree(out);
}

int main()
{
    double base = 0;
    run<0>("v_fma_f32", 1, &base);
    run<1>("v_dot4_u32_u8", 1, &base);
    run<2>("v_alignbyte_b32", 1, &base);
    run<3>("v_lshl_add_u32", 1, &base);
    run<4>("cvt_f32_ubyte3 + add_u32", 2, &base);
    run<5>("v_mul_f32", 1, &base);
    run<6>("v_add_f32", 1, &base);
    run<7>("v_add_f64", 1, &base);
    run<8>("v_mul_f64", 1, &base);
    run<9>("v_fma_f64", 1, &base);
    run<10>("f64 1/x (ieee sequence)", 1, &base);
    run<11>("ds_bpermute_b32", 1, &base);
    run<12>("v_perm_b32", 1, &base);
    run<13>("v_mul_lo_u32", 1, &base);
    run<14>("v_sad_u8", 1, &base);
    run<15>("lshr + and", 2, &base);
    run<16>("sqrtf (ieee)", 1, &base);
    run<17>("cvt f64->f32->f64 + add", 3, &base);
    run<18>("add_f32 + cvt_f64_f32 + add_f64", 3, &base);
    run<19>("add_f32 + add_f64", 2, &base);
    return 0;
}
