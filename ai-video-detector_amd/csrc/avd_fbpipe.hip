// avd_fbpipe.hip -- FarnebackUpdateFlow_Blur (winsize 15) of the SMALL pyramid levels (160 / 80 / 40 px): the three blur
// iterations of a level PIPELINED inside one launch (gfx950, round 4; fb_mode = fast).
//
// Reference site: cv2.calcOpticalFlowFarneback(prev, cur, None, 0.5, 3, 15, 3, 5, 1.2, 0), app/analyzers/video.py:45.
//
// Why.  At these sizes a launch of avd_fbfast.hip is a latency chain: steps x the slowest wave of a step (44 / 24 / 16 steps of ~1 us,
// profiles/r03_experiments.md 1.4, r04_experiments.md 3), three launches per level, and most of the chip idles meanwhile.  The
// iterations of a level depend on each other only LOCALLY: iteration i + 1 needs, at pixel (x, y), the flow iteration i left at that
// very pixel, and iteration i produces row y as soon as its own box window (rows y - 7 .. y + 7) has passed.  So iteration i + 1 can
// follow iteration i sixteen rows behind, in the same workgroup, at the same time: the level takes H / 2 + 22 steps instead of
// 3 x (H / 4 + 2), and the flow between iterations never leaves the CU (LDS rings).
//
// Arithmetic: exactly avd_fbfast.hip's (cv2's vertical running sums literally, the horizontal 15-column windows summed directly in
// double with the same grouping of four output columns aligned to x % 4 == 0, the same normal equations, solve and ill-posedness
// criterion): the flow is BIT-IDENTICAL to the one-launch-per-iteration kernels (tests/test_gpu_fbfast.py compares all fold masks).
//
// Structure.  One workgroup = one strip of one pair = 15 waves.  A strip outputs OW = 80 columns (40 at 40 px); iteration 3 needs the
// flow of iteration 2 on 7 more columns either side, that one iteration 1's on 7 more, so 128 lane columns o0 - 23 .. o0 + 104 are
// carried (two 64-column blocks; columns outside the image clamp to the edge = cv2's replicated border; the two strips of a 160-px
// pair recompute each other's halo, nothing is exchanged between workgroups).  Per iteration i = 0, 1, 2 five waves:
//   N(i, b)  normal equations of two image rows per step for block b (FarnebackUpdateMatrices incl. border attenuation and the five
//            products), gather issued two rows ahead; initial flow from HBM (i = 0) or from iteration i - 1's LDS ring -> M ring
//   C(i, b)  cv2's vertical running sums, literally: M rows in, vsum rows (double) out
//   X(i)     window sums + 2 x 2 solve for two rows x up to 28 chunks of four columns; flow -> LDS ring of iteration i + 1 (i < 2)
//            or -> HBM (i = 2)
// One workgroup barrier per step of TWO rows (LDS cannot hold four-row groups of three iterations); N works on group u, C on u - 1,
// X on the rows whose window C completed; iteration i + 1 runs SKEW = 8 steps behind iteration i.  Every wave executes exactly
// STEPS barriers (idle steps before and after its active range are counted at compile time).
#include <cstdio>
#include <cstdlib>
#include "avd_internal.h"
#include "avd_fb_device.h"

#pragma clang fp contract(off)

namespace {

constexpr int kM = 7;                 // (winsize - 1) / 2
constexpr int SKEW = 8;               // steps iteration i + 1 runs behind iteration i

typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int W_>
struct PGeo {
    static constexpr int W = W_, H = W_;
    static constexpr int OW = W >= 80 ? 80 : 40;       // output columns of a strip
    static constexpr int NSTRIPS = W / OW;
    static constexpr int XLO = 23;                     // lane column 0 = image column o0 - 23 (3 x 7 halo, rounded so that chunks stay aligned)
    static constexpr int NE = H + kM;                  // entries of the vertical chain: image row min(e, H - 1)
    static constexpr int TN = H / 2;                   // N: local steps 0 .. TN - 1, entries 2 u, 2 u + 1
    static constexpr int TC = H / 2 + 5;               // C: local step u >= 1 processes entries 2 (u - 1), 2 (u - 1) + 1
    static constexpr int TX = H / 2 + 6;               // X: local step u solves rows 2 u - 11, 2 u - 10
    static constexpr int STEPS = 2 * SKEW + TX;
    static constexpr int ROWLEN = 130;                 // doubles per (row slot, channel) line of vsum: index = lane column
    static constexpr int VS_SLOT = 5 * ROWLEN;
    static constexpr int VS_IT = 4 * VS_SLOT;          // doubles per iteration: ring of 4 rows
    static constexpr int M_SLOT = 5 * 64;              // floats of one row of one block
    static constexpr int M_RING = 4 * M_SLOT;          // ring of 4 rows per (iteration, block)
    static constexpr int F_ROW = 2 * 128;              // floats of one flow row (two components x 128 lane columns)
    static constexpr int F_RING = 8 * F_ROW;           // ring of 8 rows per iteration boundary
    static constexpr int LDS_BYTES = 3 * VS_IT * 8 + 6 * M_RING * 4 + 2 * F_RING * 4;
    static_assert(LDS_BYTES <= 163840 && H % 4 == 0 && W % OW == 0, "LDS layout");
    // output range of iteration `it` relative to o0, widened to whole chunks of four columns
    static constexpr int lo(int it) { return it == 0 ? -16 : (it == 1 ? -8 : 0); }
    static constexpr int hi(int it) { return it == 0 ? OW + 16 : (it == 1 ? OW + 8 : OW); }
};

// ------------------------------------------------------------------------------------------------------------------
// N(it, b): entry i = image row i.  Slots rotate modulo 3 (entries i, i + 1 in flight while i + 2 is issued); iteration 0 also
// keeps the flow of entries up to i + 4 in flight (the gather's address needs it: it is fetched two entries before the gather).
// ------------------------------------------------------------------------------------------------------------------
template <typename Ge, int IT>
__device__ __forceinline__ void pipe_ne(const float* __restrict__ R, const float* __restrict__ flow, float* __restrict__ mring,
                                        const float* __restrict__ fring, int p, int x, int xr, int lane, bool zf)
{
    constexpr int W = Ge::W, H = W, plane = W * H, TN = Ge::TN, OFF = IT * SKEW;
    const unsigned r0base = (unsigned)p * 5u * plane, r1base = r0base + 5u * plane, flbase = (unsigned)p * 2u * plane;
    const float sxn = border_factor(x, W);
    NeIn in[3];
    NeG2 g[3];
    float fl[6][2];                                        // IT == 0: flow of entries in flight (slot = entry % 6)
    auto flow_issue = [&](int j, int s6) __attribute__((always_inline)) {
        if (IT == 0 && j < H) {
            const unsigned o = (unsigned)(j * W + x);
            fl[s6][0] = ld_off<float>(flow, (flbase + o) * 4u);
            fl[s6][1] = ld_off<float>(flow, (flbase + plane + o) * 4u);
        }
    };
    auto issue = [&](int j, int s3, int s6) __attribute__((always_inline)) {       // R0, flow and the bilinear gather of entry j
        if (j >= H) return;
        ne_load_r0(R, r0base, x, j, W, in[s3]);
        if (IT == 0) { in[s3].dx = fl[s6][0]; in[s3].dy = fl[s6][1]; }
        else { const float* f = fring + (j & 7) * Ge::F_ROW + xr; in[s3].dx = f[0]; in[s3].dy = f[128]; }
        ne_gather2(R, r1base, in[s3], x, j, W, H, g[s3], zf);
    };
    auto finish = [&](int i, int s3) __attribute__((always_inline)) {
        float a[5], mm[5];
        ne_finish_r(in[s3], g[s3], x, i, W, H, a, zf);
        ne_products(a, sxn * border_factor(i, H), mm);
        float* dst = mring + (i & 3) * Ge::M_SLOT + lane;
#pragma unroll
        for (int c = 0; c < 5; c++) dst[c * 64] = mm[c];
    };
    // local step u (q = u % 3 static): entries 2 u, 2 u + 1
    auto step = [&](int u, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = 2 * u + k, s = (2 * q + k) % 6;                      // s = i % 6 (static)
            flow_issue(i + 4, (s + 4) % 6);
            issue(i + 2, (s + 2) % 3, (s + 2) % 6);
            __builtin_amdgcn_sched_barrier(0);
            finish(i, s % 3);
        }
    };
    for (int s = 0; s < OFF; s++) __syncthreads();
    // local step 0 begins with the prologue: entries 0 and 1 are issued (iteration 0: the flow of entries 0 .. 3 first)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) flow_issue(j, j);
    issue(0, 0, 0);
    issue(1, 1, 1);
    step(0, 0);
    int u = 1;
    // steps 1, 2 complete the first period; then whole periods of three steps
    if (TN > 1) { __syncthreads(); step(1, 1); u = 2; }
    if (TN > 2) { __syncthreads(); step(2, 2); u = 3; }
    for (; u + 3 <= TN; u += 3) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            __syncthreads();
            step(u + q, q);
        }
    }
#pragma unroll
    for (int q = 0; q < (TN >= 3 ? (TN - 3) % 3 : 0); q++) {
        __syncthreads();
        step(u + q, q);
    }
    constexpr int DONE = OFF + TN;
    for (int s = DONE; s < Ge::STEPS; s++) __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------------
// C(it, b): cv2's vertical running sums, literally.  Entry e brings image row min(e, H - 1) in; from e = 7 on, row e - 15 (row 0 while
// the window still touches the top edge) leaves and the vsum row of image row e - 7 is published.  Local step u >= 1: entries
// 2 (u - 1), 2 (u - 1) + 1.  The 15-row history lives in a statically indexed register ring: the loop is unrolled over 8 steps.
// ------------------------------------------------------------------------------------------------------------------
template <typename Ge, int IT>
__device__ __forceinline__ void pipe_chain(const float* __restrict__ mring, double* __restrict__ vsring, int b, int lane)
{
    constexpr int H = Ge::H, NE = Ge::NE, OFF = IT * SKEW, TA = Ge::TC - 1;   // TA active steps (u = 1 .. TC - 1)
    float ring[16][5];
    double vs[5] = {0., 0., 0., 0., 0.};
    double* vdst = vsring + 64 * b + lane;
    // active step index v = u - 1 = 0 .. TA - 1, entries 2 v, 2 v + 1; kk0 = (2 v) & 15 static with q = v % 8
    auto step = [&](int v, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int e = 2 * v + k, kk = (2 * q + k) & 15;                    // e & 15, static
            if (e < NE) {                                                      // wave-uniform
                const int er = e < H - 1 ? e : H - 1;
                const float* src = mring + (er & 3) * Ge::M_SLOT + lane;
                float a[5];
#pragma unroll
                for (int c = 0; c < 5; c++) a[c] = src[c * 64];
                if (e == 0) {
#pragma unroll
                    for (int c = 0; c < 5; c++) vs[c] = (double)(a[c] * (float)(kM + 2));
                } else if (e < kM) {
#pragma unroll
                    for (int c = 0; c < 5; c++) vs[c] += (double)a[c];
                } else {
                    const bool top = e < 16;                                   // the leaving row is still row 0
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        const float lv = top ? ring[0][c] : ring[(kk + 1) & 15][c];
                        vs[c] += (double)(a[c] - lv);
                    }
                    double* d = vdst + ((e - kM) & 3) * Ge::VS_SLOT;
#pragma unroll
                    for (int c = 0; c < 5; c++) d[c * Ge::ROWLEN] = vs[c];
                }
#pragma unroll
                for (int c = 0; c < 5; c++) ring[kk][c] = a[c];
            }
        }
    };
    for (int s = 0; s < OFF + 1; s++) __syncthreads();                         // local step 0 has nothing to read yet
    int v = 0;
    for (; v + 8 <= TA; v += 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            __syncthreads();
            step(v + q, q);
        }
    }
#pragma unroll
    for (int q = 0; q < TA % 8; q++) {
        __syncthreads();
        step(v + q, q);
    }
    constexpr int DONE = OFF + 1 + TA;
    for (int s = DONE; s < Ge::STEPS; s++) __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------------
// X(it): window sums of 15 columns, 2 x 2 solve (double, cv2's operation order).  Lane = (row r of the step's two rows, chunk jj of
// four output columns).  The chunk's image columns are o0 + lo + 4 jj .. + 3 (a multiple of four: the grouping of the additions is the
// one avd_fbfast.hip uses); lane column index of image column c is c - o0 + 23, so the window of the chunk's first column starts at
// lane column c0 - o0 + 16 (even: 16-byte LDS reads).
// ------------------------------------------------------------------------------------------------------------------
template <typename Ge, int IT>
__device__ __forceinline__ void pipe_solve(const double* __restrict__ vsring, float* __restrict__ fring_out, float* __restrict__ flow_out,
                                           int* __restrict__ flags, int p, int lane, int o0)
{
    constexpr int W = Ge::W, H = W, plane = W * H, OFF = IT * SKEW, TX = Ge::TX;
    const double scale = 1. / (15 * 15);
    const int r = lane >> 5, jj = lane & 31;
    const int c0 = o0 + Ge::lo(IT) + 4 * jj;                                  // first image column of the chunk
    const bool colok = c0 < o0 + Ge::hi(IT) && c0 >= 0 && c0 < W;
    const int w0 = c0 - o0 + 16;                                              // lane column where the first window starts (>= 0 when colok)
    const double* vsrc = vsring + (colok ? w0 : 0);
    bool ill = false;
    for (int s = 0; s < OFF; s++) __syncthreads();
    for (int u = 0; u < TX; u++) {
        __syncthreads();
        const int y = 2 * u - 11 + r;
        if (colok && y >= 0 && y < H) {
            const dbl2* sp = reinterpret_cast<const dbl2*>(vsrc + (y & 3) * Ge::VS_SLOT);
            double o[5][4];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                double v[18];
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const dbl2 w2 = sp[c * (Ge::ROWLEN / 2) + i];
                    v[2 * i] = w2.x; v[2 * i + 1] = w2.y;
                }
                double A = v[3];
#pragma unroll
                for (int i = 4; i < 15; i++) A += v[i];
                const double p12 = v[1] + v[2], q2 = v[15] + v[16];
                o[c][0] = A + (v[0] + p12);
                o[c][1] = A + (p12 + v[15]);
                o[c][2] = A + (v[2] + q2);
                o[c][3] = A + (q2 + v[17]);
            }
            float fx[4], fy[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double g11 = o[0][i] * scale, g12 = o[1][i] * scale, g22 = o[2][i] * scale;
                const double h1 = o[3][i] * scale, h2 = o[4][i] * scale;
                const double t1 = g11 * g22, t2 = g12 * g12, den = t1 - t2 + 1e-3;
                const double idet = recip_exact(den);
                fx[i] = (float)((g11 * h2 - g12 * h1) * idet);
                fy[i] = (float)((g22 * h1 - g12 * h2) * idet);
                ill |= !(t1 + t2 <= kCondMax * den) | !(fmaxf(fabsf(fx[i]), fabsf(fy[i])) <= kFlowMax * (float)W);
            }
            if (IT < 2) {
                float* d = fring_out + (y & 7) * Ge::F_ROW + (c0 - o0 + Ge::XLO);
                *reinterpret_cast<float4*>(d) = make_float4(fx[0], fx[1], fx[2], fx[3]);
                *reinterpret_cast<float4*>(d + 128) = make_float4(fy[0], fy[1], fy[2], fy[3]);
            } else {
                float* d = flow_out + (size_t)p * 2 * plane + y * W + c0;
                *reinterpret_cast<float4*>(d) = make_float4(fx[0], fx[1], fx[2], fx[3]);
                *reinterpret_cast<float4*>(d + plane) = make_float4(fy[0], fy[1], fy[2], fy[3]);
            }
        }
    }
    constexpr int DONE = OFF + TX;
    for (int s = DONE; s < Ge::STEPS; s++) __syncthreads();
    if (flags && __builtin_amdgcn_ballot_w64(ill) != 0 && lane == 0) atomicOr(flags + p, 1 << (W == 160 ? 1 : W == 80 ? 2 : 3));
}

template <int W>
__global__ __launch_bounds__(960) void k_fb_pipe(const float* __restrict__ R, const float* __restrict__ flow_in, float* __restrict__ flow_out,
                                                int* __restrict__ flags, int npairs, int zero_first)
{
    using Ge = PGeo<W>;
    __shared__ __align__(16) double lds[(Ge::LDS_BYTES + 7) / 8];
    double* vsrings = lds;
    float* mrings = reinterpret_cast<float*>(lds + 3 * Ge::VS_IT);
    float* frings = mrings + 6 * Ge::M_RING;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ppx = (npairs + 7) >> 3;
    const int local = blockIdx.x >> 3;
    const int p = (blockIdx.x & 7) * ppx + local / Ge::NSTRIPS;
    const int s = local % Ge::NSTRIPS;
    if (p >= npairs) return;                              // whole workgroup
    const int o0 = s * Ge::OW;
    // wave -> (iteration, role): five waves per iteration: N block 0, N block 1, C block 0, C block 1, X
    const int it = wave / 5, role = wave - 5 * it;
    const int b = role & 1;
    const int xu = o0 - Ge::XLO + 64 * b + lane;
    const int x = xu < 0 ? 0 : (xu > W - 1 ? W - 1 : xu);
    // lane column (index into the flow ring of the previous iteration) of the image column this lane reads its flow at: the lane's own
    // column clamped to the image AND to what that iteration produced (lanes outside it feed no window of this iteration)
    const int vlo = o0 + Ge::lo(it > 0 ? it - 1 : 0) < 0 ? 0 : o0 + Ge::lo(it > 0 ? it - 1 : 0);
    const int vhi = o0 + Ge::hi(it > 0 ? it - 1 : 0) > W ? W : o0 + Ge::hi(it > 0 ? it - 1 : 0);
    const int xc = x < vlo ? vlo : (x > vhi - 1 ? vhi - 1 : x);
    const int xr = xc - o0 + Ge::XLO;
    float* mring = mrings + (it * 2 + b) * Ge::M_RING;
    double* vsring = vsrings + it * Ge::VS_IT;
    const float* fring_in = frings + (it > 0 ? it - 1 : 0) * Ge::F_RING;
    float* fring_out = frings + (it < 2 ? it : 0) * Ge::F_RING;
    if (role < 2) {
        if (it == 0) pipe_ne<Ge, 0>(R, flow_in, mring, fring_in, p, x, xr, lane, zero_first != 0);
        else if (it == 1) pipe_ne<Ge, 1>(R, flow_in, mring, fring_in, p, x, xr, lane, false);
        else pipe_ne<Ge, 2>(R, flow_in, mring, fring_in, p, x, xr, lane, false);
    } else if (role < 4) {
        __builtin_amdgcn_s_setprio(3);                    // the sequential part
        if (it == 0) pipe_chain<Ge, 0>(mring, vsring, b, lane);
        else if (it == 1) pipe_chain<Ge, 1>(mring, vsring, b, lane);
        else pipe_chain<Ge, 2>(mring, vsring, b, lane);
    } else {
        if (it == 0) pipe_solve<Ge, 0>(vsring, fring_out, flow_out, flags, p, lane, o0);
        else if (it == 1) pipe_solve<Ge, 1>(vsring, fring_out, flow_out, flags, p, lane, o0);
        else pipe_solve<Ge, 2>(vsring, fring_out, flow_out, flags, p, lane, o0);
    }
}

template <int W>
void launch_pipe(hipStream_t stream, const float* R, const float* fin, float* fout, int* flags, int np, int zero_first)
{
    const int grid = 8 * ((np + 7) / 8) * PGeo<W>::NSTRIPS;
    hipLaunchKernelGGL(k_fb_pipe<W>, dim3(grid), dim3(960), 0, stream, R, fin, fout, flags, np, zero_first);
}

}  // namespace

// All three blur iterations of one small pyramid level (w = 160 / 80 / 40) for `np` pairs in one launch: flow_in = the level's initial
// flow ([pair][2][w][w]; ignored when zero_first), flow_out (a different buffer) receives the final one.
int launch_fb_pipe(avd_ctx* ctx, hipStream_t stream, int w, const float* R, const float* flow_in, float* flow_out, int* flags, int np, int zero_first)
{
    if (np <= 0) return 0;
    if (flow_in == flow_out) { ctx->err = "launch_fb_pipe: the flow is not updated in place"; return AVD_ERR_ARG; }
    switch (w) {
    case 160: launch_pipe<160>(stream, R, flow_in, flow_out, flags, np, zero_first); break;
    case 80: launch_pipe<80>(stream, R, flow_in, flow_out, flags, np, zero_first); break;
    case 40: launch_pipe<40>(stream, R, flow_in, flow_out, flags, np, zero_first); break;
    default: ctx->err = "launch_fb_pipe: unsupported level size"; return AVD_ERR_ARG;
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
