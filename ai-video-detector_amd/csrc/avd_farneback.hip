// avd_farneback.hip -- dense Farneback optical flow for batches of 320x320 frame pairs (gfx950).
//
// Replaces cv2.calcOpticalFlowFarneback(prev, cur, None, 0.5, 3, 15, 3, 5, 1.2, 0) and the
// flow-magnitude statistics of reference app/analyzers/video.py:45-48 for all consecutive
// pairs of a clip at once.  The arithmetic follows OpenCV 4.10's CPU path operation by
// operation (float where it is float, double accumulators where it keeps doubles, fused
// multiply-add only in the Gaussian row/column filters), so results are bit-identical to
// oracle/avd_oracle.c; what is re-designed is the schedule:
//   * per-FRAME work (Gaussian pyramid, polynomial expansion) is done once per frame and
//     shared by the two pairs a frame belongs to (cv2 recomputes it per pair);
//   * all pairs of a clip advance through level/iteration in lock step, one launch per
//     stage, so every launch has >= 10^5 independent work items;
//   * planar (structure-of-arrays) R / M / flow planes so that neighbouring lanes touch
//     neighbouring addresses (cv2 interleaves 5 channels per pixel).
// Not GEMM-shaped (11/19-tap separable stencils, 15x15 box sums, per-pixel 2x2 solves):
// VALU + cache bound, MFMA is not applicable (DESIGN.md "Farneback").
#include "avd_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int S = AVD_SMALL;

__device__ __forceinline__ int reflect101(int p, int len)
{
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int floor_f(float v) { int i = (int)v; return i - (i > v); }

// ---------------------------------------------------------------------------------------
// Gaussian pyramid level K (scale 2^-K): GaussianBlur(full-res, ksize, sigma) then the
// INTER_LINEAR decimation, which at these exact power-of-two scales is the 2x2 mean
// ((p00+p01)+(p10+p11))*0.25 of the two centre pixels.  Only the columns/rows the
// decimation reads are filtered.
//   row pass  : tmp[f][y][j]  j = 2*dx+{0,1} <-> source column (dx<<K)+off+{0,1}
//   col pass  : I[f][dy][dx]
// ---------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_pyr_row(const uint8_t* __restrict__ small, int n,
                                                const FbConsts* __restrict__ C, float* __restrict__ tmp)
{
    constexpr int WL = S >> K;
    constexpr int NC = K == 0 ? S : 2 * WL;          // filtered columns per row
    constexpr int OFF = K == 0 ? 0 : (1 << K) / 2 - 1;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)n * S * NC) return;
    const int j = (int)(gid % NC);
    const int y = (int)((gid / NC) % S);
    const int f = (int)(gid / ((int64_t)NC * S));
    const int x = K == 0 ? j : ((j >> 1) << K) + OFF + (j & 1);
    const uint8_t* row = small + (int64_t)f * AVD_NPIX + y * S;
    const int ks = C->gksize[K];
    const float* kx = C->gk[K];
    float r;
    if (ks == 3) {
        const float l = (float)row[reflect101(x - 1, S)], c = (float)row[x], rr = (float)row[reflect101(x + 1, S)];
        const float t = (l + rr) * kx[0];
        r = __builtin_fmaf(c, kx[1], t);
    } else {
        const int half = ks >> 1;
        r = 0.f;
        for (int t = 0; t < ks; t++) r = __builtin_fmaf((float)row[reflect101(x - half + t, S)], kx[t], r);
    }
    tmp[gid] = r;
}

template <int K>
__global__ __launch_bounds__(256) void k_pyr_col(const float* __restrict__ tmp, int n,
                                                const FbConsts* __restrict__ C, float* __restrict__ I)
{
    constexpr int WL = S >> K;
    constexpr int NC = K == 0 ? S : 2 * WL;
    constexpr int OFF = K == 0 ? 0 : (1 << K) / 2 - 1;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)n * WL * WL) return;
    const int dx = (int)(gid % WL);
    const int dy = (int)((gid / WL) % WL);
    const int f = (int)(gid / (WL * WL));
    const float* T = tmp + (int64_t)f * S * NC;
    const int ks = C->gksize[K], half = ks >> 1;
    const float* kc = C->gk[K] + half;
    auto colf = [&](int y, int j) {
        float s = __builtin_fmaf(T[y * NC + j], kc[0], 0.f);
        for (int t = 1; t <= half; t++) {
            const float a = T[reflect101(y + t, S) * NC + j], b = T[reflect101(y - t, S) * NC + j];
            s = __builtin_fmaf(a + b, kc[t], s);
        }
        return s;
    };
    float out;
    if (K == 0) {
        out = colf(dy, dx);
    } else {
        const int y0 = (dy << K) + OFF;
        const float p00 = colf(y0, 2 * dx), p01 = colf(y0, 2 * dx + 1);
        const float p10 = colf(y0 + 1, 2 * dx), p11 = colf(y0 + 1, 2 * dx + 1);
        out = ((p00 + p01) + (p10 + p11)) * 0.25f;
    }
    I[gid] = out;
}

// ---------------------------------------------------------------------------------------
// FarnebackPolyExp (poly_n = 5): one workgroup per image row.  Vertical 11-tap pass in
// float into LDS (3 moment planes, replicate border), horizontal pass with double
// accumulators, output 5 planar coefficient planes R[c][y][x].
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(320) void k_polyexp(const float* __restrict__ I, int w, int h,
                                                const FbConsts* __restrict__ C, float* __restrict__ R)
{
    __shared__ float row[3][S + 10];
    const int y = blockIdx.x % h, f = blockIdx.x / h;
    const int x = threadIdx.x;
    const float* img = I + (int64_t)f * w * h;
    const float* g = C->g + 5; const float* xg = C->xg + 5; const float* xxg = C->xxg + 5;
    if (x < w) {
        float t0 = img[y * w + x] * g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 1; k <= 5; k++) {
            const float a = img[max(y - k, 0) * w + x], b = img[min(y + k, h - 1) * w + x];
            const float p = a + b;
            t0 = t0 + g[k] * p;
            t1 = t1 + xg[k] * (b - a);
            t2 = t2 + xxg[k] * p;
        }
        row[0][x + 5] = t0; row[1][x + 5] = t1; row[2][x + 5] = t2;
        if (x == 0)
            for (int k = 0; k < 5; k++) { row[0][k] = t0; row[1][k] = t1; row[2][k] = t2; }
        if (x == w - 1)
            for (int k = 0; k < 5; k++) { row[0][w + 5 + k] = t0; row[1][w + 5 + k] = t1; row[2][w + 5 + k] = t2; }
    }
    __syncthreads();
    if (x >= w) return;
    const float* r0 = row[0] + x + 5; const float* r1 = row[1] + x + 5; const float* r2 = row[2] + x + 5;
    double b1 = (double)(r0[0] * g[0]), b2 = 0, b3 = (double)(r1[0] * g[0]), b4 = 0,
           b5 = (double)(r2[0] * g[0]), b6 = 0;
#pragma unroll
    for (int k = 1; k <= 5; k++) {
        const double tg = (double)(r0[k] + r0[-k]);
        b1 += tg * (double)g[k];
        b4 += tg * (double)xxg[k];
        b2 += (double)((r0[k] - r0[-k]) * xg[k]);
        b3 += (double)((r1[k] + r1[-k]) * g[k]);
        b6 += (double)((r1[k] - r1[-k]) * xg[k]);
        b5 += (double)((r2[k] + r2[-k]) * g[k]);
    }
    const int64_t plane = (int64_t)w * h;
    float* out = R + (int64_t)f * 5 * plane + y * w + x;
    out[0] = (float)(b3 * C->ig11);
    out[plane] = (float)(b2 * C->ig11);
    out[2 * plane] = (float)(b1 * C->ig03 + b5 * C->ig33);
    out[3 * plane] = (float)(b1 * C->ig03 + b4 * C->ig33);
    out[4 * plane] = (float)(b6 * C->ig55);
}

// ---------------------------------------------------------------------------------------
// Initial flow of a level: zeros at the coarsest level, otherwise the previous level's
// flow resized x2 (INTER_LINEAR, float weights, cv2 edge rules) and multiplied by 2.
// flow planes: [pair][2][h][w]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_up(const float* __restrict__ prev, int pw, int ph,
                                                float* __restrict__ flow, int w, int h, int npairs)
{
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)npairs * 2 * w * h) return;
    const int dx = (int)(gid % w);
    const int dy = (int)((gid / w) % h);
    const int64_t pc = gid / ((int64_t)w * h);             // pair*2 + channel
    const float* src = prev + pc * pw * ph;
    const double scale_x = 1. / ((double)w / pw), scale_y = 1. / ((double)h / ph);
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = floor_f(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    bool edge = false;                                      // dx >= xmax: value copied, no weights
    if (sx + 1 >= pw) { edge = true; if (sx >= pw - 1) { fx = 0; sx = pw - 1; } }
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = floor_f(fy);
    fy -= sy;
    const int y0 = clampi(sy, 0, ph - 1), y1 = clampi(sy + 1, 0, ph - 1);
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const int x1 = min(sx + 1, pw - 1);
    float d0, d1;
    if (edge) { d0 = src[y0 * pw + sx] * 1.f; d1 = src[y1 * pw + sx] * 1.f; }
    else {
        d0 = src[y0 * pw + sx] * a0 + src[y0 * pw + x1] * a1;
        d1 = src[y1 * pw + sx] * a0 + src[y1 * pw + x1] * a1;
    }
    flow[gid] = (d0 * b0 + d1 * b1) * 2.f;
}

// ---------------------------------------------------------------------------------------
// FarnebackUpdateMatrices: per pixel, warp R1 by the current flow (bilinear), build the
// 2x2 normal equations G, h.  R planes [frame][5][h][w]; M planes [pair][5][h][w].
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_matrices(const float* __restrict__ R, const float* __restrict__ flow,
                                                        float* __restrict__ M, int w, int h, int npairs)
{
    const int64_t plane = (int64_t)w * h;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npairs * plane) return;
    const int x = (int)(gid % w);
    const int y = (int)((gid / w) % h);
    const int p = (int)(gid / plane);
    const float* R0 = R + (int64_t)p * 5 * plane + y * w + x;
    const float* R1 = R + (int64_t)(p + 1) * 5 * plane;
    const float* fl = flow + (int64_t)p * 2 * plane + y * w + x;
    const float dx = fl[0], dy = fl[plane];
    float fx = x + dx, fy = y + dy;
    const int x1 = floor_f(fx), y1 = floor_f(fy);
    float r2, r3, r4, r5, r6;
    fx -= x1; fy -= y1;
    if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        const float* q = R1 + y1 * w + x1;
        r2 = a00 * q[0] + a01 * q[1] + a10 * q[w] + a11 * q[w + 1]; q += plane;
        r3 = a00 * q[0] + a01 * q[1] + a10 * q[w] + a11 * q[w + 1]; q += plane;
        r4 = a00 * q[0] + a01 * q[1] + a10 * q[w] + a11 * q[w + 1]; q += plane;
        r5 = a00 * q[0] + a01 * q[1] + a10 * q[w] + a11 * q[w + 1]; q += plane;
        r6 = a00 * q[0] + a01 * q[1] + a10 * q[w] + a11 * q[w + 1];
        r4 = (R0[2 * plane] + r4) * 0.5f;
        r5 = (R0[3 * plane] + r5) * 0.5f;
        r6 = (R0[4 * plane] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = R0[2 * plane];
        r5 = R0[3 * plane];
        r6 = R0[4 * plane] * 0.5f;
    }
    r2 = (R0[0] - r2) * 0.5f;
    r3 = (R0[plane] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if ((unsigned)(x - 5) >= (unsigned)(w - 10) || (unsigned)(y - 5) >= (unsigned)(h - 10)) {
        auto border = [](int d) { return d < 2 ? 0.14f : 0.4472f; };      // {.14,.14,.4472,.4472,.4472}
        const float scale = (x < 5 ? border(x) : 1.f) * (x >= w - 5 ? border(w - x - 1) : 1.f) *
                            (y < 5 ? border(y) : 1.f) * (y >= h - 5 ? border(h - y - 1) : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    float* out = M + (int64_t)p * 5 * plane + y * w + x;
    out[0] = r4 * r4 + r6 * r6;
    out[plane] = (r4 + r5) * r6;
    out[2 * plane] = r5 * r5 + r6 * r6;
    out[3 * plane] = r4 * r2 + r6 * r3;
    out[4 * plane] = r6 * r2 + r5 * r3;
}

// ---------------------------------------------------------------------------------------
// FarnebackUpdateFlow_Blur, winsize 15 (m = 7).  cv2 keeps RUNNING box sums in double and
// rounds at every slide, so the value at (y,x) depends on the whole column / row prefix:
// the chains are reproduced literally, one lane per chain.
//   k_blur_v : lane = (pair, channel, x), walks y;  VS[pair][c][y][x] (double)
//   k_blur_h : lane = (pair, y), walks x with 5 running sums, solves the 2x2 system.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blur_v(const float* __restrict__ M, double* __restrict__ VS,
                                               int w, int h, int npairs)
{
    constexpr int m = 7;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)npairs * 5 * w) return;
    const int x = (int)(gid % w);
    const int64_t pc = gid / w;
    const float* src = M + pc * w * h + x;
    double* dst = VS + pc * w * h + x;
    double vs = (double)(src[0] * (float)(m + 2));
    for (int y = 1; y < m; y++) vs += (double)src[min(y, h - 1) * w];
    for (int y = 0; y < h; y++) {
        const float a = src[min(y + m, h - 1) * w], b = src[max(y - m - 1, 0) * w];
        vs += (double)(a - b);
        dst[y * w] = vs;
    }
}

__global__ __launch_bounds__(64) void k_blur_h(const double* __restrict__ VS, float* __restrict__ flow,
                                              int w, int h, int npairs)
{
    constexpr int m = 7;
    const int gid = blockIdx.x * 64 + threadIdx.x;
    if (gid >= npairs * h) return;
    const int y = gid % h, p = gid / h;
    const int64_t plane = (int64_t)w * h;
    const double* v = VS + (int64_t)p * 5 * plane + y * w;
    float* fl = flow + (int64_t)p * 2 * plane + y * w;
    const double scale = 1. / (15 * 15);
    double g[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const double* vc = v + c * plane;
        double s = vc[0] * (double)(m + 2);
        for (int x = 1; x < m; x++) s += vc[min(x, w - 1)];
        g[c] = s;
    }
    for (int x = 0; x < w; x++) {
        const int xa = min(x + m, w - 1), xb = max(x - m - 1, 0);
#pragma unroll
        for (int c = 0; c < 5; c++) g[c] += v[c * plane + xa] - v[c * plane + xb];
        const double g11 = g[0] * scale, g12 = g[1] * scale, g22 = g[2] * scale;
        const double h1 = g[3] * scale, h2 = g[4] * scale;
        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
        fl[x] = (float)((g11 * h2 - g12 * h1) * idet);
        fl[plane + x] = (float)((g22 * h1 - g12 * h2) * idet);
    }
}

// ---------------------------------------------------------------------------------------
// Flow statistics in numpy's float32 order (video.py:46-48): mag = sqrt(fx*fx + fy*fy);
// add.reduce = pairwise sums (128-element leaves, 8 strided accumulators) inside 8192-element
// iterator buffers whose results are added sequentially.  One workgroup per pair.
// ---------------------------------------------------------------------------------------
constexpr int kChunk = 8192;

__global__ __launch_bounds__(1024) void k_flow_stats(const float* __restrict__ flow, float* __restrict__ stats,
                                                    float* __restrict__ flow_il)
{
    __shared__ float buf[kChunk];
    __shared__ float leaf[kChunk / 16];              // 64 leaves x 8 accumulators
    __shared__ float node[64];
    __shared__ float bc;
    const int p = blockIdx.x, tid = threadIdx.x;
    const float* fxp = flow + (int64_t)p * 2 * AVD_NPIX;
    const float* fyp = fxp + AVD_NPIX;
    float total = 0.f, mean32 = 0.f;
    for (int pass = 0; pass < 2; pass++) {
        total = 0.f;
        for (int base = 0; base < AVD_NPIX; base += kChunk) {
            const int len = min(kChunk, AVD_NPIX - base);
            for (int i = tid; i < len; i += 1024) {
                const float fx = fxp[base + i], fy = fyp[base + i];
                const float a = fx * fx, b = fy * fy;
                float mg = sqrtf(a + b);
                if (pass == 0 && flow_il) {
                    flow_il[((int64_t)p * AVD_NPIX + base + i) * 2] = fx;
                    flow_il[((int64_t)p * AVD_NPIX + base + i) * 2 + 1] = fy;
                }
                if (pass == 1) { const float d = mg - mean32; mg = d * d; }
                buf[i] = mg;
            }
            __syncthreads();
            const int nleaf = len >> 7;
            for (int it = tid; it < nleaf * 8; it += 1024) {
                const float* q = buf + (it >> 3) * 128 + (it & 7);
                float r = q[0];
#pragma unroll
                for (int k = 1; k < 16; k++) r += q[8 * k];
                leaf[it] = r;
            }
            __syncthreads();
            if (tid < nleaf) {
                const float* r = leaf + tid * 8;
                node[tid] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            }
            __syncthreads();
            // balanced pairwise tree over the leaves (nleaf is a power of two: 64 or 32)
            for (int stride = 1; stride < nleaf; stride <<= 1) {
                if (tid < nleaf && (tid % (2 * stride)) == 0) node[tid] = node[tid] + node[tid + stride];
                __syncthreads();
            }
            if (tid == 0) total = base == 0 ? node[0] : total + node[0];
            __syncthreads();
        }
        if (pass == 0) {
            if (tid == 0) {
                bc = total / (float)AVD_NPIX;                       // _var: float32 array true_divide
                stats[2 * p] = (float)((double)total / (double)AVD_NPIX);   // _mean
            }
            __syncthreads();
            mean32 = bc;
        } else if (tid == 0) {
            stats[2 * p + 1] = (float)((double)total / (double)AVD_NPIX);
        }
    }
}

template <typename... A>
inline void launch1d(void (*k)(A...), int64_t items, int block, hipStream_t s, A... args)
{
    const int grid = (int)((items + block - 1) / block);
    if (grid > 0) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, s, args...);
}

template <int K>
void pyramid_level(avd_ctx* ctx, const uint8_t* d_small, int n)
{
    Workspace& ws = ctx->ws;
    constexpr int WL = S >> K;
    constexpr int NC = K == 0 ? S : 2 * WL;
    const FbConsts* C = (const FbConsts*)ctx->d_fbc;
    launch1d(k_pyr_row<K>, (int64_t)n * S * NC, 256, ctx->stream, d_small, n, C, ws.d_tmp);
    launch1d(k_pyr_col<K>, (int64_t)n * WL * WL, 256, ctx->stream, (const float*)ws.d_tmp, n, C, ws.d_pyr[K]);
    hipLaunchKernelGGL(k_polyexp, dim3(n * WL), dim3(320), 0, ctx->stream, (const float*)ws.d_pyr[K], WL, WL, C,
                       ws.d_poly[K]);
}

}  // namespace

// All pairs (f, f+1), f in [0, n-1), of n resident 320x320 frames.
int launch_farneback(avd_ctx* ctx, const uint8_t* d_small, int n)
{
    if (n < 2) return 0;
    Workspace& ws = ctx->ws;
    const int np = n - 1;
    pyramid_level<3>(ctx, d_small, n);
    pyramid_level<2>(ctx, d_small, n);
    pyramid_level<1>(ctx, d_small, n);
    pyramid_level<0>(ctx, d_small, n);
    for (int k = AVD_FB_LEVELS - 1; k >= 0; k--) {
        const int w = S >> k, h = S >> k;
        const int64_t plane = (int64_t)w * h;
        if (k == AVD_FB_LEVELS - 1)
            HIP_TRY(ctx, hipMemsetAsync(ws.d_flow[k], 0, sizeof(float) * 2 * plane * np, ctx->stream));
        else
            launch1d(k_flow_up, (int64_t)np * 2 * plane, 256, ctx->stream, (const float*)ws.d_flow[k + 1],
                     w / 2, h / 2, ws.d_flow[k], w, h, np);
        launch1d(k_update_matrices, np * plane, 256, ctx->stream, (const float*)ws.d_poly[k],
                 (const float*)ws.d_flow[k], ws.d_M[0], w, h, np);
        for (int it = 0; it < 3; it++) {
            launch1d(k_blur_v, (int64_t)np * 5 * w, 256, ctx->stream, (const float*)ws.d_M[0], ws.d_vs, w, h, np);
            launch1d(k_blur_h, (int64_t)np * h, 64, ctx->stream, (const double*)ws.d_vs, ws.d_flow[k], w, h, np);
            if (it < 2)
                launch1d(k_update_matrices, np * plane, 256, ctx->stream, (const float*)ws.d_poly[k],
                         (const float*)ws.d_flow[k], ws.d_M[0], w, h, np);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int launch_flow_stats(avd_ctx* ctx, int n)
{
    if (n < 2) return 0;
    Workspace& ws = ctx->ws;
    hipLaunchKernelGGL(k_flow_stats, dim3(n - 1), dim3(1024), 0, ctx->stream, (const float*)ws.d_flow[0],
                       ws.d_stats, ws.d_flow_il);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
