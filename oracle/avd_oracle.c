/*
 * avd_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See avd_oracle.h / oracle/README.md.  PARITY UNPINNED at the OpenCV boundary.
 *
 * Build: gcc -std=gnu11 -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * Every float expression below is written in the evaluation order of the
 * OpenCV 4.10 x86-64 code path it restates; fused multiply-adds appear only
 * where that path uses them (explicit fmaf), never by compiler contraction.
 *
 * Restated third-party algorithms (sources absent from /root/reference):
 *   OpenCV 4.10  modules/imgproc/src/color_rgb.simd.hpp   (RGB2Gray<uchar>)
 *                modules/imgproc/src/resize.cpp           (INTER_AREA, INTER_LINEAR)
 *                modules/imgproc/src/deriv.cpp            (Laplacian ksize=1)
 *                modules/imgproc/src/smooth.dispatch.cpp  (GaussianBlur, kernels)
 *                modules/imgproc/src/filter.simd.hpp      (row/column filters, AVX2+FMA3 dispatch)
 *                modules/video/src/optflowgf.cpp          (Farneback CPU path)
 *   numpy 1.26   core/src/umath/loops_utils.h.src         (pairwise_sum), core/_methods.py (_mean,_var)
 * Call sites pinned by the reference: app/analyzers/video.py:4-8,36-57.
 */
#include "avd_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------- OpenCV scalar helpers (core/fast_math.hpp, saturate.hpp) ---------- */
static inline int cv_round_d(double v) { return (int)lrint(v); }   /* round-half-even */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil_d(double v) { int i = (int)v; return i + (i < v); }
static inline short sat_short_f(float v) {
    int i = cv_round_f(v);
    return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}
static inline uint8_t sat_u8_f(float v) {
    int i = cv_round_f(v);
    return (uint8_t)(i < 0 ? 0 : i > 255 ? 255 : i);
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int clip_i(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = len - 1 - (p - len) - 1;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* ---------- operation counter (bench.py's roofline; SURVEY.md 8d asks for an exact count, not an estimate) ----------
 * Every float / double add, subtract, multiply, divide and square root of the Farneback chain and of the flow statistics
 * is tallied where it executes (an fma counts as 2; conversions, comparisons, floor and index arithmetic are not counted).
 * Not thread-safe: one analysing thread per process, which is how this library is used. */
static uint64_t g_ops[2];                     /* [0] float operations, [1] double operations */
void avdo_ops_reset(void) { g_ops[0] = g_ops[1] = 0; }
void avdo_ops_get(uint64_t out[2]) { out[0] = g_ops[0]; out[1] = g_ops[1]; }
#define OPS32(n) (g_ops[0] += (uint64_t)(n))
#define OPS64(n) (g_ops[1] += (uint64_t)(n))

/* ---------- cvtColor BGR2GRAY, uint8 (video.py:5,43,51) ----------
 * RGB2Gray<uchar>: 15-bit fixed point, BY=3735 GY=19235 RY=9798, CV_DESCALE. */
void avdo_bgr2gray(const uint8_t* bgr, int h, int w, int64_t row_stride, uint8_t* gray)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = bgr + (int64_t)y * row_stride;
        uint8_t* d = gray + (int64_t)y * w;
        for (int x = 0; x < w; x++, s += 3)
            d[x] = (uint8_t)((s[0] * 3735 + s[1] * 19235 + s[2] * 9798 + (1 << 14)) >> 15);
    }
}

/* ---------- resize INTER_AREA, uint8 (video.py:6) ----------
 * computeResizeAreaTab + ResizeArea_Invoker<uchar,float> (general path), or
 * ResizeAreaFast_Invoker when both scales are integers. */
typedef struct { int si, di; float alpha; } DecimateAlpha;

static int area_tab(int ssize, int dsize, double scale, DecimateAlpha* tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cellWidth = fmin(scale, ssize - fsx1);
        int sx1 = cv_ceil_d(fsx1), sx2 = cv_floor_d(fsx2);
        sx2 = imin(sx2, ssize - 1);
        sx1 = imin(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx; tab[k].si = sx1 - 1;
            tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx; tab[k].si = sx;
            tab[k++].alpha = (float)(1.0 / cellWidth);
        }
        if (fsx2 - sx2 > 1e-3) {
            tab[k].di = dx; tab[k].si = sx2;
            tab[k++].alpha = (float)(fmin(fmin(fsx2 - sx2, 1.), cellWidth) / cellWidth);
        }
    }
    return k;
}

int avdo_resize_area_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw)
{
    if (dh == sh && dw == sw) { memcpy(dst, src, (size_t)sh * sw); return 0; }
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    if (!(scale_x >= 1 && scale_y >= 1)) return -1;   /* upscaling: not on the reference path */
    int iscale_x = cv_round_d(scale_x), iscale_y = cv_round_d(scale_y);
    int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (is_area_fast) {
        /* ResizeAreaFast_Invoker<uchar,int>: integer box sum, D = saturate_cast<uchar>(sum*scale)
         * with float scale = 1.f/area (round-half-even).  The 2x2 case runs
         * ResizeAreaFastVec_SIMD_8u first, (sum+2)>>2 on whole 8-pixel groups. */
        int area = iscale_x * iscale_y;
        float scale = 1.f / area;
        int simd_w = (iscale_x == 2 && iscale_y == 2) ? (dw & ~7) : 0;
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++) {
                int sum = 0;
                for (int yy = 0; yy < iscale_y; yy++)
                    for (int xx = 0; xx < iscale_x; xx++)
                        sum += src[(int64_t)(dy * iscale_y + yy) * sw + dx * iscale_x + xx];
                dst[dy * dw + dx] = dx < simd_w ? (uint8_t)((sum + 2) >> 2) : sat_u8_f(sum * scale);
            }
        return 0;
    }
    DecimateAlpha* xtab = (DecimateAlpha*)malloc(sizeof(DecimateAlpha) * (size_t)(sw + sh) * 2);
    DecimateAlpha* ytab = xtab + sw * 2;
    int xtab_size = area_tab(sw, dw, scale_x, xtab);
    int ytab_size = area_tab(sh, dh, scale_y, ytab);
    float* buf = (float*)malloc(sizeof(float) * (size_t)dw * 2);
    float* sum = buf + dw;
    for (int dx = 0; dx < dw; dx++) sum[dx] = 0.f;
    int prev_dy = ytab[0].di;
    for (int j = 0; j < ytab_size; j++) {
        float beta = ytab[j].alpha;
        int dy = ytab[j].di, sy = ytab[j].si;
        const uint8_t* S = src + (int64_t)sy * sw;
        for (int dx = 0; dx < dw; dx++) buf[dx] = 0.f;
        for (int k = 0; k < xtab_size; k++) {
            int dxn = xtab[k].di;
            float alpha = xtab[k].alpha;
            buf[dxn] += S[xtab[k].si] * alpha;
        }
        if (dy != prev_dy) {
            uint8_t* D = dst + (int64_t)prev_dy * dw;
            for (int dx = 0; dx < dw; dx++) {
                D[dx] = sat_u8_f(sum[dx]);
                sum[dx] = beta * buf[dx];
            }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < dw; dx++) sum[dx] += beta * buf[dx];
        }
    }
    {
        uint8_t* D = dst + (int64_t)prev_dy * dw;
        for (int dx = 0; dx < dw; dx++) D[dx] = sat_u8_f(sum[dx]);
    }
    free(buf); free(xtab);
    return 0;
}

/* ---------- resize INTER_LINEAR, uint8 (video.py:43) ----------
 * resizeGeneric_<HResizeLinear<uchar,int,short,2048>, VResizeLinear<uchar,int,short,FixedPtCast<22>>> */
int avdo_resize_linear_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw)
{
    if (dh == sh && dw == sw) { memcpy(dst, src, (size_t)sh * sw); return 0; }
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    {   /* INTER_LINEAR at an exact 2x2 decimation is rerouted to INTER_AREA (fast) */
        int isx = cv_round_d(scale_x), isy = cv_round_d(scale_y);
        if (fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON && isx == 2 && isy == 2)
            return avdo_resize_area_u8(src, sh, sw, dst, dh, dw);
    }
    int* xofs = (int*)malloc(sizeof(int) * (size_t)(dw + dh));
    int* yofs = xofs + dw;
    short* ialpha = (short*)malloc(sizeof(short) * (size_t)(dw + dh) * 2);
    short* ibeta = ialpha + dw * 2;
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = imin(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        ialpha[dx * 2] = sat_short_f(c0 * 2048);
        ialpha[dx * 2 + 1] = sat_short_f(c1 * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        yofs[dy] = sy;
        float c0 = 1.f - fy, c1 = fy;
        ibeta[dy * 2] = sat_short_f(c0 * 2048);
        ibeta[dy * 2 + 1] = sat_short_f(c1 * 2048);
    }
    int* rows = (int*)malloc(sizeof(int) * (size_t)dw * 2);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy];
        for (int k = 0; k < 2; k++) {
            int sy = clip_i(sy0 + k, 0, sh);
            const uint8_t* S = src + (int64_t)sy * sw;
            int* D = rows + k * dw;
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx];
                if (dx < xmax) D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
                else D[dx] = S[sx] * 2048;
            }
        }
        int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        const int *S0 = rows, *S1 = rows + dw;
        uint8_t* D = dst + (int64_t)dy * dw;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(rows); free(ialpha); free(xofs);
    return 0;
}

/* ---------- aHash bits (video.py:7-8): mean in float64, g >= mean ---------- */
void avdo_hash_bits(const uint8_t* area, int n, uint8_t* bits)
{
    int64_t s = 0;
    for (int i = 0; i < n; i++) s += area[i];
    double mean = (double)s / (double)n;     /* np.uint8 array .mean() -> float64 */
    for (int i = 0; i < n; i++) bits[i] = (uint8_t)((double)area[i] >= mean);
}

/* ---------- Laplacian ksize=1 (video.py:52) ----------
 * filter2D with [[0,1,0],[1,-4,1],[0,1,0]], BORDER_REFLECT_101, uint8 -> float64 (exact ints) */
static inline int lap_at(const uint8_t* g, int h, int w, int y, int x)
{
    int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
    int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    return (int)g[(int64_t)ym * w + x] + g[(int64_t)yp * w + x] + g[(int64_t)y * w + xm] +
           g[(int64_t)y * w + xp] - 4 * (int)g[(int64_t)y * w + x];
}
void avdo_laplacian_f64(const uint8_t* gray, int h, int w, double* out)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[(int64_t)y * w + x] = (double)lap_at(gray, h, w, y, x);
}
void avdo_laplacian_sums(const uint8_t* gray, int h, int w, int64_t* sum, int64_t* sumsq)
{
    int64_t s = 0, q = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) { int v = lap_at(gray, h, w, y, x); s += v; q += (int64_t)v * v; }
    *sum = s; *sumsq = q;
}

/* ---------- numpy float32 add.reduce over a contiguous array ----------
 * pairwise_sum (loops_utils.h.src, PW_BLOCKSIZE=128, 8 accumulators) applied per
 * iterator buffer of 8192 elements (ufunc buffersize default), chunk results added
 * sequentially -- verified bit-for-bit against numpy in tests/test_oracle_numpy.py. */
static float np_pairwise_f32(const float* a, int64_t n)
{
    if (n < 8) {
        float res = 0.f;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_f32(a, n2) + np_pairwise_f32(a + n2, n - n2);
    }
}
float avdo_np_sum_f32(const float* a, int64_t n)
{
    const int64_t B = 8192;
    if (n <= B) return np_pairwise_f32(a, n);
    float res = np_pairwise_f32(a, B);
    for (int64_t i = B; i < n; i += B) res += np_pairwise_f32(a + i, n - i < B ? n - i : B);
    return res;
}

/* video.py:46-48.  numpy 1.26 semantics (reference pin numpy>=1.26,<2.0):
 *  _mean: ret = umr_sum(f32) ; ret = f32(ret / n)        (scalar/int -> float64 divide, cast back)
 *  _var : arrmean = umr_sum(f32, keepdims) ; arrmean /= n (float32 array true_divide)
 *         x = arr - arrmean ; x = x*x ; ret = umr_sum(x) ; ret = f32(ret / n) */
void avdo_flow_stats(const float* flow, int64_t npix, float* mean, float* var, float* mag_out)
{
    float* mag = mag_out ? mag_out : (float*)malloc(sizeof(float) * (size_t)npix);
    float* dev = (float*)malloc(sizeof(float) * (size_t)npix);
    for (int64_t i = 0; i < npix; i++) {
        float fx = flow[2 * i], fy = flow[2 * i + 1];
        float a = fx * fx, b = fy * fy;
        mag[i] = sqrtf(a + b);
    }
    OPS32(npix * 8 + 4);      /* magnitude (2 mul, add, sqrt), its sum, deviation (sub, mul), its sum; the two final divides are double */
    float s = avdo_np_sum_f32(mag, npix);
    *mean = (float)((double)s / (double)npix);
    float arrmean = s / (float)npix;
    for (int64_t i = 0; i < npix; i++) { float d = mag[i] - arrmean; dev[i] = d * d; }
    float q = avdo_np_sum_f32(dev, npix);
    *var = (float)((double)q / (double)npix);
    free(dev);
    if (!mag_out) free(mag);
}

/* ---------- model switches (sensitivity analysis only; default 0 = the model of README.md) ----------
 * The OpenCV build the reference runs on is not available here, so a few implementation choices of the wheel are
 * modelled, not observed (oracle/README.md "open questions").  tests/test_oracle_sensitivity.py flips each choice
 * and bounds how far it moves flow_mean / ai_susp against the 1e-4 parity tolerance. */
static int g_model = 0;
void avdo_set_model(int flags) { g_model = flags; }
int avdo_get_model(void) { return g_model; }
#define MODEL(bit) ((g_model & (bit)) != 0)

static inline float mac(float a, float b, float c)     /* a*b + c as the Gaussian filters do it */
{
    if (MODEL(AVDO_MODEL_GAUSS_MULADD)) { float p = a * b; return p + c; }   /* -ffp-contract=off: two roundings */
    return fmaf(a, b, c);
}

/* deterministic +-1 ulp pattern (what a different but equally valid operation order would produce) */
static void ulp_jitter(float* a, size_t n, uint32_t seed)
{
    uint32_t st = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; i++) {
        st = st * 1664525u + 1013904223u;
        uint32_t r = st >> 30;                        /* 0,1: keep   2: +1 ulp   3: -1 ulp */
        if (r == 2) a[i] = nextafterf(a[i], INFINITY);
        else if (r == 3) a[i] = nextafterf(a[i], -INFINITY);
    }
}

/* ---------- GaussianBlur on CV_32F (smooth.dispatch.cpp + filter.simd.hpp) ---------- */
/* getGaussianKernel(n, sigma, CV_32F): bit-exact softdouble kernel then cast to float.
 * Restated in IEEE double with libm exp (softfloat's exp may differ in the last double
 * ulp, invisible after the float cast except at measure-zero ties). */
int avdo_gaussian_kernel_f32(int n, double sigma, float* k)
{
    if (n <= 0 || (n & 1) == 0) return -1;
    if (sigma <= 0) {
        static const double k1[] = {1.};
        static const double k3[] = {0.25, 0.5, 0.25};
        static const double k5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
        static const double k7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
        const double* t = n == 1 ? k1 : n == 3 ? k3 : n == 5 ? k5 : n == 7 ? k7 : 0;
        if (t) { for (int i = 0; i < n; i++) k[i] = (float)t[i]; return 0; }
    }
    double sigmaX = sigma > 0 ? sigma : fma((double)n, 0.15, 0.35);
    double scale2X = -0.125 / (sigmaX * sigmaX);
    int n2 = (n - 1) / 2;
    double values[64];
    if (n2 + 1 > 64) return -1;
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        double t = exp((double)(x * x) * scale2X);
        values[i] = t;
        sum += t;
    }
    sum *= 2;
    sum += 1;
    double mul1 = 1. / sum;
    for (int i = 0; i < n2; i++) {
        double t = values[i] * mul1;
        k[i] = (float)t;
        k[n - 1 - i] = (float)t;
    }
    k[n2] = (float)(1. * mul1);
    return 0;
}

/* sepFilter2D f32->f32, symmetric float kernel, BORDER_REFLECT_101.
 * Model: the AVX2+FMA3 dispatch of filter.simd.hpp (what an x86-64 deployment runs):
 *   ksize>5 : RowVec_32f        s=0; s=fma(x[k],kx[k],s)              k=0..ksize-1
 *             SymmColumnVec_32f s=fma(c,ky[0],0); s=fma(x[+k]+x[-k],ky[k],s) k=1..half
 *   ksize==3: SymmRowSmallVec_32f    fma(c,k0,(l+r)*k1)
 *             SymmColumnSmallVec_32f fma(u+d,k1,fma(c,k0,0)) */
void avdo_gaussian_blur_f32(const float* src, int h, int w, int ksize, double sigma, float* dst)
{
    float kbuf[64];
    avdo_gaussian_kernel_f32(ksize, sigma, kbuf);
    int half = ksize / 2;
    const float* kc = kbuf + half;      /* centre */
    float* tmp = (float*)malloc(sizeof(float) * (size_t)h * w);
    for (int y = 0; y < h; y++) {
        const float* S = src + (int64_t)y * w;
        float* D = tmp + (int64_t)y * w;
        for (int x = 0; x < w; x++) {
            if (ksize == 3) {
                float l = S[reflect101(x - 1, w)], r = S[reflect101(x + 1, w)];
                float t = (l + r) * kc[1];
                D[x] = mac(S[x], kc[0], t);
            } else {
                float s = 0.f;
                for (int k = 0; k < ksize; k++)
                    s = mac(S[reflect101(x - half + k, w)], kbuf[k], s);
                D[x] = s;
            }
        }
        OPS32((int64_t)w * (ksize == 3 ? 4 : 2 * ksize));
    }
    for (int y = 0; y < h; y++) {
        float* D = dst + (int64_t)y * w;
        OPS32((int64_t)w * (2 + 3 * half));
        for (int x = 0; x < w; x++) {
            float c = tmp[(int64_t)y * w + x];
            float s = mac(c, kc[0], 0.f);
            for (int k = 1; k <= half; k++) {
                float a = tmp[(int64_t)reflect101(y + k, h) * w + x];
                float b = tmp[(int64_t)reflect101(y - k, h) * w + x];
                s = mac(a + b, kc[k], s);
            }
            D[x] = s;
        }
    }
    free(tmp);
}

/* resize INTER_LINEAR on CV_32F, cn channels (resize.cpp, baseline build: no FMA).
 * HResizeLinear<float>: S[sx]*a0 + S[sx+1]*a1 ; VResizeLinear<float>: S0*b0 + S1*b1.
 * An exact 2x2 decimation is rerouted to INTER_AREA fast:
 * ResizeAreaFastVec_SIMD_32f: ((a00+a01)+(a10+a11))*0.25f  (cn==1 only here). */
int avdo_resize_linear_f32(const float* src, int sh, int sw, int cn, float* dst, int dh, int dw)
{
    if (dh == sh && dw == sw) { memcpy(dst, src, sizeof(float) * (size_t)sh * sw * cn); return 0; }
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    {
        int isx = cv_round_d(scale_x), isy = cv_round_d(scale_y);
        if (fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON && isx == 2 && isy == 2) {
            if (cn != 1) return -1;
            for (int dy = 0; dy < dh; dy++)
                for (int dx = 0; dx < dw; dx++) {
                    const float* S0 = src + (int64_t)(2 * dy) * sw + 2 * dx;
                    const float* S1 = S0 + sw;
                    dst[(int64_t)dy * dw + dx] = ((S0[0] + S0[1]) + (S1[0] + S1[1])) * 0.25f;
                }
            OPS32((int64_t)dh * dw * 4);
            return 0;
        }
    }
    int* xofs = (int*)malloc(sizeof(int) * (size_t)(dw + dh));
    int* yofs = xofs + dw;
    float* alpha = (float*)malloc(sizeof(float) * (size_t)(dw + dh) * 2);
    float* beta = alpha + dw * 2;
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = imin(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        alpha[dx * 2] = 1.f - fx;
        alpha[dx * 2 + 1] = fx;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[dy * 2] = 1.f - fy;
        beta[dy * 2 + 1] = fy;
    }
    float* rows = (float*)malloc(sizeof(float) * (size_t)dw * cn * 2);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy];
        for (int k = 0; k < 2; k++) {
            int sy = clip_i(sy0 + k, 0, sh);
            const float* S = src + (int64_t)sy * sw * cn;
            float* D = rows + (int64_t)k * dw * cn;
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx] * cn;
                for (int c = 0; c < cn; c++) {
                    if (dx < xmax && MODEL(AVDO_MODEL_RESIZE_LERP))
                        D[dx * cn + c] = S[sx + c] + (S[sx + cn + c] - S[sx + c]) * alpha[dx * 2 + 1];
                    else if (dx < xmax) D[dx * cn + c] = S[sx + c] * alpha[dx * 2] + S[sx + cn + c] * alpha[dx * 2 + 1];
                    else D[dx * cn + c] = S[sx + c] * 1.f;
                }
            }
        }
        OPS32(2 * ((int64_t)xmax * cn * 3 + (int64_t)(dw - xmax) * cn) + (int64_t)dw * cn * 3);
        float b0 = beta[dy * 2], b1 = beta[dy * 2 + 1];
        const float *S0 = rows, *S1 = rows + (int64_t)dw * cn;
        float* D = dst + (int64_t)dy * dw * cn;
        if (MODEL(AVDO_MODEL_RESIZE_LERP)) for (int x = 0; x < dw * cn; x++) D[x] = S0[x] + (S1[x] - S0[x]) * b1;
        else for (int x = 0; x < dw * cn; x++) D[x] = S0[x] * b0 + S1[x] * b1;
    }
    free(rows); free(alpha); free(xofs);
    return 0;
}

/* ---------- Farneback (optflowgf.cpp, CPU path, flags=0) ---------- */
/* hal::Cholesky64f (CholImpl<double>) solving A X = I for the 6x6 moment matrix */
static int chol_inv6(double A[6][6], double B[6][6])
{
    const int m = 6, n = 6;
    double(*L)[6] = A;
    double s;
    int i, j, k;
    for (i = 0; i < m; i++) {
        for (j = 0; j < i; j++) {
            s = A[i][j];
            for (k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            L[i][j] = s * L[j][j];
        }
        s = A[i][i];
        for (k = 0; k < j; k++) { double t = L[i][k]; s -= t * t; }
        if (s < DBL_EPSILON) return 0;
        L[i][i] = 1. / sqrt(s);
    }
    for (i = 0; i < m; i++)
        for (j = 0; j < n; j++) {
            s = B[i][j];
            for (k = 0; k < i; k++) s -= L[i][k] * B[k][j];
            B[i][j] = s * L[i][i];
        }
    for (i = m - 1; i >= 0; i--)
        for (j = 0; j < n; j++) {
            s = B[i][j];
            for (k = m - 1; k > i; k--) s -= L[k][i] * B[k][j];
            B[i][j] = s * L[i][i];
        }
    return 1;
}

/* FarnebackPrepareGaussian.  g/xg/xxg point at the CENTRE tap (index -n..n valid). */
void avdo_poly_prepare(int n, double sigma, float* g, float* xg, float* xxg, double ig[4])
{
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6], invG[6][6];
    memset(G, 0, sizeof G);
    memset(invG, 0, sizeof invG);
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            /* float products exactly as the C++ expression types dictate */
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    for (int i = 0; i < 6; i++) invG[i][i] = 1.;
    chol_inv6(G, invG);
    ig[0] = invG[1][1];   /* ig11 */
    ig[1] = invG[0][3];   /* ig03 */
    ig[2] = invG[3][3];   /* ig33 */
    ig[3] = invG[5][5];   /* ig55 */
}

/* FarnebackPolyExp: src f32[h][w] -> dst f32[h][w][5] */
void avdo_poly_exp(const float* src, int h, int w, int n, double sigma, float* dst)
{
    float kbuf[3 * (2 * 16 + 1)];
    if (n > 16) return;
    float* g = kbuf + n;
    float* xg = g + n * 2 + 1;
    float* xxg = xg + n * 2 + 1;
    double ig[4];
    avdo_poly_prepare(n, sigma, g, xg, xxg, ig);
    double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
    float* rowbuf = (float*)malloc(sizeof(float) * (size_t)(w + n * 2) * 3);
    float* row = rowbuf + n * 3;
    for (int y = 0; y < h; y++) {
        float g0 = g[0], g1, g2;
        const float* srow0 = src + (int64_t)y * w;
        const float* srow1 = 0;
        float* drow = dst + (int64_t)y * w * 5;
        for (int x = 0; x < w; x++) {
            row[x * 3] = srow0[x] * g0;
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (int k = 1; k <= n; k++) {
            g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
            srow0 = src + (int64_t)imax(y - k, 0) * w;
            srow1 = src + (int64_t)imin(y + k, h - 1) * w;
            for (int x = 0; x < w; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0;
                row[x * 3 + 1] = t1;
                row[x * 3 + 2] = t2;
            }
        }
        OPS32((int64_t)w * (1 + 8 * n) + (int64_t)w * (3 + 5 * n));      /* vertical pass; float part of the horizontal one */
        OPS64((int64_t)w * (12 * n + 9));
        for (int x = 0; x < n * 3; x++) {
            row[-1 - x] = row[2 - x];
            row[w * 3 + x] = row[w * 3 + x - 3];
        }
        for (int x = 0; x < w; x++) {
            g0 = g[0];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0,
                   b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0;
                b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(rowbuf);
}

/* FarnebackUpdateMatrices: rows [y0,y1) of M from R0, R1 warped by flow */
void avdo_update_matrices(const float* R0_, const float* R1, const float* flow_, float* M_,
                          int h, int w, int y0, int y1)
{
    enum { BORDER = 5 };
    static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    const int64_t step1 = (int64_t)w * 5;
    for (int y = y0; y < y1; y++) {
        const float* flow = flow_ + (int64_t)y * w * 2;
        const float* R0 = R0_ + (int64_t)y * w * 5;
        float* M = M_ + (int64_t)y * w * 5;
        for (int x = 0; x < w; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = cv_floor_f(fx), y1i = cv_floor_f(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1; fy -= y1i;
            if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1i < (unsigned)(h - 1)) {
                const float* ptr = R1 + (int64_t)y1i * step1 + x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy),
                      a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
                OPS32(6 + 35 + 6);
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2];
                r5 = R0[x * 5 + 3];
                r6 = R0[x * 5 + 4] * 0.5f;
                OPS32(1);
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) *
                              (x >= w - BORDER ? border[w - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) *
                              (y >= h - BORDER ? border[h - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
                OPS32(8);
            }
            M[x * 5] = r4 * r4 + r6 * r6;
            M[x * 5 + 1] = (r4 + r5) * r6;
            M[x * 5 + 2] = r5 * r5 + r6 * r6;
            M[x * 5 + 3] = r4 * r2 + r6 * r3;
            M[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
        OPS32((int64_t)w * (4 + 4 + 8 + 14));      /* fx, fy and their fractions; r2, r3; the flow terms; the five products of M */
    }
}

/* FarnebackUpdateFlow_Blur: box window with running double sums, then 2x2 solve */
void avdo_update_flow_blur(const float* R0, const float* R1, float* flow_, float* matM,
                           int h, int w, int block_size, int update_matrices)
{
    int x, y;
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = imax((1 << 10) / w, block_size);
    double scale = 1. / (block_size * block_size);
    double* vbuf = (double*)malloc(sizeof(double) * (size_t)(w + m * 2 + 2) * 5);
    double* vsum = vbuf + (m + 1) * 5;
    const float* srow0 = matM;
    for (x = 0; x < w * 5; x++) vsum[x] = srow0[x] * (m + 2);   /* float * int -> float */
    OPS32((int64_t)w * 5);
    for (y = 1; y < m; y++) {
        srow0 = matM + (int64_t)imin(y, h - 1) * w * 5;
        for (x = 0; x < w * 5; x++) vsum[x] += srow0[x];
    }
    OPS64((int64_t)w * 5 * (m - 1));
    for (y = 0; y < h; y++) {
        double g11, g12, g22, h1, h2;
        float* flow = flow_ + (int64_t)y * w * 2;
        srow0 = matM + (int64_t)imax(y - m - 1, 0) * w * 5;
        const float* srow1 = matM + (int64_t)imin(y + m, h - 1) * w * 5;
        for (x = 0; x < w * 5; x++) vsum[x] += srow1[x] - srow0[x];   /* float subtract */
        OPS32((int64_t)w * 5);
        OPS64((int64_t)w * 5 + 5 * m + (int64_t)w * (10 + 5 + 4 + 1 + 8));   /* vertical adds; row init; per pixel: window, scale, det, 1/det, two flows */
        for (x = 0; x < (m + 1) * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[w * 5 + x] = vsum[w * 5 + x - 5];
        }
        g11 = vsum[0] * (m + 2);
        g12 = vsum[1] * (m + 2);
        g22 = vsum[2] * (m + 2);
        h1 = vsum[3] * (m + 2);
        h2 = vsum[4] * (m + 2);
        for (x = 1; x < m; x++) {
            g11 += vsum[x * 5];
            g12 += vsum[x * 5 + 1];
            g22 += vsum[x * 5 + 2];
            h1 += vsum[x * 5 + 3];
            h2 += vsum[x * 5 + 4];
        }
        for (x = 0; x < w; x++) {
            g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
            g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
            g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
            h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
            h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
            double g11_ = g11 * scale;
            double g12_ = g12 * scale;
            double g22_ = g22 * scale;
            double h1_ = h1 * scale;
            double h2_ = h2 * scale;
            double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            flow[x * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
            flow[x * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
        }
        y1 = y == h - 1 ? h : y - block_size;
        if (update_matrices && (y1 == h || y1 >= y0 + min_update_stripe)) {
            avdo_update_matrices(R0, R1, flow_, matM, h, w, y0, y1);
            y0 = y1;
        }
    }
    free(vbuf);
}

/* FarnebackOpticalFlowImpl::calc, CPU path, flags=0, no initial flow */
int avdo_farneback(const uint8_t* prev, const uint8_t* next, int h, int w, float* flow0,
                   double pyr_scale, int levels, int winsize, int iterations,
                   int poly_n, double poly_sigma)
{
    const int min_size = 32;
    const uint8_t* img[2] = {prev, next};
    int i, k;
    double scale;
    if (!(pyr_scale < 1) || h <= 0 || w <= 0) return -1;
    for (k = 0, scale = 1; k < levels; k++) {
        scale *= pyr_scale;
        if (w * scale < min_size || h * scale < min_size) break;
    }
    levels = k;
    if (MODEL(AVDO_MODEL_THREE_SCALES) && levels > 0) levels--;      /* "for (k = levels - 1 ...)" reading of the loop */
    size_t npix = (size_t)h * w;
    float* fimg = (float*)malloc(sizeof(float) * npix);
    float* blur = (float*)malloc(sizeof(float) * npix);
    float* I = (float*)malloc(sizeof(float) * npix);
    float* R[2] = {(float*)malloc(sizeof(float) * npix * 5), (float*)malloc(sizeof(float) * npix * 5)};
    float* M = (float*)malloc(sizeof(float) * npix * 5);
    float* prevFlow = 0;
    int pw = 0, ph = 0;
    for (k = levels; k >= 0; k--) {
        for (i = 0, scale = 1; i < k; i++) scale *= pyr_scale;
        double sigma = (1. / scale - 1) * 0.5;
        int smooth_sz = cv_round_d(sigma * 5) | 1;
        smooth_sz = imax(smooth_sz, 3);
        int width = cv_round_d(w * scale);
        int height = cv_round_d(h * scale);
        float* flow = k > 0 ? (float*)malloc(sizeof(float) * (size_t)width * height * 2) : flow0;
        if (!prevFlow) {
            memset(flow, 0, sizeof(float) * (size_t)width * height * 2);
        } else {
            avdo_resize_linear_f32(prevFlow, ph, pw, 2, flow, height, width);
            float mul = (float)(1. / pyr_scale);
            for (size_t t = 0; t < (size_t)width * height * 2; t++) flow[t] = flow[t] * mul;
            OPS32((int64_t)width * height * 2);
            if (MODEL(AVDO_MODEL_JITTER_FLOW)) ulp_jitter(flow, (size_t)width * height * 2, 77u + (uint32_t)k);
        }
        for (i = 0; i < 2; i++) {
            for (size_t t = 0; t < npix; t++) fimg[t] = (float)img[i][t];
            avdo_gaussian_blur_f32(fimg, h, w, smooth_sz, sigma, blur);
            avdo_resize_linear_f32(blur, h, w, 1, I, height, width);
            if (MODEL(AVDO_MODEL_JITTER_PYRAMID)) ulp_jitter(I, (size_t)width * height, 31u * (uint32_t)k + (uint32_t)i);
            avdo_poly_exp(I, height, width, poly_n, poly_sigma, R[i]);
        }
        avdo_update_matrices(R[0], R[1], flow, M, height, width, 0, height);
        for (i = 0; i < iterations; i++)
            avdo_update_flow_blur(R[0], R[1], flow, M, height, width, winsize, i < iterations - 1);
        if (prevFlow) free(prevFlow);
        prevFlow = flow;
        pw = width; ph = height;
    }
    free(fimg); free(blur); free(I); free(R[0]); free(R[1]); free(M);
    return 0;
}

/* ---------- NV12 -> BGR24 as libswscale's C converter does it (row N1: decode -> BGR ingest) ----------
 * cv2.VideoCapture's FFmpeg backend hands `retrieve()` a BGR24 frame made by sws_scale from the decoder's YUV
 * 4:2:0 picture (reference app/analyzers/video.py:28-32).  For equal sizes swscale takes its unscaled special
 * converter, libswscale/yuv2rgb.c: table driven, nearest chroma (a 2x2 block shares U and V), BT.601 limited range
 * for untagged streams (SWS_CS_DEFAULT).  Restated here from the published algorithm:
 *   ff_yuv2rgb_c_init_tables(): coefficients {crv 104597, cbu 132201, cgu 25675, cgv 53279} (16.16), luma gain
 *   cy = 65536*255/219, oy = 16<<16, chroma increments rescaled by 1/cy with rounding, a 1-D clipping table
 *   y_table[i] = clip_uint8((yb + 0x8000) >> 16), yb = -(384<<16) - HEADROOM*cy - oy + i*cy, and per chroma value a
 *   POINTER into that table: table_rV[V] = y_table + yoffs + ((V*crv)>>16) - (crv>>9) (fill_table), likewise bU, gU,
 *   and an integer offset gV (fill_gv_table); yoffs = 326 + HEADROOM for limited range.
 *   yuv2rgb_c_24_bgr(): dst[0] = b[Y], dst[1] = g[Y], dst[2] = r[Y] with r = table_rV[V], g = table_gU[U] + table_gV[V],
 *   b = table_bU[U].
 * PARITY UNPINNED twice over: there is no libswscale here to check the constants against (the 326 luma offset and the
 * HEADROOM of 512 are from memory), and x86 builds of FFmpeg dispatch this conversion to SSSE3/AVX2 code whose 16-bit
 * fixed-point arithmetic differs from the C tables by +-1 grey level.  NV12 (what hardware decoders emit) and planar
 * yuv420p go through the same tables; only the chroma addressing differs. */
#define YUVRGB_HEADROOM 512
typedef struct {
    uint8_t ytab[1024 + 2 * YUVRGB_HEADROOM];
    int r_off[256], b_off[256], gu_off[256], gv_off[256];   /* offsets into ytab, relative to ytab + yoffs */
    int yoffs;
} Yuv2RgbTables;

static int64_t asr64(int64_t v, int s) { return v >= 0 ? v >> s : -((-v + ((int64_t)1 << s) - 1) >> s); }   /* floor */

static void yuv2rgb_init_tables(Yuv2RgbTables* t)
{
    int64_t crv = 104597, cbu = 132201, cgu = -25675, cgv = -53279;      /* ff_yuv2rgb_coeffs[SWS_CS_DEFAULT] */
    int64_t cy = ((int64_t)1 << 16) * 255 / 219, oy = (int64_t)16 << 16;   /* limited range */
    /* contrast = saturation = 1<<16, brightness = 0: the scaling steps are identities */
    crv = (crv * 65536 + 0x8000) / cy;
    cbu = (cbu * 65536 + 0x8000) / cy;
    cgu = (cgu * 65536 + 0x8000) / cy;           /* C division truncates toward zero, as in the original */
    cgv = (cgv * 65536 + 0x8000) / cy;
    int64_t yb = -((int64_t)384 << 16) - YUVRGB_HEADROOM * cy - oy;
    for (int i = 0; i < 1024 + 2 * YUVRGB_HEADROOM; i++) {
        int64_t v = asr64(yb + 0x8000, 16);
        t->ytab[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        yb += cy;
    }
    t->yoffs = 326 + YUVRGB_HEADROOM;
    for (int i = 0; i < 256; i++) {
        t->r_off[i] = (int)(asr64(i * crv, 16) - asr64(crv, 9));
        t->b_off[i] = (int)(asr64(i * cbu, 16) - asr64(cbu, 9));
        t->gu_off[i] = (int)(asr64(i * cgu, 16) - asr64(cgu, 9));
        t->gv_off[i] = (int)(asr64(i * cgv, 16) - asr64(cgv, 9));
    }
}

/* the integer constants of the conversion, for a kernel that evaluates the tables arithmetically:
 * out[0..5] = cy, crv, cbu, cgu, cgv (after rescaling), c0 with  value = clip8((c0 + (Y + off) * cy) >> 16) */
void avdo_yuv2rgb_consts(int64_t out[6])
{
    int64_t crv = 104597, cbu = 132201, cgu = -25675, cgv = -53279;
    int64_t cy = ((int64_t)1 << 16) * 255 / 219, oy = (int64_t)16 << 16;
    out[0] = cy;
    out[1] = (crv * 65536 + 0x8000) / cy;
    out[2] = (cbu * 65536 + 0x8000) / cy;
    out[3] = (cgu * 65536 + 0x8000) / cy;
    out[4] = (cgv * 65536 + 0x8000) / cy;
    out[5] = -((int64_t)384 << 16) - oy + 326 * cy + 0x8000;
}

int avdo_nv12_to_bgr24(const uint8_t* yp, const uint8_t* uvp, int h, int w, int64_t y_stride, int64_t uv_stride,
                       uint8_t* bgr, int64_t bgr_stride)
{
    static Yuv2RgbTables T;
    static int ready = 0;
    if ((h & 1) || (w & 1) || h <= 0 || w <= 0) return -1;
    if (!ready) { yuv2rgb_init_tables(&T); ready = 1; }
    for (int y = 0; y < h; y++) {
        const uint8_t* Y = yp + (int64_t)y * y_stride;
        const uint8_t* C = uvp + (int64_t)(y >> 1) * uv_stride;
        uint8_t* D = bgr + (int64_t)y * bgr_stride;
        for (int x = 0; x < w; x++) {
            const int U = C[(x >> 1) * 2], V = C[(x >> 1) * 2 + 1];
            const uint8_t* r = T.ytab + T.yoffs + T.r_off[V];
            const uint8_t* g = T.ytab + T.yoffs + T.gu_off[U] + T.gv_off[V];
            const uint8_t* b = T.ytab + T.yoffs + T.b_off[U];
            D[3 * x] = b[Y[x]]; D[3 * x + 1] = g[Y[x]]; D[3 * x + 2] = r[Y[x]];
        }
    }
    return 0;
}

/* ---------- frame-level twins of include/avd.h ---------- */
int avdo_preprocess_bgr(const uint8_t* bgr, int n, int h, int w, int64_t row_stride,
                        int64_t frame_stride, uint8_t* small320, uint8_t* hash1024,
                        int64_t* lap_sum, int64_t* lap_sumsq)
{
    uint8_t* gray = (uint8_t*)malloc((size_t)h * w);
    uint8_t area[1024];
    int rc = 0;
    for (int f = 0; f < n && rc == 0; f++) {
        avdo_bgr2gray(bgr + (int64_t)f * frame_stride, h, w, row_stride, gray);
        rc = avdo_resize_area_u8(gray, h, w, area, 32, 32);
        if (rc) break;
        avdo_hash_bits(area, 1024, hash1024 + (int64_t)f * 1024);
        rc = avdo_resize_linear_u8(gray, h, w, small320 + (int64_t)f * 102400, 320, 320);
        avdo_laplacian_sums(gray, h, w, lap_sum + f, lap_sumsq + f);
    }
    free(gray);
    return rc;
}

int avdo_farneback_pairs(const uint8_t* small, int n, float* flow_mean, float* flow_var)
{
    float* flow = (float*)malloc(sizeof(float) * 320 * 320 * 2);
    for (int p = 0; p + 1 < n; p++) {
        int rc = avdo_farneback(small + (int64_t)p * 102400, small + (int64_t)(p + 1) * 102400,
                                320, 320, flow, 0.5, 3, 15, 3, 5, 1.2);
        if (rc) { free(flow); return rc; }
        avdo_flow_stats(flow, 102400, flow_mean + p, flow_var + p, 0);
    }
    free(flow);
    return 0;
}
