// avd_tables.cpp -- host-side coefficient tables for the HIP kernels.
//
// The kernels are table-driven: every rounding decision that OpenCV 4.10 takes on
// the host while preparing a resize / Gaussian / polynomial-expansion (double and
// float arithmetic on scalars) is taken here, once per frame geometry, and the
// kernels only consume the resulting integers and floats.  Call sites replaced:
// reference app/analyzers/video.py:6 (INTER_AREA 32x32), :43 (INTER_LINEAR 320x320),
// :45 (calcOpticalFlowFarneback: pyramid Gaussian taps, poly_n=5 / poly_sigma=1.2).
// Compiled with -ffp-contract=off.
#include <cfloat>
#include <cmath>
#include <cstring>
#include "avd_internal.h"

namespace {

inline int round_half_even(double v) { return (int)std::nearbyint(v); }
inline int round_half_even(float v) { return (int)std::nearbyintf(v); }
inline int floor_i(float v) { int i = (int)v; return i - (i > v); }
inline int floor_i(double v) { int i = (int)v; return i - (i > v); }
inline int ceil_i(double v) { int i = (int)v; return i + (i < v); }
inline short to_q11(float v) {
    int i = round_half_even(v * 2048.f);
    return (short)std::min(32767, std::max(-32768, i));
}
inline int clamp_idx(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

// one axis of the bilinear map: centre-aligned source coordinate, edge replicate
void linear_axis(int src, int dst, bool snap_weights_at_edges, std::vector<int>& i0,
                 std::vector<int>& i1, std::vector<short>& w0, std::vector<short>& w1)
{
    const double scale = 1. / ((double)dst / src);
    i0.resize(dst); i1.resize(dst); w0.resize(dst); w1.resize(dst);
    for (int d = 0; d < dst; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = floor_i(f);
        f -= s;
        if (snap_weights_at_edges) {       // horizontal axis: weights collapse onto the edge pixel
            if (s < 0) { f = 0; s = 0; }
            if (s >= src - 1) { f = 0; s = src - 1; }
        }
        w0[d] = to_q11(1.f - f);
        w1[d] = to_q11(f);
        i0[d] = clamp_idx(s, src);         // vertical axis: rows are clipped, weights kept
        i1[d] = clamp_idx(s + 1, src);
    }
}

// one axis of the area map: destination cell d covers [d*scale, (d+1)*scale)
void area_axis(int src, int dst, double scale, AreaAxis& ax)
{
    ax.begin.assign(dst, 0); ax.count.assign(dst, 0);
    ax.w_first.assign(dst, 0.f); ax.w_mid.assign(dst, 0.f); ax.w_last.assign(dst, 0.f);
    for (int d = 0; d < dst; d++) {
        const double lo = d * scale, hi = lo + scale;
        const double cell = std::min(scale, src - lo);
        int s1 = ceil_i(lo), s2 = floor_i(hi);
        s2 = std::min(s2, src - 1);
        s1 = std::min(s1, s2);
        std::vector<float> wts;
        int first = s1;
        if (s1 - lo > 1e-3) { first = s1 - 1; wts.push_back((float)((s1 - lo) / cell)); }
        for (int s = s1; s < s2; s++) wts.push_back((float)(1.0 / cell));
        if (hi - s2 > 1e-3) wts.push_back((float)(std::min(std::min(hi - s2, 1.), cell) / cell));
        ax.begin[d] = first;
        ax.count[d] = (int)wts.size();
        if (!wts.empty()) {
            ax.w_first[d] = wts.front();
            ax.w_last[d] = wts.back();
            ax.w_mid[d] = wts.size() > 2 ? wts[1] : wts.front();
        }
    }
}

}  // namespace

void build_linear_tab(int src_h, int src_w, int dst_h, int dst_w, LinearTab& t)
{
    linear_axis(src_w, dst_w, true, t.x0, t.x1, t.a0, t.a1);
    linear_axis(src_h, dst_h, false, t.y0, t.y1, t.b0, t.b1);
}

int build_area_tab(int src_h, int src_w, int dst_h, int dst_w, AreaTab& t)
{
    const double sx = 1. / ((double)dst_w / src_w), sy = 1. / ((double)dst_h / src_h);
    if (!(sx >= 1 && sy >= 1)) return AVD_ERR_UNSUPPORTED;
    t.iscale_x = round_half_even(sx);
    t.iscale_y = round_half_even(sy);
    t.fast = std::fabs(sx - t.iscale_x) < DBL_EPSILON && std::fabs(sy - t.iscale_y) < DBL_EPSILON;
    area_axis(src_w, dst_w, sx, t.x);
    area_axis(src_h, dst_h, sy, t.y);
    return 0;
}

// ---- Farneback constants --------------------------------------------------------
namespace {

// Gaussian taps as cv::getGaussianKernel(n, sigma, CV_32F) produces them
void gaussian_taps(int n, double sigma, float* k)
{
    if (sigma <= 0 && n == 3) { k[0] = 0.25f; k[1] = 0.5f; k[2] = 0.25f; return; }
    const double sig = sigma > 0 ? sigma : std::fma((double)n, 0.15, 0.35);
    const double scale2 = -0.125 / (sig * sig);
    const int half = (n - 1) / 2;
    double v[32];
    double sum = 0;
    for (int i = 0, x = 1 - n; i < half; i++, x += 2) {
        v[i] = std::exp((double)(x * x) * scale2);
        sum += v[i];
    }
    sum *= 2;
    sum += 1;
    const double norm = 1. / sum;
    for (int i = 0; i < half; i++) k[i] = k[n - 1 - i] = (float)(v[i] * norm);
    k[half] = (float)(1. * norm);
}

// inverse of the SPD 6x6 moment matrix by Cholesky factorisation + two triangular solves
void spd_inverse6(double A[6][6], double X[6][6])
{
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < i; j++) {
            double s = A[i][j];
            for (int k = 0; k < j; k++) s -= A[i][k] * A[j][k];
            A[i][j] = s * A[j][j];
        }
        double s = A[i][i];
        for (int k = 0; k < i; k++) { double t = A[i][k]; s -= t * t; }
        A[i][i] = 1. / std::sqrt(s);
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = X[i][j];
            for (int k = 0; k < i; k++) s -= A[i][k] * X[k][j];
            X[i][j] = s * A[i][i];
        }
    for (int i = 5; i >= 0; i--)
        for (int j = 0; j < 6; j++) {
            double s = X[i][j];
            for (int k = 5; k > i; k--) s -= A[k][i] * X[k][j];
            X[i][j] = s * A[i][i];
        }
}

}  // namespace

void build_fb_consts(FbConsts& c)
{
    std::memset(&c, 0, sizeof c);
    // polynomial expansion taps, poly_n = 5, poly_sigma = 1.2 (video.py:45)
    const int n = 5;
    const double sigma = 1.2;
    float* g = c.g + n; float* xg = c.xg + n; float* xxg = c.xxg + n;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6] = {}, inv[6][6] = {};
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            const float gg = g[y] * g[x];            // float product, as the C++ types dictate
            G[0][0] += gg;
            G[1][1] += gg * x * x;
            G[3][3] += gg * x * x * x * x;
            G[5][5] += gg * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    for (int i = 0; i < 6; i++) inv[i][i] = 1.;
    spd_inverse6(G, inv);
    c.ig11 = inv[1][1]; c.ig03 = inv[0][3]; c.ig33 = inv[3][3]; c.ig55 = inv[5][5];

    // pyramid smoothing per level k (scale = 0.5^k): sigma = (1/scale-1)/2, ksize = max(round(5 sigma)|1, 3)
    for (int k = 0; k < AVD_FB_LEVELS; k++) {
        double scale = 1;
        for (int i = 0; i < k; i++) scale *= 0.5;
        const double sig = (1. / scale - 1) * 0.5;
        int ks = round_half_even(sig * 5) | 1;
        ks = std::max(ks, 3);
        c.gksize[k] = ks;
        gaussian_taps(ks, sig, c.gk[k]);
    }
}

// ---- NV12 ingest: libswscale's yuv2rgb.c table construction reduced to its integer constants ------------------
// (ff_yuv2rgb_c_init_tables for a 24-bit destination, SWS_CS_DEFAULT = BT.601, limited range, neutral brightness /
// contrast / saturation).  The tables themselves are not materialised: every entry is
// clip_uint8((yb0 + i*cy + 0x8000) >> 16) and every chroma table is an index offset, so the kernel evaluates
// value = clip8((c0 + (Y + off)*cy) >> 16) directly (the oracle keeps the literal tables and the two are compared).
void build_yuv_consts(YuvConsts& c)
{
    const long long one = 1ll << 16;
    const long long cy = one * 255 / 219;                    // luma gain of a limited-range source
    const long long oy = 16ll << 16;
    auto rescale = [&](long long inc) { return (inc * one + 0x8000) / cy; };      // "scale coefficients by cy"
    const long long crv = rescale(104597), cbu = rescale(132201), cgu = rescale(-25675), cgv = rescale(-53279);
    auto fl = [](long long v, int s) { return v >= 0 ? v >> s : -((-v + (1ll << s) - 1) >> s); };   // floor shift
    c.cy = (int)cy; c.crv = (int)crv; c.cbu = (int)cbu; c.cgu = (int)cgu; c.cgv = (int)cgv;
    c.c0 = (int)(-(384ll << 16) - oy + 326 * cy + 0x8000);   // table bias, luma offset of the limited-range tables, rounding
    c.kr = (int)-fl(crv, 9);
    c.kb = (int)-fl(cbu, 9);
    c.kg = (int)(-fl(cgu, 9) - fl(cgv, 9));
}
