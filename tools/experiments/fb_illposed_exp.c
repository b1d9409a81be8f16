/* CPU experiment (round 4): which pairs can the "fast" level kernel (csrc/avd_fbfast.hip) NOT follow, and which cheap
 * per-pixel quantity of its solver tells?  Builds on the oracle (test infrastructure); NOT product code.
 *
 * For one pair of 320 x 320 frames:
 *   out[0]  max |fast - oracle| over the dense flow (px)          fast = mode 1 of fb_tolerance_exp.c
 *   out[1]  |flow_mean(fast) - flow_mean(oracle)|
 *   out[2]  max |oracle(+-1 ulp on every pyramid level) - oracle|  (AVDO_MODEL_JITTER_PYRAMID)
 *   out[3]  max |oracle(+-1 ulp on every up-sampled flow) - oracle| (AVDO_MODEL_JITTER_FLOW)
 *   out[4]  flow_mean(oracle), out[5] flow_var(oracle)
 *   out[8 + 8 l + i], l = level 0 (320 px) .. 3 (40 px): candidate indicators, maxima over pixels and iterations
 *     i = 0  max(|fx|, |fy|) / level width
 *     i = 1  (g11 g22 + g12^2) / (det + 1e-3)      cancellation in the determinant
 *     i = 2  the product of the two at one pixel
 *     i = 3  max(g11, g22) / (det + 1e-3)          norm of the regularised inverse
 *     i = 4  (|g11 h2| + |g12 h1| + |g22 h1| + |g12 h2|) / (det + 1e-3) / level width   cancellation in the numerators
 *     i = 6  jump max_c |R0[c] - R1[c]| at top / left border pixels whose deciding flow component is nonzero and below 1e-10 (round 5, border_ind)
 *     i = 5  the same with the threshold 1e-6
 *     i = 7  max |new flow - incoming flow| of an iteration (px of the level)
 */
#include <stdio.h>
#include "fb_tolerance_exp.c"

static double* g_ind;          /* 8 doubles of the level being processed, or NULL */


/* i = 6 / i = 5 (round 5): the SIGN of a tiny flow at the top / left border.  cv2's warp (FarnebackUpdateMatrices) is discontinuous there: at x = 0,
 * dx = -tiny gives x1 = -1, "outside" (R0 alone), dx = +tiny gives x1 = 0, "inside" (average with R1); likewise dy at y = 0.  Where both frames are
 * flat at the border the two branches agree, so the indicator is the jump: max over border pixels whose deciding component is
 * below 1e-12 of max(|R1[0]|, |R1[1]|, |R0[c] - R1[c]| c = 2..4) (cv2's own running-sum residue can reach ~1e-13 px there, with either sign);
 * i = 5: nonzero components (criterion: jump > 1e-6), i = 6: exactly zero ones, where cv2's may be +-residue (criterion: jump > 0.05 -- two
 * different flat frames, whose zero flow is structural, stay below that).  Called on the flow every matrix update reads. */
static void border_ind(const float* R0, const float* R1, const float* flow, int h, int w)
{
    if (!g_ind) return;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (x != 0 && y != 0) continue;
            const float* f = flow + ((int64_t)y * w + x) * 2;
            double d = 1e30;
            if (x == 0) d = fmin(d, fabs((double)f[0]));
            if (y == 0) d = fmin(d, fabs((double)f[1]));
            if (d >= 1e-12) continue;
            double nf = 0;
            /* what the two branches disagree by: outside takes r2 = R0[0] / 2, r3 = R0[1] / 2, r4.. = R0[2..]; inside (R0[0] - b[0]) / 2, .., (R0[2] + b[2]) / 2 ..
             * with b the sample of R1 = (for an all-but-zero deciding component) the top-left pixel of the warped position, clamped as the kernel gathers it */
            const int x1 = imin(imax((int)floorf((float)x + f[0]), 0), w - 2), y1 = imin(imax((int)floorf((float)y + f[1]), 0), h - 2);
            const float* a = R0 + ((int64_t)y * w + x) * 5; const float* b = R1 + ((int64_t)y1 * w + x1) * 5;
            nf = fmax(fabs((double)b[0]), fabs((double)b[1]));
            for (int c = 2; c < 5; c++) nf = fmax(nf, fabs((double)a[c] - (double)b[c]));
            if (d == 0 && nf > g_ind[6]) g_ind[6] = nf;     /* exactly zero: cv2's may be +-residue */
            if (d != 0 && nf > g_ind[5]) g_ind[5] = nf;     /* nonzero, below 1e-12 */
        }
}

static void blur_ind(const float* R0, const float* R1, float* flow_, float* matM, int h, int w, int block_size, int update_matrices)
{
    const int m = block_size / 2;
    const double scale = 1. / (block_size * block_size);
    double* vs = (double*)malloc(sizeof(double) * (size_t)w * 5);
    double* vrow = (double*)malloc(sizeof(double) * (size_t)(w + 2 * m + 2) * 5);
    float* newflow = (float*)malloc(sizeof(float) * (size_t)w * h * 2);
    const float* s0 = matM;
    for (int x = 0; x < w * 5; x++) vs[x] = s0[x] * (m + 2);
    for (int y = 1; y < m; y++) {
        s0 = matM + (int64_t)imin(y, h - 1) * w * 5;
        for (int x = 0; x < w * 5; x++) vs[x] += s0[x];
    }
    for (int y = 0; y < h; y++) {
        const float* a = matM + (int64_t)imin(y + m, h - 1) * w * 5;
        const float* b = matM + (int64_t)imax(y - m - 1, 0) * w * 5;
        for (int x = 0; x < w * 5; x++) vs[x] += a[x] - b[x];
        for (int x0 = 0; x0 < w; x0 += 4) {
            for (int c = 0; c < 5; c++) {
                double v[18];
                for (int j = 0; j < 18; j++) v[j] = vs[imin(imax(x0 - m + j, 0), w - 1) * 5 + c];
                double o[4];
                double A = v[3];
                for (int j = 4; j < 15; j++) A += v[j];
                const double p12 = v[1] + v[2], q = v[15] + v[16];
                o[0] = A + (v[0] + p12);
                o[1] = A + (p12 + v[15]);
                o[2] = A + (v[2] + q);
                o[3] = A + (q + v[17]);
                for (int i = 0; i < 4 && x0 + i < w; i++) vrow[(x0 + i) * 5 + c] = o[i];
            }
        }
        for (int x = 0; x < w; x++) {
            const double g11_ = vrow[x * 5] * scale, g12_ = vrow[x * 5 + 1] * scale, g22_ = vrow[x * 5 + 2] * scale;
            const double h1_ = vrow[x * 5 + 3] * scale, h2_ = vrow[x * 5 + 4] * scale;
            const double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            const float fx = (float)((g11_ * h2_ - g12_ * h1_) * idet), fy = (float)((g22_ * h1_ - g12_ * h2_) * idet);
            newflow[((int64_t)y * w + x) * 2] = fx;
            newflow[((int64_t)y * w + x) * 2 + 1] = fy;
            if (g_ind && getenv("FBX_DUMP") && w == atoi(getenv("FBX_DUMP")) && (y % 97 == 50) && (x % 61 == 30))
                fprintf(stderr, "w=%d y=%d x=%d g11=%.6g g12=%.6g g22=%.6g h1=%.6g h2=%.6g det=%.6g flow=(%.6g, %.6g)\n", w, y, x, g11_, g12_, g22_, h1_, h2_, g11_ * g22_ - g12_ * g12_, fx, fy);
            if (g_ind) {
                const double fm = fmax(fabs(fx), fabs(fy)) / w;
                const double cc = (g11_ * g22_ + g12_ * g12_) * idet;
                const double inv = fmax(g11_, g22_) * idet;
                const double nc = (fabs(g11_ * h2_) + fabs(g12_ * h1_) + fabs(g22_ * h1_) + fabs(g12_ * h2_)) * idet / w;
                if (fm > g_ind[0]) g_ind[0] = fm;
                if (cc > g_ind[1]) g_ind[1] = cc;
                if (fm * cc > g_ind[2]) g_ind[2] = fm * cc;
                if (inv > g_ind[3]) g_ind[3] = inv;
                if (nc > g_ind[4]) g_ind[4] = nc;
            }
        }
    }
    if (g_ind) {
    }
    if (g_ind) {
        /* i = 7: the iteration's UPDATE, max |new flow - incoming flow| (px of this level) */
        for (int64_t t = 0; t < (int64_t)w * h * 2; t++) {
            const double u = fabs((double)newflow[t] - (double)flow_[t]);
            if (u > g_ind[7]) g_ind[7] = u;
        }
    }
    memcpy(flow_, newflow, sizeof(float) * (size_t)w * h * 2);
    if (update_matrices) { border_ind(R0, R1, flow_, h, w); avdo_update_matrices(R0, R1, flow_, matM, h, w, 0, h); }
    free(vs); free(vrow); free(newflow);
}

static int fast_with_indicators(const uint8_t* prev, const uint8_t* next, int h, int w, float* flow0, double* ind)
{
    const double pyr_scale = 0.5; const int levels = 3, winsize = 15, iterations = 3, poly_n = 5; const double poly_sigma = 1.2;
    const uint8_t* img[2] = {prev, next};
    size_t npix = (size_t)h * w;
    float* fimg = (float*)malloc(sizeof(float) * npix); float* blur = (float*)malloc(sizeof(float) * npix);
    float* I = (float*)malloc(sizeof(float) * npix);
    float* R[2] = {(float*)malloc(sizeof(float) * npix * 5), (float*)malloc(sizeof(float) * npix * 5)};
    float* M = (float*)malloc(sizeof(float) * npix * 5);
    float* prevFlow = 0; int pw = 0, ph = 0;
    for (int k = levels; k >= 0; k--) {
        double scale = 1;
        for (int i = 0; i < k; i++) scale *= pyr_scale;
        double sigma = (1. / scale - 1) * 0.5;
        int smooth_sz = imax(cv_round_d(sigma * 5) | 1, 3);
        int width = cv_round_d(w * scale), height = cv_round_d(h * scale);
        float* flow = k > 0 ? (float*)malloc(sizeof(float) * (size_t)width * height * 2) : flow0;
        if (!prevFlow) memset(flow, 0, sizeof(float) * (size_t)width * height * 2);
        else {
            avdo_resize_linear_f32(prevFlow, ph, pw, 2, flow, height, width);
            for (size_t t = 0; t < (size_t)width * height * 2; t++) flow[t] = flow[t] * 2.f;
        }
        for (int i = 0; i < 2; i++) {
            for (size_t t = 0; t < npix; t++) fimg[t] = (float)img[i][t];
            avdo_gaussian_blur_f32(fimg, h, w, smooth_sz, sigma, blur);
            avdo_resize_linear_f32(blur, h, w, 1, I, height, width);
            avdo_poly_exp(I, height, width, poly_n, poly_sigma, R[i]);
        }
        g_ind = ind + 8 * k;
        if (prevFlow) border_ind(R[0], R[1], flow, height, width);     /* not on the coarsest level's zero flow: identical in cv2 */
        avdo_update_matrices(R[0], R[1], flow, M, height, width, 0, height);
        for (int i = 0; i < iterations; i++) blur_ind(R[0], R[1], flow, M, height, width, winsize, i < iterations - 1);
        g_ind = 0;
        if (prevFlow) free(prevFlow);
        prevFlow = flow; pw = width; ph = height;
    }
    free(fimg); free(blur); free(I); free(R[0]); free(R[1]); free(M);
    return 0;
}

static void il2mean(const float* f, int64_t npix, float* mean, float* var) { avdo_flow_stats(f, npix, mean, var, 0); }

/* prev, next: uint8[320*320]; out: double[40] */
int exp_illposed(const uint8_t* prev, const uint8_t* next, double* out, int with_sens)
{
    const int h = 320, w = 320;
    const int64_t npix = (int64_t)h * w;
    float* fo = (float*)malloc(sizeof(float) * npix * 2);
    float* ff = (float*)malloc(sizeof(float) * npix * 2);
    float* fj = (float*)malloc(sizeof(float) * npix * 2);
    memset(out, 0, sizeof(double) * 40);
    avdo_set_model(0);
    avdo_farneback(prev, next, h, w, fo, 0.5, 3, 15, 3, 5, 1.2);
    fast_with_indicators(prev, next, h, w, ff, out + 8);
    double d = 0;
    for (int64_t i = 0; i < npix * 2; i++) d = fmax(d, fabs((double)ff[i] - (double)fo[i]));
    out[0] = d;
    float mo, vo, mf, vf;
    il2mean(fo, npix, &mo, &vo);
    il2mean(ff, npix, &mf, &vf);
    out[1] = fabs((double)mf - (double)mo);
    out[4] = mo; out[5] = vo;
    out[6] = fabs((double)vf - (double)vo);
    if (with_sens) {
        for (int j = 0; j < 2; j++) {
            avdo_set_model(j == 0 ? AVDO_MODEL_JITTER_PYRAMID : AVDO_MODEL_JITTER_FLOW);
            avdo_farneback(prev, next, h, w, fj, 0.5, 3, 15, 3, 5, 1.2);
            double s = 0;
            for (int64_t i = 0; i < npix * 2; i++) s = fmax(s, fabs((double)fj[i] - (double)fo[i]));
            out[2 + j] = s;
        }
        avdo_set_model(0);
    }
    free(fo); free(ff); free(fj);
    return 0;
}

/* debug: both dense flows of one pair (oracle, fast emulation) */
int exp_flows(const uint8_t* prev, const uint8_t* next, float* fo, float* ff)
{
    double ind[40];
    memset(ind, 0, sizeof(ind));
    avdo_set_model(0);
    avdo_farneback(prev, next, 320, 320, fo, 0.5, 3, 15, 3, 5, 1.2);
    return fast_with_indicators(prev, next, 320, 320, ff, ind);
}
