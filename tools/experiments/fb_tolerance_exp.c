/* CPU experiment (round 3): what does a re-ordered box filter cost in flow accuracy?
 * Builds on the oracle (test infrastructure); NOT product code.
 *   mode 0: oracle as is (cv2's running sums)
 *   mode 1: vertical chain literal (cv2 order), horizontal 15-window sums formed directly in double
 *           (the "fast" level kernel of csrc/avd_fbfast.hip: sliding within 4-column chunks)
 *   mode 2: both directions as direct double sums (what a row-band split would do)
 *   mode 3: mode 1 + all sums in float (sanity: shows that double is needed)
 */
#include "../../oracle/avd_oracle.c"

static void blur_variant(const float* R0, const float* R1, float* flow_, float* matM, int h, int w, int block_size,
                         int update_matrices, int mode)
{
    const int m = block_size / 2;
    const double scale = 1. / (block_size * block_size);
    double* vs = (double*)malloc(sizeof(double) * (size_t)w * 5);
    double* vrow = (double*)malloc(sizeof(double) * (size_t)(w + 2 * m + 2) * 5);
    float* newflow = (float*)malloc(sizeof(float) * (size_t)w * h * 2);
    /* vertical */
    const float* s0 = matM;
    for (int x = 0; x < w * 5; x++) vs[x] = s0[x] * (m + 2);
    for (int y = 1; y < m; y++) {
        s0 = matM + (int64_t)imin(y, h - 1) * w * 5;
        for (int x = 0; x < w * 5; x++) vs[x] += s0[x];
    }
    for (int y = 0; y < h; y++) {
        if (mode == 2) {
            for (int x = 0; x < w * 5; x++) {
                double s = 0;
                for (int j = -m; j <= m; j++) s += (double)matM[(int64_t)imin(imax(y + j, 0), h - 1) * w * 5 + x];
                vs[x] = s;
            }
        } else {
            const float* a = matM + (int64_t)imin(y + m, h - 1) * w * 5;
            const float* b = matM + (int64_t)imax(y - m - 1, 0) * w * 5;
            for (int x = 0; x < w * 5; x++) vs[x] += a[x] - b[x];
        }
        /* horizontal: direct sums, 4-column chunks with a short slide (as the kernel does) */
        for (int x0 = 0; x0 < w; x0 += 4) {
            for (int c = 0; c < 5; c++) {
                double v[18];
                for (int j = 0; j < 18; j++) v[j] = vs[imin(imax(x0 - m + j, 0), w - 1) * 5 + c];
                double o[4];
                if (mode == 3) {
                    float A = 0;
                    for (int j = 3; j < 15; j++) A += (float)v[j];
                    o[0] = A + ((float)v[0] + ((float)v[1] + (float)v[2]));
                    o[1] = A + (((float)v[1] + (float)v[2]) + (float)v[15]);
                    o[2] = A + ((float)v[2] + ((float)v[15] + (float)v[16]));
                    o[3] = A + (((float)v[15] + (float)v[16]) + (float)v[17]);
                } else {
                    double A = v[3];
                    for (int j = 4; j < 15; j++) A += v[j];
                    const double p12 = v[1] + v[2], q = v[15] + v[16];
                    o[0] = A + (v[0] + p12);
                    o[1] = A + (p12 + v[15]);
                    o[2] = A + (v[2] + q);
                    o[3] = A + (q + v[17]);
                }
                for (int i = 0; i < 4 && x0 + i < w; i++) vrow[(x0 + i) * 5 + c] = o[i];
            }
        }
        for (int x = 0; x < w; x++) {
            const double g11_ = vrow[x * 5] * scale, g12_ = vrow[x * 5 + 1] * scale, g22_ = vrow[x * 5 + 2] * scale;
            const double h1_ = vrow[x * 5 + 3] * scale, h2_ = vrow[x * 5 + 4] * scale;
            const double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            newflow[((int64_t)y * w + x) * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
            newflow[((int64_t)y * w + x) * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
        }
    }
    memcpy(flow_, newflow, sizeof(float) * (size_t)w * h * 2);
    if (update_matrices) avdo_update_matrices(R0, R1, flow_, matM, h, w, 0, h);
    free(vs); free(vrow); free(newflow);
}

int exp_farneback(const uint8_t* prev, const uint8_t* next, int h, int w, float* flow0, int mode)
{
    const double pyr_scale = 0.5; int levels = 3; const int winsize = 15, iterations = 3, poly_n = 5; const double poly_sigma = 1.2;
    const uint8_t* img[2] = {prev, next};
    int i, k; double scale;
    for (k = 0, scale = 1; k < levels; k++) { scale *= pyr_scale; if (w * scale < 32 || h * scale < 32) break; }
    levels = k;
    size_t npix = (size_t)h * w;
    float* fimg = (float*)malloc(sizeof(float) * npix); float* blur = (float*)malloc(sizeof(float) * npix);
    float* I = (float*)malloc(sizeof(float) * npix);
    float* R[2] = {(float*)malloc(sizeof(float) * npix * 5), (float*)malloc(sizeof(float) * npix * 5)};
    float* M = (float*)malloc(sizeof(float) * npix * 5);
    float* prevFlow = 0; int pw = 0, ph = 0;
    for (k = levels; k >= 0; k--) {
        for (i = 0, scale = 1; i < k; i++) scale *= pyr_scale;
        double sigma = (1. / scale - 1) * 0.5;
        int smooth_sz = imax(cv_round_d(sigma * 5) | 1, 3);
        int width = cv_round_d(w * scale), height = cv_round_d(h * scale);
        float* flow = k > 0 ? (float*)malloc(sizeof(float) * (size_t)width * height * 2) : flow0;
        if (!prevFlow) memset(flow, 0, sizeof(float) * (size_t)width * height * 2);
        else {
            avdo_resize_linear_f32(prevFlow, ph, pw, 2, flow, height, width);
            for (size_t t = 0; t < (size_t)width * height * 2; t++) flow[t] = flow[t] * 2.f;
        }
        for (i = 0; i < 2; i++) {
            for (size_t t = 0; t < npix; t++) fimg[t] = (float)img[i][t];
            avdo_gaussian_blur_f32(fimg, h, w, smooth_sz, sigma, blur);
            avdo_resize_linear_f32(blur, h, w, 1, I, height, width);
            avdo_poly_exp(I, height, width, poly_n, poly_sigma, R[i]);
        }
        avdo_update_matrices(R[0], R[1], flow, M, height, width, 0, height);
        for (i = 0; i < iterations; i++) {
            if (mode == 0) avdo_update_flow_blur(R[0], R[1], flow, M, height, width, winsize, i < iterations - 1);
            else blur_variant(R[0], R[1], flow, M, height, width, winsize, i < iterations - 1, mode);
        }
        if (prevFlow) free(prevFlow);
        prevFlow = flow; pw = width; ph = height;
    }
    free(fimg); free(blur); free(I); free(R[0]); free(R[1]); free(M);
    return 0;
}
