// avd_audio.hip -- per-window spectral features of the audio analyzer (gfx950): SURVEY.md section 8f row N3.
//
// Replaces the loop body of reference app/analyzers/audio.py:40-61 for ALL half-second windows of a 16 kHz mono
// float32 waveform at once (the reference walks them one by one in Python):
//   rms, zero-crossing sum              audio.py:44-46   (sum of squares / |diff(sign)| per window)
//   seg * np.hanning, np.fft.rfft, |.|  audio.py:47-49   -> magnitudes, double
//   flatness / rolloff / centroid sums  audio.py:50-61   -> sum log(mag), sum mag, sum freq*mag, first index reaching 85 %
// The scalar tail (percentiles, variances, tts_like, timeline; audio.py:63-110) is O(windows) numpy on the host
// (avd_hip/audio.py), exactly the reference's calls.
//
// The transform: a window is 8000 samples (0.5 s at 16 kHz) = 80 x 100, a real DFT of 4001 bins, in double with EXACT
// twiddles (cos(2 pi j / 8000) tabulated once on the host; the sine is the entry a quarter period away):
//  * full windows: a hundred 80-point DFTs, a twiddle and eighty 100-point DFTs (k_audio_fft_a / _b), each small DFT a
//    direct sum whose factors are entries of that one table -- 2.9 M multiply-adds per window, 0.58 ms per 60 s track;
//  * any other length (the short last window of a stream, or a caller's own window size): the direct 4001 x L sum
//    (k_audio_dft: one lane per bin, window and table in LDS; 64 M multiply-adds per 8000-sample window, 3.4 ms per track
//    when every window goes this way).  Windows whose length is not a multiple of 4 read the sine from a second table.
// Either way the error against numpy's pocketfft is ~1e-13 relative (the order of additions), far inside the 1e-4
// tolerance of the path.
#include <cmath>
#include <vector>
#include "avd_internal.h"

#pragma clang fp contract(off)

namespace {

// ---- pass 1: time-domain sums and the windowed segment ----------------------------------------------------------
__global__ __launch_bounds__(256) void k_audio_prepare(const float* __restrict__ wav, int64_t n, int win,
                                                      const double* __restrict__ hann_full, const double* __restrict__ hann_last,
                                                      double* __restrict__ xw, avd_audio_window* __restrict__ out)
{
    __shared__ double ssum[4];
    __shared__ int szc[4];
    const int wdx = blockIdx.x, tid = threadIdx.x;
    const int64_t base = (int64_t)wdx * win;
    const int len = (int)((n - base) < win ? (n - base) : win);
    const double* hann = len == win ? hann_full : hann_last;
    const float* seg = wav + base;
    double sq = 0.;
    int zc = 0;
    for (int i = tid; i < len; i += 256) {
        const float v = seg[i];
        sq += (double)(v * v);                              // seg**2 is float32 in the reference; summed in double here
        xw[(int64_t)wdx * win + i] = (double)v * hann[i];
        if (i + 1 < len) {
            const float u = seg[i + 1];
            const int sv = (v > 0.f) - (v < 0.f), su = (u > 0.f) - (u < 0.f);
            zc += abs(su - sv);                             // |diff(sign(seg))|, values 0, 1, 2
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sq += __shfl_down(sq, o, 64); zc += __shfl_down(zc, o, 64); }
    if ((tid & 63) == 0) { ssum[tid >> 6] = sq; szc[tid >> 6] = zc; }
    __syncthreads();
    if (tid == 0) {
        out[wdx].sumsq = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        out[wdx].zero_cross = szc[0] + szc[1] + szc[2] + szc[3];
        out[wdx].length = len;
        out[wdx].nbins = len / 2 + 1;
    }
}

// ---- pass 2: |rfft| + 1e-9 by direct evaluation -----------------------------------------------------------------
// grid (windows, ceil(max_bins / 256)); LDS: window [L] + cosine table [L] (dynamic, 16 * L bytes)
__global__ __launch_bounds__(256) void k_audio_dft(const double* __restrict__ xw, int64_t n, int win,
                                                  const double* __restrict__ cos_full, const double* __restrict__ sin_full,
                                                  const double* __restrict__ cos_last, const double* __restrict__ sin_last,
                                                  double* __restrict__ mag, int first_window)
{
    extern __shared__ __align__(16) double lds[];
    const int wdx = first_window + blockIdx.x, tid = threadIdx.x;   // the windows before first_window went through the FFT path
    const int64_t base = (int64_t)wdx * win;
    const int L = (int)((n - base) < win ? (n - base) : win);
    const int nb = L / 2 + 1;
    if ((int)blockIdx.y * 256 >= nb) return;                // whole workgroup
    const bool full = L == win;
    const double* ct = full ? cos_full : cos_last;
    double* x = lds;
    double* c = lds + win;
    for (int i = tid; i < L; i += 256) { x[i] = xw[(int64_t)wdx * win + i]; c[i] = ct[i]; }
    __syncthreads();
    const int k = blockIdx.y * 256 + tid;
    if (k >= nb) return;
    double re = 0., im = 0.;
    if ((L & 3) == 0) {
        // sin(2 pi j / L) = cos(2 pi (j - L/4) / L): one table serves both
        const int q = L >> 2;
        int j = 0, js = L - q;                              // js = (j - q) mod L
        for (int i = 0; i < L; i++) {
            const double v = x[i];
            re += v * c[j];
            im -= v * c[js];
            j += k; if (j >= L) j -= L;
            js += k; if (js >= L) js -= L;
        }
    } else {
        // a length that is not a multiple of 4 has no quarter period in its cosine table: sines of THIS length from
        // global memory (full windows of a caller-chosen size as well as the short last window of a stream)
        const double* st = full ? sin_full : sin_last;
        int j = 0;
        for (int i = 0; i < L; i++) {
            const double v = x[i];
            re += v * c[j];
            im -= v * st[j];
            j += k; if (j >= L) j -= L;
        }
    }
    mag[(int64_t)wdx * (win / 2 + 1) + k] = hypot(re, im) + 1e-9;
}

// ---- pass 2, fast path: full windows of 8000 samples ---------------------------------------------------------------
// 8000 = 80 x 100.  With n = 100 n1 + n2 and k = k1 + 80 k2:
//   X[k1 + 80 k2] = sum_{n2} W_100^(n2 k2) * ( W_8000^(n2 k1) * sum_{n1} x[100 n1 + n2] W_80^(n1 k1) )
// i.e. a hundred 80-point DFTs, a twiddle, and eighty 100-point DFTs of which only k2 <= 50 is needed (bins 0..4000):
// 2.9 M real multiply-adds per window instead of 64 M, each of the small DFTs evaluated directly in double with entries of
// the SAME 8000-entry cosine table the direct form uses (W_80^m = W_8000^(100 m), W_100^m = W_8000^(80 m), sines a quarter
// period away): no approximation anywhere, the result differs from the direct sum only by the order of additions
// (~1e-14 relative).  The intermediate B[n2][k1] (16 bytes per entry, 128 KB per window) goes through global memory / L2.
constexpr int kFftN = 8000, kN1 = 80, kN2 = 100;

__global__ __launch_bounds__(256) void k_audio_fft_a(const double* __restrict__ xw, const double* __restrict__ cos_full,
                                                    double2* __restrict__ B)
{
    extern __shared__ __align__(16) double lds[];          // window [8000] | cosine table [8000]
    double* x = lds;
    double* c = lds + kFftN;
    const int wdx = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < kFftN; i += 256) { x[i] = xw[(int64_t)wdx * kFftN + i]; c[i] = cos_full[i]; }
    __syncthreads();
    double2* out = B + (int64_t)wdx * kFftN;
    for (int o = tid; o < kFftN; o += 256) {
        const int k1 = o / kN2, n2 = o - k1 * kN2;          // consecutive lanes: consecutive n2, (almost) the same k1
        double re = 0., im = 0.;
        int j = 0;                                           // (n1 k1 mod 80) * 100: the index of W_80^(n1 k1) in the table
        const int stepj = k1 * 100;
        for (int n1 = 0; n1 < kN1; n1++) {
            const double v = x[n1 * kN2 + n2];
            const int js = j >= 2000 ? j - 2000 : j + 6000;  // sin(t) = cos(t - pi / 2)
            re += v * c[j];
            im -= v * c[js];
            j += stepj; if (j >= kFftN) j -= kFftN;
        }
        // twiddle W_8000^(n2 k1) = cos - i sin
        const int t = n2 * k1, ts = t >= 2000 ? t - 2000 : t + 6000;
        const double tc = c[t], tsn = c[ts];
        double2 r;
        r.x = re * tc + im * tsn;
        r.y = im * tc - re * tsn;
        out[n2 * kN1 + k1] = r;
    }
}

__global__ __launch_bounds__(256) void k_audio_fft_b(const double2* __restrict__ B, const double* __restrict__ cos_full,
                                                    double* __restrict__ mag)
{
    extern __shared__ __align__(16) double lds[];          // cosine table [8000]
    double* c = lds;
    const int wdx = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < kFftN; i += 256) c[i] = cos_full[i];
    __syncthreads();
    const double2* in = B + (int64_t)wdx * kFftN;
    for (int k = tid; k <= kFftN / 2; k += 256) {
        const int k2 = k / kN1, k1 = k - k2 * kN1;          // consecutive lanes: consecutive k1 -> consecutive B entries
        double re = 0., im = 0.;
        int j = 0;                                           // (n2 k2 mod 100) * 80
        const int stepj = k2 * 80;
        for (int n2 = 0; n2 < kN2; n2++) {
            const double2 b = in[n2 * kN1 + k1];
            const int js = j >= 2000 ? j - 2000 : j + 6000;
            const double wc = c[j], ws = c[js];              // W_100^(n2 k2) = wc - i ws
            re += b.x * wc + b.y * ws;
            im += b.y * wc - b.x * ws;
            j += stepj; if (j >= kFftN) j -= kFftN;
        }
        mag[(int64_t)wdx * (kFftN / 2 + 1) + k] = hypot(re, im) + 1e-9;
    }
}

// ---- pass 3: spectral sums and the roll-off index ---------------------------------------------------------------
__global__ __launch_bounds__(256) void k_audio_reduce(const double* __restrict__ mag, int win, avd_audio_window* __restrict__ out)
{
    __shared__ double sl[4], sm[4], sf[4];
    __shared__ double total;
    const int wdx = blockIdx.x, tid = threadIdx.x;
    const int nb = out[wdx].nbins;
    const double* m = mag + (int64_t)wdx * (win / 2 + 1);
    // np.linspace(0, 1, nb): k * step with step = 1 / (nb - 1), the last element exactly 1
    const double step = nb > 1 ? 1.0 / (double)(nb - 1) : 0.0;
    double a = 0., b = 0., f = 0.;
    for (int k = tid; k < nb; k += 256) {
        const double v = m[k];
        a += log(v);
        b += v;
        f += (k == nb - 1 && nb > 1 ? 1.0 : (double)k * step) * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); f += __shfl_down(f, o, 64); }
    if ((tid & 63) == 0) { sl[tid >> 6] = a; sm[tid >> 6] = b; sf[tid >> 6] = f; }
    __syncthreads();
    if (tid == 0) {
        out[wdx].sum_log = (sl[0] + sl[1]) + (sl[2] + sl[3]);
        total = (sm[0] + sm[1]) + (sm[2] + sm[3]);
        out[wdx].sum_mag = total;
        out[wdx].sum_fmag = (sf[0] + sf[1]) + (sf[2] + sf[3]);
        // audio.py:52-58: sequential running sum (python floats = double), first k with s >= 0.85 * sum; 0 if never
        const double cutoff = 0.85 * total;
        double s = 0.;
        int idx = 0;
        for (int k = 0; k < nb; k++) {
            s += m[k];
            if (s >= cutoff) { idx = k; break; }
        }
        out[wdx].rolloff_index = idx;
    }
}

}  // namespace

// per-window features of n float32 samples (device pointer), window length `win`; out: device array of ceil(n / win)
int launch_audio_features(avd_ctx* ctx, const float* d_wav, int64_t n, int win, avd_audio_window* d_out, int nwin)
{
    if (nwin <= 0) return 0;
    if (win < 1 || win > 8192) { ctx->err = "audio window must be 1..8192 samples"; return AVD_ERR_ARG; }
    Workspace& ws = ctx->ws;
    const int last = (int)(n - (int64_t)(nwin - 1) * win);
    if (ws.audio_win != win || ws.audio_last != last) {
        // tables: np.hanning of both window lengths, cos(2 pi j / L) (and sin for the last length)
        std::vector<double> hf(win), hl(last), cf(win), sf(win), cl(last), sl(last);
        // np.hanning(M) = 0.5 + 0.5 cos(pi n / (M-1)) for n = 1-M, 3-M, ..., M-1 (ones for M = 1)
        for (int i = 0; i < win; i++) hf[i] = win == 1 ? 1.0 : 0.5 + 0.5 * std::cos(M_PI * (double)(2 * i + 1 - win) / (double)(win - 1));
        for (int i = 0; i < last; i++) hl[i] = last == 1 ? 1.0 : 0.5 + 0.5 * std::cos(M_PI * (double)(2 * i + 1 - last) / (double)(last - 1));
        for (int j = 0; j < win; j++) { cf[j] = std::cos(2.0 * M_PI * (double)j / (double)win); sf[j] = std::sin(2.0 * M_PI * (double)j / (double)win); }
        for (int j = 0; j < last; j++) { cl[j] = std::cos(2.0 * M_PI * (double)j / (double)last); sl[j] = std::sin(2.0 * M_PI * (double)j / (double)last); }
        const size_t need = (size_t)3 * win + 3 * (size_t)last;
        ws.audio_win = ws.audio_last = 0;                     // the tables are not valid again until every copy below is enqueued
        if (ws.audio_tab_elems < need) {
            if (ws.d_audio_tab) (void)hipFree(ws.d_audio_tab);
            ws.d_audio_tab = nullptr;
            ws.audio_tab_elems = 0;
            if (hipMalloc((void**)&ws.d_audio_tab, need * sizeof(double)) != hipSuccess) { ctx->err = "hipMalloc (audio tables)"; return AVD_ERR_NOMEM; }
            ws.audio_tab_elems = need;
        }
        double* t = ws.d_audio_tab;
        HIP_TRY(ctx, hipMemcpyAsync(t, hf.data(), sizeof(double) * win, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(t + win, cf.data(), sizeof(double) * win, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(t + 2 * win, sf.data(), sizeof(double) * win, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(t + 3 * win, hl.data(), sizeof(double) * last, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(t + 3 * win + last, cl.data(), sizeof(double) * last, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(t + 3 * win + 2 * last, sl.data(), sizeof(double) * last, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the host vectors go out of scope
        ws.audio_win = win; ws.audio_last = last;
    }
    // full windows of 8000 samples (the reference's half second at 16 kHz) take the two-step FFT path, everything else
    // (another window length, the short last window of a stream) the direct form
    static const bool no_fft = [] { const char* e = std::getenv("AVD_AUDIO_DIRECT"); return e && std::atoi(e) != 0; }();   // A/B switch
    const int nfull = (win == kFftN && !no_fft) ? (last == win ? nwin : nwin - 1) : 0;
    const size_t xw_need = (size_t)nwin * win, mag_need = (size_t)nwin * (win / 2 + 1), b_need = (size_t)nfull * kFftN * 2;
    if (ws.audio_buf_elems < xw_need + mag_need + b_need) {
        if (ws.d_audio_buf) (void)hipFree(ws.d_audio_buf);
        ws.d_audio_buf = nullptr;
        ws.audio_buf_elems = 0;
        if (hipMalloc((void**)&ws.d_audio_buf, (xw_need + mag_need + b_need) * sizeof(double)) != hipSuccess) { ctx->err = "hipMalloc (audio scratch)"; return AVD_ERR_NOMEM; }
        ws.audio_buf_elems = xw_need + mag_need + b_need;
    }
    double* t = ws.d_audio_tab;
    double *xw = ws.d_audio_buf, *mag = ws.d_audio_buf + xw_need;
    double2* bbuf = reinterpret_cast<double2*>(ws.d_audio_buf + xw_need + mag_need);
    hipLaunchKernelGGL(k_audio_prepare, dim3(nwin), dim3(256), 0, ctx->stream, d_wav, n, win, (const double*)t,
                       (const double*)(t + 3 * win), xw, d_out);
    // per call (a function attribute belongs to the current device; a process may hold contexts on several)
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_audio_dft, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_audio_fft_a, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * kFftN));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_audio_fft_b, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kFftN));
    if (nfull > 0) {
        hipLaunchKernelGGL(k_audio_fft_a, dim3(nfull), dim3(256), (size_t)16 * kFftN, ctx->stream, (const double*)xw, (const double*)(t + win), bbuf);
        hipLaunchKernelGGL(k_audio_fft_b, dim3(nfull), dim3(256), (size_t)8 * kFftN, ctx->stream, (const double2*)bbuf, (const double*)(t + win), mag);
    }
    if (nfull < nwin) {
        const size_t lds = (size_t)16 * win;
        hipLaunchKernelGGL(k_audio_dft, dim3(nwin - nfull, (win / 2 + 1 + 255) / 256), dim3(256), lds, ctx->stream, (const double*)xw, n, win,
                           (const double*)(t + win), (const double*)(t + 2 * win), (const double*)(t + 3 * win + last),
                           (const double*)(t + 3 * win + 2 * last), mag, nfull);
    }
    hipLaunchKernelGGL(k_audio_reduce, dim3(nwin), dim3(256), 0, ctx->stream, (const double*)mag, win, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
