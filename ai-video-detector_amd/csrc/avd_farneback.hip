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
//   * the 5-channel normal-equation image M of cv2 never exists in memory (k_uv), the only large
//     intermediate is D = vsum(x+7) - vsum(x-8) in double, stored in LDS-stageable tiles;
//   * layouts follow the access pattern of their consumer: R interleaved [y][x][5] (bilinear gathers
//     read 10 consecutive floats), flow planar, D tiled + XOR-swizzled.
// Not GEMM-shaped (11/19-tap separable stencils, 15x15 box sums, per-pixel 2x2 solves): HBM-traffic
// bound by the exact re-enactment of cv2's running double sums; MFMA is not applicable.
#include <cstdio>
#include <cstdlib>
#include "avd_internal.h"
#include "avd_fb_device.h"

#pragma clang fp contract(off)

namespace {

constexpr int S = AVD_SMALL;

// ---------------------------------------------------------------------------------------
// Gaussian pyramid level K (scale 2^-K): GaussianBlur(full-res, ksize, sigma) then the
// INTER_LINEAR decimation, which at these exact power-of-two scales is the 2x2 mean
// ((p00+p01)+(p10+p11))*0.25 of the two centre pixels.  Only the columns/rows the
// decimation reads are filtered.  One workgroup = TR output rows of one frame:
//   source rows (uint8, reflect-101) -> LDS;  row pass (FMA chain, cv2's RowVec_32f order)
//   -> LDS float [rows][NC];  column pass (SymmColumnVec_32f order) + 2x2 mean -> I[f][dy][dx].
// ---------------------------------------------------------------------------------------
template <int K>
struct PyrGeo {
    static constexpr int WL = S >> K;
    static constexpr int NC = K == 0 ? S : 2 * WL;              // filtered columns per row
    static constexpr int OFF = K == 0 ? 0 : (1 << K) / 2 - 1;
    // output rows per workgroup: four at the two coarse scales -- their long filters need (TR - 1) 2^K + 2 + 2 HALF source rows,
    // and the 40-px scale's 76 rows (50 KB of LDS for every workgroup of the launch) held a CU to three workgroups
    static constexpr int TR = K >= 2 ? 4 : 8;                   // (2 / 4 / 4 / 8 rows -- 22 KB, seven workgroups per CU -- is slower: 57 vs 52 us)
    static constexpr int TILES = WL / TR;                       // workgroups per frame
    static constexpr int KS = K == 3 ? 19 : (K == 2 ? 9 : 3), HALF = KS / 2;
    static constexpr int SROWS = K == 0 ? TR + 2 * HALF : ((TR - 1) << K) + 2 + 2 * HALF;   // source rows needed
    static constexpr int PADX = 12, PS = S + 2 * PADX;          // source rows carry their reflected borders (HALF <= 9 < PADX, PADX % 4 == 0)
    static constexpr int SRC_BYTES = SROWS * PS, LDS_BYTES = SRC_BYTES + SROWS * NC * 4;
};
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int kPyrLds = cmax(cmax(PyrGeo<3>::LDS_BYTES, PyrGeo<2>::LDS_BYTES), cmax(PyrGeo<1>::LDS_BYTES, PyrGeo<0>::LDS_BYTES));   // 29 KB

// blk = frame * TILES + tile of this scale
// pairdiff (K == 1 only, may be null): the tile also compares its source rows with the same rows of the NEXT frame and leaves "they differ" in
// pairdiff[f * TILES + t] -- the 20 tiles of the 160-px scale read every row of the frame (with overlap), so their OR says whether pair (f, f + 1)
// is a pair of bit-identical frames (which the fast level kernels exempt from the border-sign criterion, avd_fbfast.hip)
template <int K>
__device__ __forceinline__ void pyramid_body(char* lds, int blk, const uint8_t* __restrict__ small, const FbConsts* __restrict__ C,
                                             float* __restrict__ I, int n = 0, int* __restrict__ pairdiff = nullptr)
{
    using G = PyrGeo<K>;
    constexpr int WL = G::WL, NC = G::NC, OFF = G::OFF, TR = G::TR, KS = G::KS, HALF = G::HALF, SROWS = G::SROWS;
    constexpr int PADX = G::PADX, PS = G::PS;
    uint8_t (*src)[PS] = reinterpret_cast<uint8_t (*)[PS]>(lds);
    float (*rowf)[NC] = reinterpret_cast<float (*)[NC]>(lds + G::SRC_BYTES);
    const int tiles = WL / TR;
    const int f = blk / tiles, t = blk - f * tiles;
    const int tid = threadIdx.x;
    const int dy0 = t * TR;
    const int y_first = (K == 0 ? dy0 : (dy0 << K) + OFF) - HALF;        // first source row (may be < 0)
    const uint8_t* img = small + (int64_t)f * AVD_NPIX;
    const bool cmp = K == 1 && pairdiff != nullptr && f + 1 < n;       // workgroup-uniform
    int differs = 0;
    for (int it = tid; it < SROWS * (S / 4); it += 256) {
        const int r = it / (S / 4), c4 = it - r * (S / 4);
        const int y = reflect101(y_first + r, S);
        const unsigned v = reinterpret_cast<const unsigned*>(img + y * S)[c4];
        reinterpret_cast<unsigned*>(src[r] + PADX)[c4] = v;
        if (K == 1 && cmp) differs |= v != reinterpret_cast<const unsigned*>(img + AVD_NPIX + y * S)[c4];
    }
    // BORDER_REFLECT_101 columns, so that the taps below need no index arithmetic: x = -1 - i is x = 1 + i, x = S + i is S - 2 - i
    for (int it = tid; it < SROWS * 2 * HALF; it += 256) {
        const int r = it / (2 * HALF), i = it - r * (2 * HALF);
        const uint8_t* g = img + reflect101(y_first + r, S) * S;
        if (i < HALF) src[r][PADX - 1 - i] = g[1 + i];
        else src[r][PADX + S + (i - HALF)] = g[S - 2 - (i - HALF)];
    }
    if (K == 1 && cmp) {
        const int any = __syncthreads_or(differs);
        if (tid == 0) pairdiff[f * tiles + t] = any;
    } else
        __syncthreads();
    const float* kx = C->gk[K];
    if (KS == 3) {
        // every column is filtered at these scales (x = j): a lane does four of them from three aligned words
        static_assert(KS != 3 || NC == S, "the 3-tap scales filter whole rows");
        for (int it = tid; it < SROWS * (S / 4); it += 256) {
            const int r = it / (S / 4), x0 = (it - r * (S / 4)) * 4;
            const unsigned* wp = reinterpret_cast<const unsigned*>(src[r] + PADX + x0);
            const unsigned pw = wp[-1], cwd = wp[0], nw = wp[1];
            const float b[6] = {(float)(pw >> 24), (float)(cwd & 0xFFu), (float)((cwd >> 8) & 0xFFu), (float)((cwd >> 16) & 0xFFu),
                                (float)(cwd >> 24), (float)(nw & 0xFFu)};
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = __builtin_fmaf(b[e + 1], kx[1], (b[e] + b[e + 2]) * kx[0]);
            *reinterpret_cast<f4*>(&rowf[r][x0]) = o;
        }
    } else
    for (int it = tid; it < SROWS * NC; it += 256) {
        const int r = it / NC, j = it - r * NC;
        const int x = K == 0 ? j : ((j >> 1) << K) + OFF + (j & 1);
        const uint8_t* row = src[r] + PADX;
        float v;
        if (KS == 3) {
            const float l = (float)row[x - 1], c = (float)row[x], rr = (float)row[x + 1];
            v = __builtin_fmaf(c, kx[1], (l + rr) * kx[0]);
        } else {
            // the KS source bytes x - HALF .. x + HALF: aligned 32-bit LDS reads, one byte alignment of the window
            // (v_alignbyte), then one v_cvt_f32_ubyteN per tap -- instead of a byte read and a reflected index per tap
            constexpr int NW = (KS + 3) / 4;                         // aligned words of the window
            const int base = PADX + x - HALF, sh = base & 3;
            const unsigned* wp = reinterpret_cast<const unsigned*>(src[r] + (base & ~3));
            unsigned d[NW + 1];
#pragma unroll
            for (int q = 0; q <= NW; q++) d[q] = wp[q];
            v = 0.f;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const unsigned wv = __builtin_amdgcn_alignbyte(d[q + 1], d[q], sh);
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (q * 4 + e < KS) v = __builtin_fmaf((float)((wv >> (8 * e)) & 0xFFu), kx[q * 4 + e], v);
            }
        }
        rowf[r][j] = v;
    }
    __syncthreads();
    const float* kc = kx + HALF;
    auto colf = [&](int rc, int j) {                     // rc = LDS row of the centre tap
        float sacc = __builtin_fmaf(rowf[rc][j], kc[0], 0.f);
#pragma unroll
        for (int k = 1; k <= HALF; k++) sacc = __builtin_fmaf(rowf[rc + k][j] + rowf[rc - k][j], kc[k], sacc);
        return sacc;
    };
    for (int it = tid; it < TR * WL; it += 256) {
        const int dyl = it / WL, dx = it - dyl * WL;
        float out;
        if (K == 0) {
            out = colf(dyl + HALF, dx);
        } else {
            const int rc = (dyl << K) + HALF;            // LDS row of source row (dy<<K)+OFF
            const float p00 = colf(rc, 2 * dx), p01 = colf(rc, 2 * dx + 1);
            const float p10 = colf(rc + 1, 2 * dx), p11 = colf(rc + 1, 2 * dx + 1);
            out = ((p00 + p01) + (p10 + p11)) * 0.25f;
        }
        I[((int64_t)f * WL + dy0 + dyl) * WL + dx] = out;
    }
}

// the four scales of every frame in ONE launch (coarsest first: its workgroups do the most work): the small scales
// alone cannot fill the chip (600 workgroups at 40 px for a 120-frame clip) and used to run one after the other
// It also clears the per-pair ill-posedness flags of the chunk (the fast level kernels, which come later on the stream, set them).
__global__ __launch_bounds__(256) void k_pyramid_all(const uint8_t* __restrict__ small, int n, const FbConsts* __restrict__ C,
                                                    float* __restrict__ I0, float* __restrict__ I1, float* __restrict__ I2,
                                                    float* __restrict__ I3, int* __restrict__ flags, int* __restrict__ pairdiff)
{
    static_assert(PyrGeo<1>::TILES == kPairDiffTiles, "avd_fbfast.hip reads one word per 160-px tile");
    __shared__ __align__(16) char lds[kPyrLds];
    if (flags && (int)(blockIdx.x * 256 + threadIdx.x) < n - 1) flags[blockIdx.x * 256 + threadIdx.x] = 0;
    const int n3 = n * PyrGeo<3>::TILES, n2 = n * PyrGeo<2>::TILES, n1 = n * PyrGeo<1>::TILES;
    int b = blockIdx.x;
    if (b < n3) { pyramid_body<3>(lds, b, small, C, I3); return; }
    b -= n3;
    if (b < n2) { pyramid_body<2>(lds, b, small, C, I2); return; }
    b -= n2;
    if (b < n1) { pyramid_body<1>(lds, b, small, C, I1, n, pairdiff); return; }
    if (I0) pyramid_body<0>(lds, b - n1, small, C, I0);     // null: the polynomial expansion forms the 320-px scale's 3 x 3 blur itself (the grid ends before)
}

// ---------------------------------------------------------------------------------------
// FarnebackPolyExp (poly_n = 5), all four scales in ONE launch.  A 320-thread workgroup owns 320 / w consecutive image
// rows of one frame (one row at 320 px, eight at 40 px: every lane has a pixel at every scale).  Vertical 11-tap pass in
// float into LDS (3 moment planes, replicate border), horizontal pass with double accumulators, 5 interleaved
// coefficients R[y][x][c] (cv2's own layout: the level kernel gathers a pixel's five coefficients and its right-hand
// neighbour's as ten consecutive floats).  The workgroup's output is ONE contiguous span of 1600 floats: it is staged in
// LDS and stored as 16-byte pieces (a lane's own five floats are 20 bytes apart from its neighbour's).
// ---------------------------------------------------------------------------------------
struct PolyPtrs { const float* I[AVD_FB_LEVELS]; float* R[AVD_FB_LEVELS]; const uint8_t* small; };   // small != null: the 320-px scale is blurred here

constexpr int kPolyRows = 16;                            // image rows per workgroup of the full-resolution scale

// The 320-px scale (three quarters of the stage's pixels): a workgroup walks kPolyRows consecutive rows.  A lane keeps the
// eleven rows of its column that the vertical pass reads in registers and slides them (one new load per row instead of
// eleven, 18 / 8 loads per row with the halo), the two LDS buffers alternate, so a row costs ONE workgroup barrier, and the
// coalesced 16-byte stores of row i - 1 are issued while row i is in its horizontal pass.  Same arithmetic per pixel as
// the one-row form below (which still serves the three small scales): bit-identical.
// FOLD: the scale's input is not the pyramid kernel's blurred float image but the 320 x 320 gray bytes themselves; the 3 x 3 Gaussian of
// pyramid_body<0> (BORDER_REFLECT_101, the same two float passes in the same operation order: bit-identical) is formed on the way into the
// register window -- one 4-byte load per new row instead of one float load, a rolling window of three horizontally filtered rows.  Saves the
// 320-px tiles of the pyramid kernel and 49 MB written + read per 120 frames.
template <bool FOLD>
__device__ __forceinline__ void polyexp_rows320(const float* __restrict__ img, const uint8_t* __restrict__ small, float* __restrict__ out, int y0,
                                                const FbConsts* __restrict__ C, float (*rowb)[3][S + 80], float (*outb)[S * 5])
{
    constexpr int w = S, h = S;
    const int x = threadIdx.x;
    // FOLD: bytes x - 1, x, x + 1 of a row (reflected at the image edge) out of ONE 4-byte load at xs = clamp(x - 1, 0, w - 4)
    const int xs = x - 1 < 0 ? 0 : (x - 1 > w - 4 ? w - 4 : x - 1);
    const int shl = ((x == 0 ? 1 : x - 1) - xs) * 8, shc = (x - xs) * 8, shr = ((x == w - 1 ? w - 2 : x + 1) - xs) * 8;
    const float kx0 = C->gk[0][0], kx1 = C->gk[0][1], kx2 = C->gk[0][2];
    struct __attribute__((packed, aligned(1))) U4 { unsigned v; };
    auto hb = [&](int srow) __attribute__((always_inline)) {               // horizontally filtered value of source row reflect101(srow) at this column
        const int sr = srow < 0 ? -srow : (srow > h - 1 ? 2 * (h - 1) - srow : srow);
        const unsigned wd = reinterpret_cast<const U4*>(small + sr * w + xs)->v;
        const float l = (float)((wd >> shl) & 0xFFu), c = (float)((wd >> shc) & 0xFFu), r = (float)((wd >> shr) & 0xFFu);
        return __builtin_fmaf(c, kx1, (l + r) * kx0);
    };
    float hm = 0.f, h0 = 0.f, hp = 0.f;                      // filtered rows rcur - 1, rcur, rcur + 1
    int rcur = -1000;
    auto blurred = [&](int r) __attribute__((always_inline)) {             // value of the blurred image at (r, x), r clamped by the caller; rows only move down
        if (r != rcur) {                                     // workgroup-uniform
            if (r == rcur + 1) { hm = h0; h0 = hp; hp = hb(r + 1); }
            else { hm = hb(r - 1); h0 = hb(r); hp = hb(r + 1); }
            rcur = r;
        }
        return __builtin_fmaf(hp + hm, kx2, __builtin_fmaf(h0, kx1, 0.f));
    };
    auto pixel = [&](int r) __attribute__((always_inline)) { return FOLD ? blurred(r) : img[r * w + x]; };
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    // the kernel is VALU-bound (profiles/r03_experiments.md): the taps live in registers for all rows of the workgroup, as
    // floats and -- where cv2 multiplies a double by them -- as doubles; pairs of float products that share a factor are
    // formed as two-component vectors (v_pk_mul_f32 / v_pk_add_f32: one instruction for two IEEE operations, same results)
    float g[6], xg[6], xxg[6];
    double gd[6], xxgd[6];
#pragma unroll
    for (int q = 0; q <= 5; q++) { g[q] = C->g[5 + q]; xg[q] = C->xg[5 + q]; xxg[q] = C->xxg[5 + q]; gd[q] = (double)g[q]; xxgd[q] = (double)xxg[q]; }
    const double ig11 = C->ig11, ig03 = C->ig03, ig33 = C->ig33, ig55 = C->ig55;
    float win[11 + kPolyRows];                           // rows y0 - 5 .. y0 + kPolyRows + 4 of this column (clamped: replicate border)
#pragma unroll
    for (int q = 0; q < 10; q++) win[q] = pixel(min(max(y0 - 5 + q, 0), h - 1));
#pragma unroll
    for (int i = 0; i < kPolyRows; i++) {                // fully unrolled: the window is a static slice win[i .. i + 10]
        const int y = y0 + i, par = i & 1;
        win[i + 10] = pixel(min(y + 5, h - 1));
        {
            f2 t02 = f2{win[i + 5] * g[0], 0.f};
            float t1 = 0.f;
#pragma unroll
            for (int q = 1; q <= 5; q++) {
                const float a = win[i + 5 - q], bb = win[i + 5 + q];
                const float p = a + bb;
                t02 = t02 + f2{g[q], xxg[q]} * f2{p, p};  // t0 = t0 + g[q] * p;  t2 = t2 + xxg[q] * p
                t1 = t1 + xg[q] * (bb - a);
            }
            float* r0s = rowb[par][0]; float* r1s = rowb[par][1]; float* r2s = rowb[par][2];
            const float t0 = t02.x, t2 = t02.y;
            r0s[x + 5] = t0; r1s[x + 5] = t1; r2s[x + 5] = t2;
            if (x == 0)
                for (int q = 0; q < 5; q++) { r0s[q] = t0; r1s[q] = t1; r2s[q] = t2; }
            if (x == w - 1)
                for (int q = 0; q < 5; q++) { r0s[w + 5 + q] = t0; r1s[w + 5 + q] = t1; r2s[w + 5 + q] = t2; }
        }
        __syncthreads();
        if (i > 0) {                                     // row i - 1 is complete in outb[par ^ 1]: 1600 floats = 400 16-byte pieces
            f4* dst = reinterpret_cast<f4*>(out + (int64_t)(y - 1) * w * 5);
            const f4* srcv = reinterpret_cast<const f4*>(outb[par ^ 1]);
#if !(defined(AVD_POLY_ABL) && (AVD_POLY_ABL & 8))   // timing experiment: no stores of the 320-px scale
            __builtin_nontemporal_store(srcv[x], dst + x);
            if (x < S * 5 / 4 - 320) __builtin_nontemporal_store(srcv[x + 320], dst + x + 320);
#else
            if (x == 0) __builtin_nontemporal_store(srcv[x], dst + x);
#endif
        }
        const float* r0 = rowb[par][0] + x + 5; const float* r1 = rowb[par][1] + x + 5; const float* r2 = rowb[par][2] + x + 5;
        double b1 = (double)(r0[0] * g[0]), b2 = 0, b3 = (double)(r1[0] * g[0]), b4 = 0, b5 = (double)(r2[0] * g[0]), b6 = 0;
#if defined(AVD_POLY_ABL) && (AVD_POLY_ABL & 4)      // timing experiment: no horizontal pass (results are wrong)
        b2 = r0[1]; b4 = r1[1]; b6 = r2[1];
#else
#pragma unroll
        for (int q = 1; q <= 5; q++) {
            // scalar float operations on purpose: formed as two-component vectors ((r0, r1) differences, (r1, r2) sums) every pair had to
            // be assembled with register moves first -- 40 of the 208 VALU instructions of a row -- which cost more than the packing saved
            const float r0p = r0[q], r0m = r0[-q], r1p = r1[q], r1m = r1[-q], r2p = r2[q], r2m = r2[-q];
            // tg and the taps are float VALUES held in doubles: their product has at most 48 significant bits, i.e. it is exact in
            // double, so cv2's "b += tg * g" (a rounded product, then a rounded sum) IS the fused multiply-add, bit for bit -- one
            // double instruction instead of two on the pass that bounds this kernel
            const double tg = (double)(r0p + r0m);
            b1 = __builtin_fma(tg, gd[q], b1);
            b4 = __builtin_fma(tg, xxgd[q], b4);
            b2 += (double)((r0p - r0m) * xg[q]);
            b3 += (double)((r1p + r1m) * g[q]);
            b6 += (double)((r1p - r1m) * xg[q]);
            b5 += (double)((r2p + r2m) * g[q]);
        }
#endif
        float* o = outb[par] + x * 5;
        o[0] = (float)(b3 * ig11);
        o[1] = (float)(b2 * ig11);
        o[2] = (float)(b1 * ig03 + b5 * ig33);
        o[3] = (float)(b1 * ig03 + b4 * ig33);
        o[4] = (float)(b6 * ig55);
    }
    __syncthreads();
    {
        f4* dst = reinterpret_cast<f4*>(out + (int64_t)(y0 + kPolyRows - 1) * w * 5);
        const f4* srcv = reinterpret_cast<const f4*>(outb[(kPolyRows - 1) & 1]);
        __builtin_nontemporal_store(srcv[x], dst + x);
        if (x < S * 5 / 4 - 320) __builtin_nontemporal_store(srcv[x + 320], dst + x + 320);
    }
}

__global__ __launch_bounds__(320) void k_polyexp_all(PolyPtrs P, int n, const FbConsts* __restrict__ C)
{
    __shared__ float rowb[2][3][S + 80];                 // per sub-row: w + 10 entries (5 replicated on either side); two buffers
    __shared__ __align__(16) float outbb[2][S * 5];
    float (*row)[S + 80] = rowb[0];
    float* outb = outbb[0];
    // the 320-px scale first: n * 320 / kPolyRows workgroups of kPolyRows rows (consecutive row bands of a frame share an XCD)
    {
        const int wgs0 = n * (S / kPolyRows), cnt0 = ((wgs0 + 7) >> 3) << 3;
        if ((int)blockIdx.x < cnt0) {
            const int per = cnt0 >> 3, lid = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
            if (lid >= wgs0) return;
            const int f = lid / (S / kPolyRows), y0 = (lid - f * (S / kPolyRows)) * kPolyRows;
            if (P.small) polyexp_rows320<true>(nullptr, P.small + (int64_t)f * S * S, P.R[0] + (int64_t)f * S * S * 5, y0, C, rowb, outbb);
            else polyexp_rows320<false>(P.I[0] + (int64_t)f * S * S, nullptr, P.R[0] + (int64_t)f * S * S * 5, y0, C, rowb, outbb);
            return;
        }
    }
    // scale k >= 1: w = 320 >> k, rows per workgroup 1 << k, workgroups n * 320 / 4^k (padded to a multiple of 8: within a
    // scale consecutive workgroups -- overlapping 11-row windows -- share an XCD)
    int b = (int)blockIdx.x - ((((n * (S / kPolyRows)) + 7) >> 3) << 3), k = 1;
    for (; k < AVD_FB_LEVELS; k++) {
        const int cnt = ((n * (S >> (2 * k)) + 7) >> 3) << 3;
        if (b < cnt) break;
        b -= cnt;
    }
    if (k >= AVD_FB_LEVELS) return;
    const int wgs = n * (S >> (2 * k));
    const int per = ((wgs + 7) >> 3), lid = (b & 7) * per + (b >> 3);
    if (lid >= wgs) return;
    const int w = S >> k, h = w;
    const int wgs_per_frame = h >> k;
    const int f = lid / wgs_per_frame, y0 = (lid - f * wgs_per_frame) << k;
    const int tid = threadIdx.x;
    const int sr = tid / w, x = tid - sr * w;            // sub-row of this lane, column
    const int y = y0 + sr;
    const float* img = P.I[k] + (int64_t)f * w * h;
    const float* g = C->g + 5; const float* xg = C->xg + 5; const float* xxg = C->xxg + 5;
    const int pitch = w + 10;
    float* r0s = row[0] + sr * pitch; float* r1s = row[1] + sr * pitch; float* r2s = row[2] + sr * pitch;
    {
        float t0 = img[y * w + x] * g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int q = 1; q <= 5; q++) {
#if defined(AVD_POLY_ABL) && (AVD_POLY_ABL & 1)      // timing experiment: one load per pixel instead of eleven
            const float a = t0 * (float)q, bb = t0 - (float)q;
#else
            const float a = img[max(y - q, 0) * w + x], bb = img[min(y + q, h - 1) * w + x];
#endif
            const float p = a + bb;
            t0 = t0 + g[q] * p;
            t1 = t1 + xg[q] * (bb - a);
            t2 = t2 + xxg[q] * p;
        }
        r0s[x + 5] = t0; r1s[x + 5] = t1; r2s[x + 5] = t2;
        if (x == 0)
            for (int q = 0; q < 5; q++) { r0s[q] = t0; r1s[q] = t1; r2s[q] = t2; }
        if (x == w - 1)
            for (int q = 0; q < 5; q++) { r0s[w + 5 + q] = t0; r1s[w + 5 + q] = t1; r2s[w + 5 + q] = t2; }
    }
    __syncthreads();
    const float* r0 = r0s + x + 5; const float* r1 = r1s + x + 5; const float* r2 = r2s + x + 5;
#if defined(AVD_POLY_ABL) && (AVD_POLY_ABL & 2)      // timing experiment: float accumulators (results differ)
    typedef float acc_t;
#else
    typedef double acc_t;
#endif
    acc_t b1 = (acc_t)(r0[0] * g[0]), b2 = 0, b3 = (acc_t)(r1[0] * g[0]), b4 = 0,
           b5 = (acc_t)(r2[0] * g[0]), b6 = 0;
#pragma unroll
    for (int q = 1; q <= 5; q++) {
        const acc_t tg = (acc_t)(r0[q] + r0[-q]);
#if defined(AVD_POLY_ABL) && (AVD_POLY_ABL & 2)
        b1 += tg * (acc_t)g[q];
        b4 += tg * (acc_t)xxg[q];
#else
        b1 = __builtin_fma(tg, (double)g[q], b1);          // exact products (two float values): the fma IS cv2's multiply, then add
        b4 = __builtin_fma(tg, (double)xxg[q], b4);
#endif
        b2 += (acc_t)((r0[q] - r0[-q]) * xg[q]);
        b3 += (acc_t)((r1[q] + r1[-q]) * g[q]);
        b6 += (acc_t)((r1[q] - r1[-q]) * xg[q]);
        b5 += (acc_t)((r2[q] + r2[-q]) * g[q]);
    }
    float* o = outb + tid * 5;                           // (sub-row, x) order = memory order of the workgroup's rows
    o[0] = (float)(b3 * C->ig11);
    o[1] = (float)(b2 * C->ig11);
    o[2] = (float)(b1 * C->ig03 + b5 * C->ig33);
    o[3] = (float)(b1 * C->ig03 + b4 * C->ig33);
    o[4] = (float)(b6 * C->ig55);
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* dst = reinterpret_cast<f4*>(P.R[k] + ((int64_t)f * w * h + (int64_t)y0 * w) * 5);
    const f4* srcv = reinterpret_cast<const f4*>(outb);
    __builtin_nontemporal_store(srcv[tid], dst + tid);
    if (tid < S * 5 / 4 - 320) __builtin_nontemporal_store(srcv[tid + 320], dst + tid + 320);
}

// ---------------------------------------------------------------------------------------
// Initial flow of a level: zeros at the coarsest level, otherwise the previous level's
// flow resized x2 (INTER_LINEAR, float weights, cv2 edge rules) and multiplied by 2.
// flow planes: [pair][2][h][w].  The destination is exactly 2x the source, so cv2's source
// coordinate (d+0.5)*0.5-0.5 = d/2 - 0.25 is exact in float: s = floor, f in {0.75, 0.25};
// horizontally the weights snap to the edge pixel (f = 0) when s falls outside [0, pw-1),
// vertically the rows are clipped and the weights kept.  A lane produces 4 consecutive outputs.
// ---------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_flow_up(const float* __restrict__ prev, float* __restrict__ flow, int npairs, const int* __restrict__ plist)
{
    constexpr int H = W, PW = W / 2, PH = H / 2, Q = W / 4;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= npairs * 2 * H * Q) return;
    const int q = gid % Q;
    const int dy = (gid / Q) % H;
    int pc = gid / (Q * H);                                // pair*2 + channel
    if (plist) pc = plist[pc >> 1] * 2 + (pc & 1);         // plist (may be null): the launch works on pairs plist[0 .. npairs) (exact re-run of flagged pairs)
    const float* src = prev + (int64_t)pc * PW * PH;
    float fy = dy * 0.5f - 0.25f;
    const int sy = floor_f(fy);
    fy -= sy;
    const float* r0 = src + clampi(sy, 0, PH - 1) * PW;
    const float* r1 = src + clampi(sy + 1, 0, PH - 1) * PW;
    const float b0 = 1.f - fy, b1 = fy;
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = q * 4 + i;
        float fx = dx * 0.5f - 0.25f;
        int sx = floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        bool edge = false;                                  // dx >= xmax: value copied, no weights
        if (sx + 1 >= PW) { edge = true; if (sx >= PW - 1) { fx = 0; sx = PW - 1; } }
        const int x1 = min(sx + 1, PW - 1);
        const float a0 = 1.f - fx, a1 = fx;
        float d0, d1;
        if (edge) { d0 = r0[sx] * 1.f; d1 = r1[sx] * 1.f; }
        else { d0 = r0[sx] * a0 + r0[x1] * a1; d1 = r1[sx] * a0 + r1[x1] * a1; }
        o[i] = (d0 * b0 + d1 * b1) * 2.f;
    }
    *reinterpret_cast<float4*>(flow + ((int64_t)pc * H + dy) * W + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---------------------------------------------------------------------------------------
// FarnebackUpdateFlow_Blur, winsize 15 (m = 7).  cv2 keeps RUNNING box sums in double and
// rounds at every slide, so the value at (y,x) depends on the whole column / row prefix; the
// chains are reproduced literally, one lane per chain, in two kernels:
//
// k_uv / k_uvp (lanes along x, sequential in y; defined further down): FarnebackUpdateMatrices fused with
//   the vertical running sums vsum in double.  The horizontal pass only ever needs
//   D(x) = vsum(x+7) - vsum(x-8), which is formed there (a strip owns 48 output columns + 8/8 halo
//   lanes, clamped at the edge = cv2's replicate border) and is the only thing written (+ columns
//   0..6 of vsum for the row init).  D is stored in 64-row x 8-column tiles (4 KiB, one tile per
//   channel), the unit k_hscan stages through LDS; inside a tile the 8 doubles of a row are
//   XOR-swizzled by (row & 7) so that lanes reading "their" row spread over the LDS banks.  8
//   consecutive lanes write one 64-byte half line.  In both kernels the waves that LOAD never STORE
//   (one in-order vmcnt for both on this hardware) and their steps have no branches around memory
//   operations (a conditional load or store makes every later s_waitcnt conservative).
//
// k_hscan (lanes along y, sequential in x): five horizontal running sums per row in one lane,
//   2x2 solve per pixel.  Workgroup = 64 rows: wave 0 scans, wave 1 streams the next chunks'
//   five tiles (20 KiB each, perfectly coalesced, two chunks in flight in registers) into the other
//   LDS buffer; the flow leaves through a 10 KiB LDS transpose so that a store instruction writes 64
//   contiguous bytes per row instead of 16.  50 KiB of LDS per workgroup = 3 workgroups per CU, so a
//   whole clip's row blocks (595 at 320 px) are resident in one round.
// ---------------------------------------------------------------------------------------
constexpr int kStripW = 48;          // output columns per wave in k_uv: 64 lanes - 8 - 8 halo

__host__ __device__ constexpr int d16_xch(int w) { return (w + 7) / 8; }
__host__ __device__ constexpr int d16_nyb(int h) { return (h + 63) / 64; }
// tiles per pair, padded to an ODD count: an unpadded 320x320 pair is exactly 4 MiB, and a power-of-two
// stride between the pairs that all workgroups touch in lock step lands on the same HBM channels
__host__ __device__ constexpr int d16_pair_tiles(int w) { return (d16_nyb(w) * 5 * d16_xch(w)) | 1; }

// ---------------------------------------------------------------------------------------
// k_uv = FarnebackUpdateMatrices fused into the vertical pass: every lane evaluates the normal equations of its
// column row by row and feeds them straight into the vertical running sums, so the five M
// planes never exist in memory.  A row's evaluation needs two dependent memory round trips
// (flow/R0, then the bilinear gather of R1 at the warped position); they are software
// pipelined by hand: at the step that consumes row r, the gathers of row r+2 and the flow/R0
// loads of row r+4 are issued (explicit register stages).  The vsum rows go to LDS, where the strip's
// second wave forms vsum(x+7) - vsum(x-8) and stores it (ds_bpermute would cost ~20 cycles per wave64 on
// gfx950, an LDS write + two reads ~1/3 of that).
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// k_uvp: the same computation as a producer / consumer workgroup.  Per row of a strip, ~80 % of the
// instructions (loads, bilinear gather, normal equations) do not depend on the previous row; only five
// double adds per row chain.  A strip gets NPROD + 2 waves:
//   waves 2..NPROD+1 (producers): wave w evaluates the normal equations of entry NPROD*k+(w-2) in phase k
//       (entry e = image row min(e, H-1); entries 0..6 initialise the box, entry y+7 enters at step y) and
//       writes the row into a 24-slot ring in LDS.  They only LOAD: loads and stores share one in-order
//       vmcnt on this hardware, so a wave that also stores waits for its own store acknowledgements
//       whenever it waits for a prefetched load;
//   wave 0 (summer): the only sequential part -- per entry, vsum += entering row - leaving row (both read
//       from the ring, all LDS reads of a phase first), publishes the vsum rows of the phase in LDS;
//   wave 1 (storer): one phase later forms D = vsum(x+7) - vsum(x-8) from those rows and is the only wave
//       that STORES (D tiles, vsum columns 0..6) -- it never waits on memory.
// (s_memtime stamps: with a single consumer wave doing sums, D and stores, that wave was the critical path.)
// ONE barrier per NPROD rows.  Producers software-pipeline their own entries (stride NPROD rows): flow/R0
// loads three phases ahead, gathers one phase ahead, static register slots; their steps have no branches
// around memory operations.  50 KiB of LDS = 3 workgroups per CU by LDS, 2 by registers (6 waves x 104 VGPRs).
// ---------------------------------------------------------------------------------------
template <int W, int NPROD>
__global__ __launch_bounds__(64 * (NPROD + 2), (NPROD <= 4 ? 4 : 1)) void k_uvp(const float* __restrict__ R, const float* __restrict__ flow,
                                                           double* __restrict__ D16, double* __restrict__ VS0, int npairs, const int* __restrict__ plist)
{
    // NPROD = 2 .. 4: the throughput shapes (three workgroups per CU).  NPROD = 12 (round 5): the LATENCY shape of the exact re-run of a few
    // flagged pairs -- a phase lasts about one memory round trip (the gather issued in phase k is consumed in phase k + 1) however many rows
    // it brings in, so twelve producers walk a level's rows in a third of the phases (28 instead of 82 at 320 px); one 14-wave workgroup per CU.
    static_assert((NPROD >= 2 && NPROD <= 4 && 24 % NPROD == 0) || NPROD == 8 || NPROD == 12, "ring size below covers these producer counts");
    constexpr int H = W, m = 7;
    constexpr int NSTRIP = (W + kStripW - 1) / kStripW, XCH = d16_xch(W);
    constexpr int plane = W * H;
    constexpr int RSL = NPROD <= 4 ? 24 : 4 * NPROD;     // M-row ring in LDS: 15 rows of history + two phases in flight (producers write phase k+1
                                                         // while the summer still reads the leaving rows of phase k); 20 would collide
    static_assert(RSL % NPROD == 0 && RSL >= 15 + 2 * NPROD, "ring: whole phases, history + two phases in flight");
    constexpr int CH = NPROD <= 4 ? NPROD : 4;           // entries the summer / storer hold in registers at a time
    // gather lead in phases.  With four producers a phase is about one memory round trip and the gather of the next phase hides behind it.  With twelve, a
    // phase is the CU's own work for twelve rows (~1.5 us of VALU + texture addresser) and a gather issued at its END would be waited for at the START of the next:
    // it is issued TWO phases ahead, so the round trip overlaps a whole phase of work
    constexpr int GL = NPROD >= 8 ? 2 : 1, GR = GL + 1;
    constexpr int U = NPROD >= 8 ? 12 : 4;               // phases per unrolled body (static producer register slots: four input sets, GR gather sets)
    constexpr int NE = H + m;                            // entries
    constexpr int NP = (NE + NPROD - 1) / NPROD;         // producing phases
    constexpr int NPH = ((NP + 2 + U - 1) / U) * U;      // loop trip count (two drain phases + round up to the unroll)
    __shared__ float ringM[RSL][5][64];                  // normal-equation rows, slot = entry % RSL (30 KiB; 60 KiB with twelve producers)
    __shared__ double Vb[2][NPROD][5][64];               // vsum rows of a phase, summer -> storer
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroups are dealt round-robin to the 8 XCDs.  The strips of a pair share their halo columns and the
    // gathered R1 rows, and pair p+1 reads as R0 the frame that pair p gathers as R1, at about the same rows at
    // about the same time: so an XCD (one L2) gets all strips of a CONTIGUOUS run of pairs.
    const int sj = blockIdx.x >> 3, ppx = (npairs + 7) >> 3;
    const int ps = (blockIdx.x & 7) * ppx + sj / NSTRIP, strip = sj % NSTRIP;
    if (sj / NSTRIP >= ppx || ps >= npairs) return;
    const int p = plist ? plist[ps] : ps;              // ps indexes the scratch (D, vsum columns), p the pair's R and flow
    const int xl = strip * kStripW - 8 + lane;         // logical column of this lane
    const int x = clampi(xl, 0, W - 1);                // edge replicate = duplicate chain

    if (wave == 0) {
        // ------------------------------- summer ----------------------------------------------
        // The only sequential part: five running double sums per column.  Phase k takes the rows the producers
        // evaluated in phase k-1 (entries NPROD*(k-1) ..) and publishes the vsum rows.
        double vs[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < NPH; k++) {
            if (k >= 1 && k - 1 < NP) {
                const int e0 = NPROD * (k - 1), pb = (k - 1) & 1;
                if (e0 < m && e0 + NPROD >= m) {
                    // the phase that completes the initial box: vs = (m+2) * row 0 + rows 1..m-1, in that order
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        vs[c] = (double)(ringM[0][c][lane] * (float)(m + 2));
#pragma unroll
                        for (int r = 1; r < m; r++) vs[c] += (double)ringM[r][c][lane];
                    }
                }
                // entering rows (this phase's entries) and the rows leaving the box with them (row y-8 = entry e-15;
                // row 0 while the window still touches the top edge), all reads of a chunk of entries first
#pragma unroll
                for (int c0 = 0; c0 < NPROD; c0 += CH) {
                    float a[CH][5], b[CH][5];
#pragma unroll
                    for (int i = 0; i < CH; i++) {
                        const int e = e0 + c0 + i, y = e - m;
                        const int sa = e % RSL, sb = y >= m + 1 ? (e - 15) % RSL : 0;
#pragma unroll
                        for (int c = 0; c < 5; c++) { a[i][c] = ringM[sa][c][lane]; b[i][c] = ringM[sb][c][lane]; }
                    }
#pragma unroll
                    for (int i = 0; i < CH; i++) {
                        const int e = e0 + c0 + i;
                        if (e >= m && e < NE) {
#pragma unroll
                            for (int c = 0; c < 5; c++) {
                                vs[c] += (double)(a[i][c] - b[i][c]);
                                Vb[pb][c0 + i][c][lane] = vs[c];
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        return;
    }
    if (wave == 1) {
        // ------------------------------- storer ----------------------------------------------
        // Forms D = vsum(x+7) - vsum(x-8) of the rows the summer published a phase earlier and is the only
        // wave that stores (D tiles, vsum columns 0..6): it never waits on memory.
        const bool writer = lane >= 8 && lane < 8 + kStripW && xl < W;
        const bool head = strip == 0 && lane >= 8 && lane < 8 + m;
        const unsigned dbase = ((unsigned)ps * d16_pair_tiles(W) + (x >> 3)) * 512u + (x & 7);   // tile column of this lane
        const unsigned vbase = (unsigned)ps * 5u * H * 8u + (unsigned)(lane - 8);
        const int lhi = min(lane + m, 63), llo = max(lane - m - 1, 0);
        for (int k = 0; k < NPH; k++) {
            if (k >= 2 && k - 2 < NP) {
                const int e0 = NPROD * (k - 2), pb = (k - 2) & 1;
#pragma unroll
                for (int c0 = 0; c0 < NPROD; c0 += CH) {
                    double dv[CH][5];
#pragma unroll
                    for (int i = 0; i < CH; i++)
#pragma unroll
                        for (int c = 0; c < 5; c++) dv[i][c] = Vb[pb][c0 + i][c][lhi] - Vb[pb][c0 + i][c][llo];
#pragma unroll
                    for (int i = 0; i < CH; i++) {
                        const int e = e0 + c0 + i, y = e - m;
                        if (e >= m && e < NE) {
                            if (writer) {
                                const unsigned sw = (unsigned)((x & 7) ^ (y & 7)) - (unsigned)(x & 7);     // swizzled slot - plain slot
                                const unsigned t0 = dbase + ((unsigned)(y >> 6) * 5 * XCH) * 512u + (y & 63) * 8 + sw;
#pragma unroll
                                for (int c = 0; c < 5; c++) st_off_nt<double>(D16, (t0 + (unsigned)c * XCH * 512u) * 8u, dv[i][c]);
                            }
                            if (head) {
#pragma unroll
                                for (int c = 0; c < 5; c++) st_off<double>(VS0, (vbase + (unsigned)((c * H + y) * 8)) * 8u, Vb[pb][c0 + i][c][lane]);
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        return;
    }

    // ----------------------------------- producers ------------------------------------------
    const unsigned r0base = (unsigned)p * 5u * plane, r1base = r0base + 5u * plane, flbase = (unsigned)p * 2u * plane;
    const int pi = wave - 2;                             // entry index inside a phase
    NeIn in[4]; NeG g[GR];
    auto row_of = [&](int k) { return min(NPROD * k + pi, H - 1); };
#pragma unroll
    for (int k = 0; k < 3; k++) ne_load(R, flow, r0base, flbase, x, row_of(k), W, plane, in[k]);
#pragma unroll
    for (int q = 0; q < GL; q++) ne_gather(R, r1base, in[q], x, row_of(q), W, H, plane, g[q]);
    for (int kb = 0; kb < NPH; kb += U) {
#pragma unroll
        for (int kk = 0; kk < U; kk++) {
            const int k = kb + kk;
            if (k < NP) {
                // rows past the last entry (e >= NE, only in the final phase) are evaluated on clamped
                // addresses and never consumed: no branch around the loads
                float a[5];
                ne_finish(in[kk & 3], g[kk % GR], x, row_of(k), W, H, a);
                const int slot = (NPROD * k + pi) % RSL;
#pragma unroll
                for (int c = 0; c < 5; c++) ringM[slot][c][lane] = a[c];
                // refill: gathers of this wave's entry GL phases on, inputs three phases ahead
                ne_gather(R, r1base, in[(kk + GL) & 3], x, row_of(k + GL), W, H, plane, g[(kk + GL) % GR]);
                ne_load(R, flow, r0base, flbase, x, row_of(k + 3), W, plane, in[(kk + 3) & 3]);
            }
            __syncthreads();
        }
    }
}

template <int W>
__global__ __launch_bounds__(128) void k_uv(const float* __restrict__ R, const float* __restrict__ flow,
                                           double* __restrict__ D16, double* __restrict__ VS0, int npairs, const int* __restrict__ plist)
{
    constexpr int H = W, m = 7;
    constexpr int NSTRIP = (W + kStripW - 1) / kStripW, XCH = d16_xch(W);
    constexpr int plane = W * H;
    // Workgroup = one strip = compute wave + store wave (more strips per workgroup only couple them through the barrier).  The compute wave only LOADS: on gfx9-family
    // hardware loads and stores share one in-order vmcnt, so a wave that also stores D waits, at every
    // step, for the acknowledgement of stores it issued a step earlier (measured: 126 of 326 us at 320 px).
    // It publishes the two vsum rows of a step in LDS (double-buffered); after the step's barrier its
    // store wave forms D = vsum(x+7) - vsum(x-8) from them and writes the tiles, never waiting on memory.
    __shared__ double xw[2][2][5][64];                  // [buffer][row][channel][lane], 10 KiB
    __shared__ float ringl[16][5][64];                  // the compute wave's last 16 evaluated rows, slot = row & 15
    const int wv = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    // all strips of a contiguous run of pairs on one XCD (see k_uvp)
    const int sj = blockIdx.x >> 3, ppx = (npairs + 7) >> 3;
    const int ps = (blockIdx.x & 7) * ppx + sj / NSTRIP, strip = sj % NSTRIP;
    if (sj / NSTRIP >= ppx || ps >= npairs) return;      // both waves of a strip leave together
    const int p = plist ? plist[ps] : ps;                // ps indexes the scratch, p the pair's R and flow
    const int xl = strip * kStripW - 8 + lane;         // logical column of this lane
    const int x = clampi(xl, 0, W - 1);                // edge replicate = duplicate chain

    if (wv == 1) {
        const bool writer = lane >= 8 && lane < 8 + kStripW && xl < W;
        const bool head = strip == 0 && lane >= 8 && lane < 8 + m;
        const unsigned dbase = ((unsigned)ps * d16_pair_tiles(W) + (x >> 3)) * 512u + (x & 7);   // tile column of this lane
        const unsigned vbase = (unsigned)ps * 5u * H * 8u + (unsigned)(lane - 8);
        const int lhi = min(lane + m, 63), llo = max(lane - m - 1, 0);
        for (int y0 = 0; y0 < H; y0 += 2) {
            const int y1 = y0 + 1, buf = (y0 >> 1) & 1;
            __syncthreads();
            double d0[5], d1[5], h0[5], h1[5];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                d0[c] = xw[buf][0][c][lhi] - xw[buf][0][c][llo];
                d1[c] = xw[buf][1][c][lhi] - xw[buf][1][c][llo];
                h0[c] = xw[buf][0][c][lane];
                h1[c] = xw[buf][1][c][lane];
            }
            if (writer) {
                const unsigned sw0 = (unsigned)((x & 7) ^ (y0 & 7)) - (unsigned)(x & 7);   // swizzled slot - plain slot
                const unsigned sw1 = (unsigned)((x & 7) ^ (y1 & 7)) - (unsigned)(x & 7);
                const unsigned t0 = dbase + ((unsigned)(y0 >> 6) * 5 * XCH) * 512u + (y0 & 63) * 8 + sw0;
                const unsigned t1 = dbase + ((unsigned)(y1 >> 6) * 5 * XCH) * 512u + (y1 & 63) * 8 + sw1;
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    st_off_nt<double>(D16, (t0 + (unsigned)c * XCH * 512u) * 8u, d0[c]);
                    st_off_nt<double>(D16, (t1 + (unsigned)c * XCH * 512u) * 8u, d1[c]);
                }
            }
            if (head) {
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    st_off<double>(VS0, (vbase + (unsigned)((c * H + y0) * 8)) * 8u, h0[c]);
                    st_off<double>(VS0, (vbase + (unsigned)((c * H + y1) * 8)) * 8u, h1[c]);
                }
            }
        }
        return;
    }

    const unsigned r0base = (unsigned)p * 5u * plane, r1base = r0base + 5u * plane, flbase = (unsigned)p * 2u * plane;   // R[frame p], R[frame p+1]
    // Software pipeline, static register slots, FOUR steps deep: a row's evaluation needs two dependent
    // memory round trips (flow/R0, then the gather of R1 at the warped position), and under load one round
    // trip takes about as long as two steps.  At the step that consumes rows (r, r+1) the gathers of rows
    // r+4, r+5 and the flow/R0 loads of rows r+8, r+9 are issued.  The 16-row history of the box filter
    // lives in LDS (20 KiB), which is what leaves registers for in[8] and g[4].
    NeIn in[8]; NeG g[4];
    double vs[5];
#pragma unroll
    for (int r = 0; r < m; r++) {
        float a[5];
        ne_load(R, flow, r0base, flbase, x, r, W, plane, in[0]);
        ne_gather(R, r1base, in[0], x, r, W, H, plane, g[0]);
        ne_finish(in[0], g[0], x, r, W, H, a);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            ringl[r][c][lane] = a[c];
            if (r == 0) vs[c] = (double)(a[c] * (float)(m + 2));
            else vs[c] += (double)a[c];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) ne_load(R, flow, r0base, flbase, x, min(m + k, H - 1), W, plane, in[k]);
#pragma unroll
    for (int k = 0; k < 4; k++) ne_gather(R, r1base, in[k], x, min(m + k, H - 1), W, H, plane, g[k]);

    // Two rows per step: their normal equations are independent, only the five double adds per row chain.
    // The step has no branches and no stores, so every s_waitcnt vmcnt is exact.
    static_assert(H % 8 == 0, "four steps per unrolled body");
    for (int yb = 0; yb < H; yb += 8) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int j0 = 2 * jj, j1 = j0 + 1;
            const int y0 = yb + j0, y1 = y0 + 1;
            const int ra = min(y0 + m, H - 1), rb = min(y1 + m, H - 1);   // entering rows (clamped)
            float a0[5], a1[5];
            ne_finish(in[j0], g[j0 & 3], x, ra, W, H, a0);
            ne_finish(in[j1], g[j1 & 3], x, rb, W, H, a1);
            // The refills must not be scheduled above the arithmetic that consumes the old contents of their slots:
            // otherwise old and new values of a slot are live together, the new ones get other registers, and the
            // loop back-edge becomes ~60 v_mov of just-loaded registers behind an s_waitcnt vmcnt(7) -- a drain of
            // the whole software pipeline every four steps (seen in the ISA).
            __builtin_amdgcn_sched_barrier(0);
            // refill the slots just consumed: gathers two steps ahead, inputs four steps ahead
            ne_gather(R, r1base, in[(j0 + 4) & 7], x, min(ra + 4, H - 1), W, H, plane, g[j0 & 3]);
            ne_gather(R, r1base, in[(j1 + 4) & 7], x, min(rb + 4, H - 1), W, H, plane, g[j1 & 3]);
            ne_load(R, flow, r0base, flbase, x, min(ra + 8, H - 1), W, plane, in[j0]);
            ne_load(R, flow, r0base, flbase, x, min(rb + 8, H - 1), W, plane, in[j1]);
            // leaving rows y-8 (row 0 while the window still touches the top edge); read both before the
            // entering rows overwrite their slots (row y1+7 takes the slot of row y0-8)
            const int so0 = y0 >= m + 1 ? (y0 + 8) & 15 : 0, so1 = y1 >= m + 1 ? (y1 + 8) & 15 : 0;
            float b0[5], b1[5];
#pragma unroll
            for (int c = 0; c < 5; c++) { b0[c] = ringl[so0][c][lane]; b1[c] = ringl[so1][c][lane]; }
#pragma unroll
            for (int c = 0; c < 5; c++) {
                ringl[(y0 + m) & 15][c][lane] = a0[c];
                ringl[(y1 + m) & 15][c][lane] = a1[c];
                vs[c] += (double)(a0[c] - b0[c]);
                xw[jj & 1][0][c][lane] = vs[c];
                vs[c] += (double)(a1[c] - b1[c]);
                xw[jj & 1][1][c][lane] = vs[c];
            }
            __syncthreads();
        }
    }
}

// staging registers of the loader wave: one chunk = 5 channels x 4 x 16 B per lane
typedef double dbl2 __attribute__((ext_vector_type(2)));
struct ChunkRegs { dbl2 v[5][4]; };

__device__ __forceinline__ void chunk_issue(ChunkRegs& r, const double* tiles, int xch, int xc, int lane)
{
#pragma unroll
    for (int c = 0; c < 5; c++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            r.v[c][i] = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(tiles + ((int64_t)c * xch + xc) * 512 + i * 128 + lane * 2));
}

__device__ __forceinline__ void chunk_commit(const ChunkRegs& r, double (*buf)[512], int lane)
{
#pragma unroll
    for (int c = 0; c < 5; c++)
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<dbl2*>(&buf[c][i * 128 + lane * 2]) = r.v[c][i];
}

template <int W>
__global__ __launch_bounds__(128) void k_hscan(const double* __restrict__ D16, const double* __restrict__ VS0,
                                              float* __restrict__ flow, int npairs, const int* __restrict__ plist)
{
    constexpr int H = W, m = 7;
    constexpr int XCH = d16_xch(W), NYB = d16_nyb(H);
    constexpr int64_t plane = (int64_t)W * H;
    __shared__ __align__(16) double lds[2][5][512];
    __shared__ __align__(16) float outb[2][64][20];      // row stride 80 B: ds_write_b128 of 8 lanes covers all banks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ps = blockIdx.x / NYB, ybk = blockIdx.x - ps * NYB;
    const int p = plist ? plist[ps] : ps;                // ps indexes the scratch, p the pair's flow
    const double* tiles = D16 + ((int64_t)ps * d16_pair_tiles(W) + (int64_t)ybk * 5 * XCH) * 512;   // [c][xc][512]

    if (wave == 1) {
        // loader: the tile image is copied verbatim (16 B per lane, 4 KiB per channel).  Chunk xc+2 is in
        // flight into one register set while chunk xc+1 (the other set) is written to the LDS buffer the
        // scanner released last: two chunks (40 KiB) in flight per workgroup with two LDS buffers.
        ChunkRegs ra, rb;
        chunk_issue(ra, tiles, XCH, 0, lane);
        if (XCH > 1) chunk_issue(rb, tiles, XCH, 1, lane);
        chunk_commit(ra, lds[0], lane);
        __syncthreads();
        int xc = 0;
        // steady state without conditionals: behind an `if` the compiler has to assume the loads were skipped
        // and waits vmcnt(19..0) for the commit, i.e. for the chunk it has just issued as well
        for (; xc + 3 < XCH; xc += 2) {
            chunk_issue(ra, tiles, XCH, xc + 2, lane);
            __builtin_amdgcn_sched_barrier(0);          // keep the loads ahead of the LDS writes of the other set
            chunk_commit(rb, lds[1], lane);
            __syncthreads();
            chunk_issue(rb, tiles, XCH, xc + 3, lane);
            __builtin_amdgcn_sched_barrier(0);
            chunk_commit(ra, lds[0], lane);
            __syncthreads();
        }
        for (; xc < XCH; xc += 2) {
            if (xc + 2 < XCH) chunk_issue(ra, tiles, XCH, xc + 2, lane);
            if (xc + 1 < XCH) chunk_commit(rb, lds[1], lane);
            __syncthreads();
            if (xc + 1 < XCH) {
                if (xc + 3 < XCH) chunk_issue(rb, tiles, XCH, xc + 3, lane);
                if (xc + 2 < XCH) chunk_commit(ra, lds[0], lane);
                __syncthreads();
            }
        }
        return;
    }

    // scanner (wave 0)
    const int y = ybk * 64 + lane;
    const int yc = min(y, H - 1);
    double g[5];
    {
        const double* v0 = VS0 + ((int64_t)ps * 5 * H + yc) * 8;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const double* vc = v0 + (int64_t)c * H * 8;
            double s = vc[0] * (double)(m + 2);
#pragma unroll
            for (int k = 1; k < m; k++) s += vc[k];
            g[c] = s;
        }
    }
    __syncthreads();
    const double scale = 1. / (15 * 15);
    const int sw = lane & 7;
    auto scan8 = [&](int buf, float (&ox)[8], float (&oy)[8]) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int c = 0; c < 5; c++) g[c] += lds[buf][c][lane * 8 + (j ^ sw)];
            const double g11 = g[0] * scale, g12 = g[1] * scale, g22 = g[2] * scale;
            const double h1 = g[3] * scale, h2 = g[4] * scale;
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            ox[j] = (float)((g11 * h2 - g12 * h1) * idet);
            oy[j] = (float)((g22 * h1 - g12 * h2) * idet);
        }
    };
    // Results leave through a small LDS transpose: a lane owns a ROW, so direct stores would touch 64
    // different lines with 16 B each per instruction (measured: 43 of the kernel's 153 us at 320 px).
    // Re-read as [16 rows][4 lanes x 16 B], one store instruction covers 16 rows x 64 contiguous bytes.
    const int tr = lane >> 2, tq = lane & 3;
    float* fout = flow + (int64_t)p * 2 * plane + (int64_t)(ybk * 64 + tr) * W + tq * 4;
    for (int xc = 0; xc < XCH; xc += 2) {
        float ax[8], ay[8], bx[8], by[8];
        scan8(0, ax, ay);
        __syncthreads();
        const bool second = xc + 1 < XCH;
        if (second) scan8(1, bx, by);
        {
            float4* ox = reinterpret_cast<float4*>(outb[0][lane]);
            float4* oy = reinterpret_cast<float4*>(outb[1][lane]);
            ox[0] = make_float4(ax[0], ax[1], ax[2], ax[3]); ox[1] = make_float4(ax[4], ax[5], ax[6], ax[7]);
            oy[0] = make_float4(ay[0], ay[1], ay[2], ay[3]); oy[1] = make_float4(ay[4], ay[5], ay[6], ay[7]);
            if (second) {
                ox[2] = make_float4(bx[0], bx[1], bx[2], bx[3]); ox[3] = make_float4(bx[4], bx[5], bx[6], bx[7]);
                oy[2] = make_float4(by[0], by[1], by[2], by[3]); oy[3] = make_float4(by[4], by[5], by[6], by[7]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (second || tq < 2) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = tr + 16 * k;
                if (ybk * 64 + r < H) {
                    float* o = fout + (int64_t)k * 16 * W + xc * 8;
                    *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(&outb[0][r][tq * 4]);
                    *reinterpret_cast<float4*>(o + plane) = *reinterpret_cast<const float4*>(&outb[1][r][tq * 4]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (second) __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// k_hscan_lat: the same horizontal pass in a LATENCY shape, for the exact re-run of a few flagged pairs (round 5).  In k_hscan a lane owns a
// row and walks its 320 columns alone: five dependent double adds AND the 2 x 2 solve with its IEEE division per column, ~115 ns per column,
// 37 us per launch however few pairs there are.  Only the adds are a chain.  Here the scanner wave does nothing but the chain (g of a chunk of
// eight columns goes to LDS), and four SOLVER waves one chunk behind turn g into flow -- 512 (row, column) solves per chunk, two per lane, the
// same expressions in the same order as k_hscan (bit-identical) -- and store it through a 16-row transpose.  One barrier per chunk.
// ---------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(384) void k_hscan_lat(const double* __restrict__ D16, const double* __restrict__ VS0,
                                                  float* __restrict__ flow, int npairs, const int* __restrict__ plist)
{
    constexpr int H = W, m = 7;
    constexpr int XCH = d16_xch(W), NYB = d16_nyb(H);
    static_assert(XCH % 2 == 0, "the loader alternates two register sets");
    constexpr int64_t plane = (int64_t)W * H;
    constexpr int GS = 5 * 64 + 2;                        // doubles per column of a g buffer (+ 2: the four columns a solver instruction reads fall on different banks)
    __shared__ __align__(16) double lds[2][5][512];
    __shared__ __align__(16) double gbuf[2][8][GS];
    __shared__ __align__(16) float outb[4][2][16][8];     // per solver wave: [component][row][column of the chunk]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ps = blockIdx.x / NYB, ybk = blockIdx.x - ps * NYB;
    const int p = plist ? plist[ps] : ps;                // ps indexes the scratch, p the pair's flow
    const double* tiles = D16 + ((int64_t)ps * d16_pair_tiles(W) + (int64_t)ybk * 5 * XCH) * 512;   // [c][xc][512]
    if (wave == 1) {
        // loader: chunk xc + 1 is committed to LDS while the scanner walks chunk xc; two chunks further are in flight
        ChunkRegs ra, rb;
        chunk_issue(ra, tiles, XCH, 0, lane);
        chunk_issue(rb, tiles, XCH, 1, lane);
        chunk_commit(ra, lds[0], lane);
        __syncthreads();
        for (int xc = 0; xc < XCH; xc += 2) {
            if (xc + 2 < XCH) chunk_issue(ra, tiles, XCH, xc + 2, lane);
            chunk_commit(rb, lds[1], lane);
            __syncthreads();
            if (xc + 3 < XCH) chunk_issue(rb, tiles, XCH, xc + 3, lane);
            if (xc + 2 < XCH) chunk_commit(ra, lds[0], lane);
            __syncthreads();
        }
        return;
    }
    if (wave == 0) {
        // scanner: cv2's running sums g += D(x), literally; nothing else
        const int yc = min(ybk * 64 + lane, H - 1);
        double g[5];
        const double* v0 = VS0 + ((int64_t)ps * 5 * H + yc) * 8;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const double* vc = v0 + (int64_t)c * H * 8;
            double s = vc[0] * (double)(m + 2);
#pragma unroll
            for (int k = 1; k < m; k++) s += vc[k];
            g[c] = s;
        }
        __syncthreads();
        const int sw = lane & 7;
        for (int xc = 0; xc < XCH; xc++) {
            const int buf = xc & 1;
#pragma unroll
            for (int j = 0; j < 8; j++) {
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    g[c] += lds[buf][c][lane * 8 + (j ^ sw)];
                    gbuf[buf][j][c * 64 + lane] = g[c];
                }
            }
            __syncthreads();
        }
        return;
    }
    // solvers: wave s owns rows 16 s .. 16 s + 15 of the block; lane = (row, column j) and (row, column j + 4) of the chunk
    const int s4 = wave - 2, r16 = lane & 15, j0 = lane >> 4;
    const int row = 16 * s4 + r16;
    const double scale = 1. / (15 * 15);
    float (*ob)[16][8] = outb[s4];
    const int sr = (lane & 31) >> 1, sh = lane & 1, sc = lane >> 5;       // store phase: row, 16-byte half of the chunk's 32 bytes, component
    const bool store_ok = ybk * 64 + 16 * s4 + sr < H;
    float* fout = flow + (int64_t)p * 2 * plane + (int64_t)sc * plane + (int64_t)(ybk * 64 + 16 * s4 + sr) * W + sh * 4;
    auto solve = [&](int xc) {
        const int buf = xc & 1;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int j = j0 + 4 * i;
            const double* gp = &gbuf[buf][j][row];
            const double g11 = gp[0] * scale, g12 = gp[64] * scale, g22 = gp[128] * scale;
            const double h1 = gp[192] * scale, h2 = gp[256] * scale;
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            ob[0][r16][j] = (float)((g11 * h2 - g12 * h1) * idet);
            ob[1][r16][j] = (float)((g22 * h1 - g12 * h2) * idet);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (store_ok) *reinterpret_cast<float4*>(fout + xc * 8) = *reinterpret_cast<const float4*>(&ob[sc][sr][sh * 4]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    __syncthreads();
    for (int xc = 0; xc < XCH; xc++) {
        if (xc >= 1) solve(xc - 1);
        __syncthreads();
    }
    solve(XCH - 1);
}

// ---------------------------------------------------------------------------------------
// Flow statistics in numpy's float32 order (video.py:46-48): mag = sqrt(fx*fx + fy*fy);
// add.reduce = pairwise sums (128-element leaves, 8 strided accumulators) inside 8192-element
// iterator buffers whose results are added sequentially.  One workgroup per pair.
// ---------------------------------------------------------------------------------------
constexpr int kChunk = 8192;
constexpr int kNChunk = (AVD_NPIX + kChunk - 1) / kChunk;      // 13: twelve full buffers + one of 4096

// sequential combination of the per-buffer sums, as the ufunc reduction loop does
__device__ __forceinline__ float chunk_total(const float* part)
{
    float t = part[0];
#pragma unroll
    for (int i = 1; i < kNChunk; i++) t += part[i];
    return t;
}

// Both statistics of a pair in ONE workgroup of 512 threads, the pair's 102 400 magnitudes in REGISTERS (200 per thread):
// thread (leaf l = tid / 8, accumulator a = tid % 8) holds, for each of the 13 iterator buffers, the 16 elements
// l * 128 + a + 8 k that numpy's unrolled leaf loop adds into accumulator a.  A buffer's sum is then: the sequential
// 16-element sum per thread, the 8-accumulator tree ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the balanced tree over the
// buffer's 64 (last buffer: 32) leaves -- shuffles inside a wave (8 leaves), eight per-wave values through LDS.  The
// second pass, (mag - mean)^2 with mean = sum / N in float32, runs on the registers: the magnitudes are read once.
// (18.6 us for 119 pairs; a variant with 16-byte loads and the 16-element sum relayed over four lanes took 22.9 us.)
// mag[pair][N] = |flow|: written by the last launch of the fast level kernel, or by k_mag below from the two flow planes
__global__ __launch_bounds__(512) void k_stats_pair(const float* __restrict__ mag, float* __restrict__ stats, const int* __restrict__ plist)
{
    __shared__ float wpart[kNChunk][8];
    __shared__ float ctot[kNChunk];
    const int p = plist ? plist[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
    const int l = tid >> 3, wave = tid >> 6;
    const bool in_last = l < ((AVD_NPIX - (kNChunk - 1) * kChunk) >> 7);          // the last buffer has 32 leaves
    const float* src = mag + (int64_t)p * AVD_NPIX + l * 128 + (tid & 7);
    float v[kNChunk][16];
#pragma unroll
    for (int c = 0; c < kNChunk; c++)
#pragma unroll
        for (int k = 0; k < 16; k++) v[c][k] = (c < kNChunk - 1 || in_last) ? src[c * kChunk + 8 * k] : 0.f;
    float total[2];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
        for (int c = 0; c < kNChunk; c++) {
            float r = v[c][0];
#pragma unroll
            for (int k = 1; k < 16; k++) r += v[c][k];
            // lanes whose partner holds no valid subtree compute values nobody reads
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) r = r + __shfl_down(r, d);
            if ((tid & 63) == 0) wpart[c][wave] = r;
        }
        __syncthreads();
        if (tid < kNChunk) {
            const float* w = wpart[tid];
            ctot[tid] = tid < kNChunk - 1 ? ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7])) : (w[0] + w[1]) + (w[2] + w[3]);
        }
        __syncthreads();
        total[pass] = chunk_total(ctot);
        if (pass == 0) {
            const float mean32 = total[0] / (float)AVD_NPIX;                       // _var: f32 true_divide
#pragma unroll
            for (int c = 0; c < kNChunk; c++)
#pragma unroll
                for (int k = 0; k < 16; k++) { const float d = v[c][k] - mean32; v[c][k] = d * d; }
            __syncthreads();                                                       // ctot / wpart are rewritten by the second pass
        }
    }
    // stats[pair] = { f32(f64(sum)/N), f32(f64(sumsq)/N) }   (_mean / _var final scalar divides)
    if (tid == 0) {
        stats[2 * p] = (float)((double)total[0] / (double)AVD_NPIX);
        stats[2 * p + 1] = (float)((double)total[1] / (double)AVD_NPIX);
    }
}

// mag = np.sqrt(fx * fx + fy * fy) in float32 (video.py:46) from the planar flow [pair][2][N]: the exact-mode level kernels
// leave the flow only, the fast one writes the magnitudes itself
__global__ void k_mag(const float* __restrict__ flow, float* __restrict__ mag, int64_t total4, const int* __restrict__ plist)
{
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total4) return;
    const int64_t ps = gid / (AVD_NPIX / 4), i = gid - ps * (AVD_NPIX / 4);
    const int64_t p = plist ? plist[ps] : ps;
    const float4 fx = reinterpret_cast<const float4*>(flow + p * 2 * AVD_NPIX)[i], fy = reinterpret_cast<const float4*>(flow + (p * 2 + 1) * AVD_NPIX)[i];
    float4 m;
    m.x = sqrtf(fx.x * fx.x + fy.x * fy.x);
    m.y = sqrtf(fx.y * fx.y + fy.y * fy.y);
    m.z = sqrtf(fx.z * fx.z + fy.z * fy.z);
    m.w = sqrtf(fx.w * fx.w + fy.w * fy.w);
    reinterpret_cast<float4*>(mag + p * AVD_NPIX)[i] = m;
}

// planar flow [pair][2][N] -> cv2's interleaved [pair][N][2] (only when the caller asks for the flow)
__global__ void k_flow_interleave(const float* __restrict__ flow, float* __restrict__ out, int64_t total)
{
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int64_t p = gid / AVD_NPIX, i = gid - p * AVD_NPIX;
    out[gid * 2] = flow[p * 2 * AVD_NPIX + i];
    out[gid * 2 + 1] = flow[(p * 2 + 1) * AVD_NPIX + i];
}

// A segment = a run of consecutive pairs processed on one stream, with its own slice of the workspace
// (frames [frame_off, ...) and pairs [pair_off, ...) of the chunk).  The library runs one segment per chunk:
// splitting a clip over two streams did not pay (the level-0 kernels share the bandwidth, the latency-bound
// levels are duplicated); keeping several CLIPS in flight on separate contexts is what fills the gaps.
struct Seg {
    hipStream_t stream;
    const float* pyr[AVD_FB_LEVELS];
    float* pyr_w[AVD_FB_LEVELS];
    float* poly[AVD_FB_LEVELS];
    float* flow[AVD_FB_LEVELS];
    double *vs, *vs0;
    float *stats, *flow_il;
    int* flags;                          // ill-posedness flags of the segment's pairs (fast mode)
    int* pairdiff;                       // [pair][kPairDiffTiles] "frame p differs from frame p + 1" per tile of the pyramid kernel's 160-px scale
    avd_ctx* prof;                       // non-null: record kernel events of the full-resolution blur launches
};

static Seg make_seg(avd_ctx* ctx, hipStream_t stream, int frame_off, int pair_off)
{
    Workspace& ws = ctx->ws;
    Seg g{};
    g.stream = stream;
    for (int k = 0; k < AVD_FB_LEVELS; k++) {
        const size_t plane = (size_t)(S >> k) * (S >> k);
        g.pyr_w[k] = ws.d_pyr[k] + (size_t)frame_off * plane;
        g.pyr[k] = g.pyr_w[k];
        g.poly[k] = ws.d_poly[k] + (size_t)frame_off * 5 * plane;
        g.flow[k] = ws.d_flow[k] + (size_t)pair_off * 2 * plane;
    }
    g.prof = ctx->profiling && pair_off == 0 ? ctx : nullptr;
    g.vs = ws.d_vs ? ws.d_vs + (size_t)pair_off * (5 * AVD_NPIX + 512) : nullptr;
    g.vs0 = ws.d_vs0 ? ws.d_vs0 + (size_t)pair_off * 5 * S * 8 : nullptr;
    g.stats = ws.d_stats + (size_t)pair_off * 2;
    g.flags = ws.d_fbflags ? ws.d_fbflags + pair_off : nullptr;
    g.pairdiff = ws.d_pairdiff ? ws.d_pairdiff + (size_t)pair_off * kPairDiffTiles : nullptr;
    g.flow_il = ws.d_flow_il ? ws.d_flow_il + (size_t)pair_off * AVD_NPIX * 2 : nullptr;
    return g;
}

template <typename... A>
inline void launch1d(void (*k)(A...), int64_t items, int block, hipStream_t s, A... args)
{
    const int grid = (int)((items + block - 1) / block);
    if (grid > 0) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, s, args...);
}

// Gaussian pyramid + polynomial expansion of n frames at all four scales: two launches
void pyramid_and_polyexp(avd_ctx* ctx, const Seg& g, const uint8_t* d_small, int n)
{
    const FbConsts* C = (const FbConsts*)ctx->d_fbc;
    // ctx->fb_fold_blur (default 1): the 320-px scale's 3 x 3 blur is formed inside the polynomial expansion; the pyramid kernel then has no
    // 320-px tiles and pyr[0] is not written (avd_debug_copy "pyr0" is meaningful with the option off only)
    const bool fold = ctx->fb_fold_blur != 0;
    const int wgs = n * ((fold ? 0 : PyrGeo<0>::TILES) + PyrGeo<1>::TILES + PyrGeo<2>::TILES + PyrGeo<3>::TILES);
    kmark(ctx, AVD_K_PYRAMID);
    hipLaunchKernelGGL(k_pyramid_all, dim3(wgs), dim3(256), 0, g.stream, d_small, n, C, fold ? (float*)nullptr : g.pyr_w[0], g.pyr_w[1], g.pyr_w[2],
                       g.pyr_w[3], g.flags, g.pairdiff);
    PolyPtrs P;
    P.small = fold ? d_small : nullptr;
    int grid = ((n * (S / kPolyRows) + 7) >> 3) << 3;      // the 320-px scale: kPolyRows rows per workgroup
    for (int k = 0; k < AVD_FB_LEVELS; k++) {
        P.I[k] = g.pyr[k]; P.R[k] = g.poly[k];
        if (k > 0) grid += ((n * (S >> (2 * k)) + 7) >> 3) << 3;
    }
    kmark(ctx, AVD_K_POLYEXP);
    hipLaunchKernelGGL(k_polyexp_all, dim3(grid), dim3(320), 0, g.stream, P, n, C);
}

// one FarnebackUpdateFlow_Blur iteration at level k: matrices from the current flow, box sums, solve
// plist (may be null): the launches work on the pairs plist[0 .. np) of the segment; the double intermediate is indexed by position in the list
template <int W>
void blur_iteration(const Seg& g, int k, int np, float* flow, const int* plist)
{
    constexpr int NSTRIP = (W + kStripW - 1) / kStripW;
    // AVD_UV_VARIANT (A/B knob): 0 = k_uv everywhere, 1 = k_uvp (4 producers) everywhere, 2 (default) = k_uvp
    // below 320x320 (latency / issue bound levels) and, at 320x320, k_uv for long clips (every design measured there
    // with 119 pairs lands at ~290-300 us: HBM read/write mix; k_uvp with 2 / 3 / 4 producers 293 / 377 / 340 us)
    // and k_uvp for clips short enough to be resident in one round
    static const int variant = [] { const char* e = std::getenv("AVD_UV_VARIANT"); return e ? std::atoi(e) : 2; }();
    // profiling: HIP events around the two full-resolution kernels (avd_stage_ms 4 and 5)
    auto mark = [&](void) {
        if (W == S && g.prof && g.prof->kern_ev_used < 12) (void)hipEventRecord(g.prof->kern_ev[g.prof->kern_ev_used++], g.stream);
    };
    mark();
    const int grid = 8 * ((np + 7) / 8) * NSTRIP;        // (XCD, pair-in-XCD, strip); pairs >= np exit at once
    // at 320x320 the producer / consumer form wins as long as all its workgroups are resident at once (3 per CU,
    // 50 KiB of LDS each): ~170 us per launch instead of ~290 us; beyond that it needs a second residency round
    const bool uvp_fits = np * NSTRIP <= 3 * 256;
    bool latency_shape = false;
    if constexpr (W >= S / 2) {
        // the exact re-run of a few flagged pairs (160 / 320 px): the latency shape (twelve producers: a third of the phases), one workgroup per CU
        if (plist && np * NSTRIP <= 256) {
            latency_shape = true;
            hipLaunchKernelGGL((k_uvp<W, 12>), dim3(grid), dim3(64 * 14), 0, g.stream, (const float*)g.poly[k],
                               (const float*)flow, g.vs, g.vs0, np, plist);
        }
    }
    if (latency_shape) {
    } else if (variant == 1 || (variant == 2 && (W < S || uvp_fits))) {
        hipLaunchKernelGGL((k_uvp<W, 4>), dim3(grid), dim3(384), 0, g.stream, (const float*)g.poly[k],
                           (const float*)flow, g.vs, g.vs0, np, plist);
    } else {
        hipLaunchKernelGGL(k_uv<W>, dim3(grid), dim3(128), 0, g.stream, (const float*)g.poly[k],
                           (const float*)flow, g.vs, g.vs0, np, plist);
    }
    mark(); mark();
    if constexpr (W >= S / 2) {
        if (latency_shape) {
            hipLaunchKernelGGL(k_hscan_lat<W>, dim3(np * d16_nyb(W)), dim3(384), 0, g.stream, (const double*)g.vs,
                               (const double*)g.vs0, flow, np, plist);
            mark();
            return;
        }
    }
    hipLaunchKernelGGL(k_hscan<W>, dim3(np * d16_nyb(W)), dim3(128), 0, g.stream, (const double*)g.vs,
                       (const double*)g.vs0, flow, np, plist);
    mark();
}

void flow_up_level(hipStream_t stream, int k, const float* prev, float* flow, int np, const int* plist)
{
    const int w = S >> k;
    const int items = np * 2 * w * (w / 4);
    if (k == 2) hipLaunchKernelGGL(k_flow_up<S / 4>, dim3((items + 255) / 256), dim3(256), 0, stream, prev, flow, np, plist);
    else if (k == 1) hipLaunchKernelGGL(k_flow_up<S / 2>, dim3((items + 255) / 256), dim3(256), 0, stream, prev, flow, np, plist);
    else hipLaunchKernelGGL(k_flow_up<S>, dim3((items + 255) / 256), dim3(256), 0, stream, prev, flow, np, plist);
}

// the three blur iterations of level k with the EXACT kernels (literal running sums in both directions: bit-identical to the oracle), in place in
// `flow`: the fused kernel (avd_fbfused.hip: one workgroup per pair, all three iterations in one launch) where bit k of fused_mask is set, else the
// two-kernel path.  plist: see blur_iteration.
int exact_level(avd_ctx* ctx, const Seg& g, int k, int np, float* flow, int fused_mask, const int* plist)
{
    const int w = S >> k;
    const bool coarsest = k == AVD_FB_LEVELS - 1;
    auto ev = [&] { if (k == 0 && g.prof && g.prof->kern_ev_used < 12) (void)hipEventRecord(g.prof->kern_ev[g.prof->kern_ev_used++], g.stream); };
    if ((fused_mask >> k) & 1) {
        ev();
        if (int e = launch_fb_level(ctx, g.stream, w, g.poly[k], flow, np, 3, coarsest, plist)) return e;
        ev();
        return 0;
    }
    if (!g.vs) { ctx->err = "two-kernel Farneback path: scratch not reserved"; return AVD_ERR_ARG; }
    for (int it = 0; it < 3; it++) {
        switch (k) {
        case 3: blur_iteration<S / 8>(g, k, np, flow, plist); break;
        case 2: blur_iteration<S / 4>(g, k, np, flow, plist); break;
        case 1: blur_iteration<S / 2>(g, k, np, flow, plist); break;
        default: blur_iteration<S>(g, k, np, flow, plist); break;
        }
    }
    return 0;
}

}  // namespace

// All pairs (f, f+1), f in [0, n-1), of n resident 320x320 frames, on one stream, using the workspace
// slice (frame_off, pair_off) of the current chunk.
int launch_farneback(avd_ctx* ctx, hipStream_t stream, const uint8_t* d_small, int n, int frame_off, int pair_off)
{
    if (n < 2) return 0;
    const Seg g = make_seg(ctx, stream, frame_off, pair_off);
    if (g.prof) ctx->kern_ev_used = 0;
    const int np = n - 1;
    ctx->ws.mag_valid = ctx->fb_mode == 1;
    pyramid_and_polyexp(ctx, g, d_small, n);
    for (int k = AVD_FB_LEVELS - 1; k >= 0; k--) {
        const int w = S >> k, h = S >> k;
        const int64_t plane = (int64_t)w * h;
        const bool fast = ctx->fb_mode == 1;
        // fast mode, what is folded into the level's launches (ctx->fb_fold_up, bit mask; no effect on results): bit 0 the 320-px level's
        // first launch resizes the previous flow itself (chain wave); bit 1 the 160- and 80-px levels' first launch does it in a prologue;
        // bit 2 the 80- and 40-px levels run their three iterations in ONE launch (a pair is one workgroup there)
        const bool fold_chain = fast && k == 0 && (ctx->fb_fold_up & 1);
        const bool fold_pro = fast && (k == 1 || k == 2) && (ctx->fb_fold_up & 2);
        const bool one_launch = fast && (k == 2 || k == 3) && (ctx->fb_fold_up & 4);
        if (k == AVD_FB_LEVELS - 1) {
            // the coarsest level starts from zero flow: the fused kernel is told so, the two-kernel path reads a cleared buffer
            if (!fast && !((ctx->fb_fused >> k) & 1)) HIP_TRY(ctx, hipMemsetAsync(g.flow[k], 0, sizeof(float) * 2 * plane * np, stream));
        } else if (fold_chain || fold_pro) {
            // the level's first launch forms its initial flow from the previous level's (avd_fbfast.hip)
        } else {
            kmark(ctx, k == 2 ? AVD_K_FLOWUP80 : (k == 1 ? AVD_K_FLOWUP160 : AVD_K_FLOWUP320));
            flow_up_level(stream, k, ctx->ws.flow_res[k + 1], g.flow[k], np, nullptr);
        }
        ctx->ws.flow_res[k] = g.flow[k];
        kmark(ctx, k == 3 ? AVD_K_LEVEL40 : (k == 2 ? AVD_K_LEVEL80 : (k == 1 ? AVD_K_LEVEL160 : AVD_K_LEVEL320)));
        if (fast) {
            // fast level kernel (avd_fbfast.hip): the flow ping-pongs between the level's two buffers, a = initial flow, results b, a, b
            float* a = g.flow[k];
            float* b = ctx->ws.d_flow2[k] + (size_t)pair_off * 2 * plane;
            int* fl = ctx->fb_rerun ? g.flags : nullptr;
            const int* pd = ctx->fb_rerun ? g.pairdiff : nullptr;
            if (k == 0 && g.prof && g.prof->kern_ev_used < 12) (void)hipEventRecord(g.prof->kern_ev[g.prof->kern_ev_used++], stream);
            const bool prev_in = fold_chain || fold_pro;                     // the first launch reads the previous level's flow
            const float* prev = prev_in ? ctx->ws.flow_res[k + 1] : nullptr;
            if (one_launch) {
                // all three iterations in one launch: (prologue: prev -> a,) a -> b -> a -> b
                if (int e = launch_fb_fast(ctx, stream, w, g.poly[k], prev_in ? prev : a, b, a, nullptr, fl, pd, np, k == AVD_FB_LEVELS - 1, prev_in ? 4 : 3)) return e;
                a = b;
            } else {
                for (int it = 0; it < 3; it++) {
                    float* mag = (k == 0 && it == 2) ? ctx->ws.d_mag + (size_t)pair_off * AVD_NPIX : nullptr;
                    const bool first_prev = prev_in && it == 0;
                    const int mode = !first_prev ? 0 : (fold_chain ? 1 : 2);
                    if (int e = launch_fb_fast(ctx, stream, w, g.poly[k], first_prev ? prev : a, b, a, mag, fl, pd, np, k == AVD_FB_LEVELS - 1 && it == 0, mode)) return e;
                    float* t = a; a = b; b = t;
                }
            }
            if (k == 0 && g.prof && g.prof->kern_ev_used < 12) (void)hipEventRecord(g.prof->kern_ev[g.prof->kern_ev_used++], stream);
            ctx->ws.flow_res[k] = a;                       // the buffer written last
            // (pairs the level kernels flagged as ill-posed are re-run by the exact kernels once the HOST has seen the flags: launch_farneback_rerun)
            continue;
        }
        // ctx->fb_fused (AVD_FB_FUSED / avd_set_option "fb_fused"): bit k set = level k runs the fused kernel (avd_fbfused.hip: all three iterations in one launch,
        // D never leaves the chip); clear = the two-kernel path
        if (int e = exact_level(ctx, g, k, np, g.flow[k], ctx->fb_fused, nullptr)) return e;
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// Exact re-run (fast mode) of the m pairs h_list[0 .. m) of the chunk the workspace still holds (polynomial expansions of all its frames): the four
// levels with the exact kernels from a compacted pair list, then |flow| and the two statistics of those pairs, where the record kernel reads
// them.  The HOST calls this after it has seen the chunk's flag words (avd_capi.hip): nothing is launched for a chunk without flagged pairs, and the
// work is sized by their number -- up to kRerunTwoKernelMax pairs the 160- / 320-px levels run the two-kernel path (a pair spread over seven strips /
// 64-row bands: ~0.15 ms per level for one pair, where the fused kernel's single workgroup per pair takes 0.23 / 0.83 ms however few pairs there are),
// beyond that the fused kernels (one workgroup per pair = the exact mode's own launches, its time for a clip of nothing but flagged pairs).
// ctx->fb_rerun_fused (tuning / tests): level mask of the fused kernel for the few-pairs case, default 0xC (40 and 80 px).
int launch_farneback_rerun(avd_ctx* ctx, hipStream_t stream, const int* h_list, int m, int pair_off, int np_chunk)
{
    if (m <= 0) return 0;
    Workspace& ws = ctx->ws;
    if (m > ws.fb_cap) { ctx->err = "re-run list longer than the chunk"; return AVD_ERR_ARG; }
    if (!ws.d_rlist) {
        if (int e = dev_alloc(ctx, ws.d_rlist, (size_t)ws.fb_cap)) return e;
        ws.rlist_cap = ws.fb_cap;
    }
    const bool few = m <= kRerunTwoKernelMax;
    const int fused_mask = few ? (ctx->fb_rerun_fused & 0xF) : 0xF;
    if (fused_mask != 0xF && !ws.d_vs_rerun) {
        if (int e = dev_alloc(ctx, ws.d_vs0_rerun, (size_t)kRerunTwoKernelMax * 5 * S * 8)) return e;
        if (int e = dev_alloc(ctx, ws.d_vs_rerun, (size_t)kRerunTwoKernelMax * (5 * AVD_NPIX + 512))) return e;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ws.d_rlist, h_list, sizeof(int) * m, hipMemcpyHostToDevice, stream));
    Seg g = make_seg(ctx, stream, 0, pair_off);
    g.prof = nullptr;
    g.vs = ws.d_vs_rerun; g.vs0 = ws.d_vs0_rerun;         // indexed by position in the list
    const int* plist = ws.d_rlist;
    kmark(ctx, AVD_K_RERUN);
    for (int k = AVD_FB_LEVELS - 1; k >= 0; k--) {
        // levels 3 .. 1 work in the level's first flow buffer, level 0 in the buffer the fast kernels left their final flow in (the one the
        // caller may read back)
        float* flow = k == 0 ? const_cast<float*>(ws.flow_res[0]) : g.flow[k];
        if (k == AVD_FB_LEVELS - 1) {
            if (!((fused_mask >> k) & 1)) { ctx->err = "exact re-run: the coarsest level runs the fused kernel (it needs no cleared flow)"; return AVD_ERR_ARG; }
        } else {
            flow_up_level(stream, k, g.flow[k + 1], flow, m, plist);
        }
        if (int e = exact_level(ctx, g, k, m, flow, fused_mask, plist)) return e;
    }
    float* mg = ws.d_mag + (size_t)pair_off * AVD_NPIX;
    launch1d(k_mag, (int64_t)m * (AVD_NPIX / 4), 256, stream, (const float*)ws.flow_res[0], mg, (int64_t)m * (AVD_NPIX / 4), plist);
    hipLaunchKernelGGL(k_stats_pair, dim3(m), dim3(512), 0, stream, (const float*)mg, g.stats, plist);
    if (g.flow_il)                                        // the caller wants the dense flow (tests / debugging): interleave the chunk again
        launch1d(k_flow_interleave, (int64_t)np_chunk * AVD_NPIX, 256, stream, (const float*)ws.flow_res[0], g.flow_il, (int64_t)np_chunk * AVD_NPIX);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int launch_flow_stats(avd_ctx* ctx, hipStream_t stream, int n, int frame_off, int pair_off)
{
    if (n < 2) return 0;
    const Seg g = make_seg(ctx, stream, frame_off, pair_off);
    const int np = n - 1;
    const float* fl = ctx->ws.flow_res[0] ? ctx->ws.flow_res[0] : g.flow[0];
    float* mg = ctx->ws.d_mag + (size_t)pair_off * AVD_NPIX;
    kmark(ctx, AVD_K_STATS);
    if (!ctx->ws.mag_valid) launch1d(k_mag, (int64_t)np * (AVD_NPIX / 4), 256, stream, fl, mg, (int64_t)np * (AVD_NPIX / 4), (const int*)nullptr);
    hipLaunchKernelGGL(k_stats_pair, dim3(np), dim3(512), 0, stream, (const float*)mg, g.stats, (const int*)nullptr);
    if (g.flow_il)
        launch1d(k_flow_interleave, (int64_t)np * AVD_NPIX, 256, stream, fl, g.flow_il, (int64_t)np * AVD_NPIX);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
