// avd_preprocess.hip -- fused full-resolution pass over decoded BGR frames (gfx950).
//
// One read of each BGR frame from HBM produces everything the reference derives
// from full-resolution pixels (reference app/analyzers/video.py):
//   :5,:43,:51  cv2.cvtColor(BGR2GRAY) x3   -> gray lives only in LDS
//   :6          cv2.resize(32x32, INTER_AREA) -> per-row partial sums (float, cv2's order)
//   :43         cv2.resize(320x320) INTER_LINEAR (11-bit fixed point)
//   :52         cv2.Laplacian(CV_64F).var()  -> exact int64 sum / sum of squares
// HBM-bound: algorithmic traffic = one frame read + 102 400 B small image + 16 B moments
// + the 32 floats/row area partials (DESIGN.md "preprocess kernel").
//
// Work decomposition: a workgroup owns a band of `rows_per_band` full-width rows of one
// frame.  Phase 1 streams the band (+1 halo row above/below) through the 15-bit
// fixed-point gray conversion into an LDS tile of uint8; phases 2-4 read only LDS.
// Consecutive bands of a frame are placed on the same XCD (blockIdx remap) so the
// halo rows hit that XCD's L2.
#include "avd_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPad = 16;     // bytes of padding left of pixel 0 in every LDS tile row

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// 15-bit fixed-point luma, split into byte coefficients for v_dot4_u32_u8:
//   3735 = 14*256+151 (B)   19235 = 75*256+35 (G)   9798 = 38*256+70 (R)
constexpr unsigned kHi = 14u | (75u << 8) | (38u << 16);
constexpr unsigned kLo = 151u | (35u << 8) | (70u << 16);

__device__ __forceinline__ unsigned gray_of(unsigned px, unsigned hi, unsigned lo)
{
    unsigned h = __builtin_amdgcn_udot4(px, hi, 0u, false);
    unsigned l = __builtin_amdgcn_udot4(px, lo, 1u << 14, false);
    return ((h << 8) + l) >> 15;
}

// 12 bytes (4 BGR pixels) -> 4 gray bytes packed little-endian
__device__ __forceinline__ unsigned gray4(unsigned w0, unsigned w1, unsigned w2)
{
    unsigned p1 = __builtin_amdgcn_alignbyte(w1, w0, 3);
    unsigned p2 = __builtin_amdgcn_alignbyte(w2, w1, 2);
    unsigned g0 = gray_of(w0, kHi, kLo);
    unsigned g1 = gray_of(p1, kHi, kLo);
    unsigned g2 = gray_of(p2, kHi, kLo);
    unsigned g3 = gray_of(w2, kHi << 8, kLo << 8);
    return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}

__device__ __forceinline__ unsigned gray1(const uint8_t* p)
{
    return (p[0] * 3735u + p[1] * 19235u + p[2] * 9798u + (1u << 14)) >> 15;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// blockIdx -> (frame, band): logical ids that are consecutive share blockIdx % 8, i.e. an XCD.
__device__ __forceinline__ int xcd_remap(int bid, int total)
{
    const int per = (total + 7) >> 3;
    const int lid = (bid & 7) * per + (bid >> 3);
    return lid;      // may be >= total (idle tail block)
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_preprocess(const uint8_t* __restrict__ bgr, int n,
                                                        PreParams P, uint8_t* __restrict__ small,
                                                        float* __restrict__ rowbuf,
                                                        unsigned long long* __restrict__ lap_acc)
{
    extern __shared__ __align__(16) uint8_t tile[];
    __shared__ long long red[2][kThreads / 64];

    const int total = n * P.nbands;
    const int lid = xcd_remap(blockIdx.x, total);
    if (lid >= total) return;
    const int f = lid / P.nbands;
    const int band = lid - f * P.nbands;
    const int h = P.h, w = P.w, pitch = P.pitch;
    const int r0 = band * P.rows_per_band;
    const int rows = min(P.rows_per_band, h - r0);
    const int tid = threadIdx.x;
    const uint8_t* frame = bgr + (int64_t)f * P.frame_stride;

    // ---- phase 1: BGR -> gray tile rows [r0-1, r0+rows] (tile row 0 = r0-1) ----------
    const int trows = rows + 2;
    if (VEC) {
        const int chunks = w >> 4;                       // 16 pixels = 48 bytes per chunk
        for (int it = tid; it < trows * chunks; it += kThreads) {
            const int tr = it / chunks, c = it - tr * chunks;
            const int y = reflect101(r0 - 1 + tr, h);
            const uint4* src = reinterpret_cast<const uint4*>(frame + (int64_t)y * P.row_stride + c * 48);
            const uint4 a = src[0], b = src[1], d = src[2];
            uint4 g;
            g.x = gray4(a.x, a.y, a.z);
            g.y = gray4(a.w, b.x, b.y);
            g.z = gray4(b.z, b.w, d.x);
            g.w = gray4(d.y, d.z, d.w);
            *reinterpret_cast<uint4*>(tile + tr * pitch + kPad + c * 16) = g;
        }
    } else {
        for (int it = tid; it < trows * w; it += kThreads) {
            const int tr = it / w, x = it - tr * w;
            const int y = reflect101(r0 - 1 + tr, h);
            tile[tr * pitch + kPad + x] = (uint8_t)gray1(frame + (int64_t)y * P.row_stride + x * 3);
        }
    }
    __syncthreads();
    // column halo (BORDER_REFLECT_101): pixel -1 := pixel 1, pixel w := pixel w-2
    for (int tr = tid; tr < trows; tr += kThreads) {
        uint8_t* row = tile + tr * pitch + kPad;
        row[-1] = row[reflect101(-1, w)];
        row[w] = row[reflect101(w, w)];
    }
    __syncthreads();

    // ---- phase 2: Laplacian moments over rows [r0, r0+rows) ----------------------------
    // 4 pixels per step as two packed 16-bit lanes (even / odd bytes); lap+1020 stays positive.
    int s_acc = 0;
    unsigned q_acc = 0;
    {
        const int quads = (w + 3) >> 2;
        for (int it = tid; it < rows * quads; it += kThreads) {
            const int r = it / quads, qx = it - r * quads;
            const uint8_t* c_row = tile + (r + 1) * pitch + kPad + qx * 4;
            const unsigned C = *reinterpret_cast<const unsigned*>(c_row);
            const unsigned Lw = *reinterpret_cast<const unsigned*>(c_row - 4);
            const unsigned Rw = *reinterpret_cast<const unsigned*>(c_row + 4);
            const unsigned U = *reinterpret_cast<const unsigned*>(c_row - pitch);
            const unsigned D = *reinterpret_cast<const unsigned*>(c_row + pitch);
            const unsigned L = __builtin_amdgcn_alignbyte(C, Lw, 3);
            const unsigned R = __builtin_amdgcn_alignbyte(Rw, C, 1);
            const unsigned NC = ~C;                       // 255 - c per byte
            const unsigned m = 0x00FF00FFu;
            // even bytes (pixels 0,2) and odd bytes (pixels 1,3): u+d+l+r + 4*(255-c) = lap + 1020
            unsigned te = (U & m) + (D & m) + (L & m) + (R & m) + ((NC & m) << 2);
            unsigned to = ((U >> 8) & m) + ((D >> 8) & m) + ((L >> 8) & m) + ((R >> 8) & m) + (((NC >> 8) & m) << 2);
            int l0 = (int)(te & 0xFFFF) - 1020, l2 = (int)(te >> 16) - 1020;
            int l1 = (int)(to & 0xFFFF) - 1020, l3 = (int)(to >> 16) - 1020;
            const int valid = w - qx * 4;                 // >= 4 except in a ragged last quad
            if (valid < 4) {
                if (valid < 2) l1 = 0;
                if (valid < 3) l2 = 0;
                l3 = 0;
            }
            s_acc += l0 + l1 + l2 + l3;
            q_acc += (unsigned)(l0 * l0) + (unsigned)(l1 * l1) + (unsigned)(l2 * l2) + (unsigned)(l3 * l3);
        }
    }

    // ---- phase 3: INTER_AREA horizontal partials, one float chain per (row, cell) ------
    {
        float* out = rowbuf + ((int64_t)f * h + r0) * AVD_HASH;
        for (int it = tid; it < rows * AVD_HASH; it += kThreads) {
            const int r = it >> 5, dx = it & 31;
            const uint8_t* src = tile + (r + 1) * pitch + kPad + P.ax_begin[dx];
            const int cnt = P.ax_count[dx];
            if (P.area_fast) {
                int acc = 0;
                for (int k = 0; k < cnt; k++) acc += src[k];
                out[it] = __int_as_float(acc);
            } else {
                const float wf = P.ax_first[dx], wm = P.ax_mid[dx], wl = P.ax_last[dx];
                float acc = 0.f;
                for (int k = 0; k < cnt; k++) {
                    const float wgt = k == 0 ? wf : (k == cnt - 1 ? wl : wm);
                    acc = __fadd_rn(acc, __fmul_rn((float)src[k], wgt));
                }
                out[it] = acc;
            }
        }
    }

    // ---- phase 4: INTER_LINEAR 320x320 rows whose upper source row lies in this band ---
    {
        const int d0 = P.band_dy[band], d1 = P.band_dy[band + 1];
        uint8_t* dst = small + (int64_t)f * AVD_NPIX;
        for (int it = tid; it < (d1 - d0) * AVD_SMALL; it += kThreads) {
            const int dy = d0 + it / AVD_SMALL, dx = it % AVD_SMALL;
            const uint8_t* ra = tile + (P.ly0[dy] - r0 + 1) * pitch + kPad;
            const uint8_t* rb = tile + (P.ly1[dy] - r0 + 1) * pitch + kPad;
            const int x0 = P.lx0[dx], x1 = P.lx1[dx];
            const int a0 = P.la0[dx], a1 = P.la1[dx], b0 = P.lb0[dy], b1 = P.lb1[dy];
            const int ha = ra[x0] * a0 + ra[x1] * a1;
            const int hb = rb[x0] * a0 + rb[x1] * a1;
            dst[dy * AVD_SMALL + dx] = (uint8_t)((((b0 * (ha >> 4)) >> 16) + ((b1 * (hb >> 4)) >> 16) + 2) >> 2);
        }
    }

    // ---- block reduction of the Laplacian moments, one 64-bit atomic pair per band -----
    long long s64 = wave_sum((long long)s_acc);
    long long q64 = wave_sum((long long)q_acc);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s64; red[1][tid >> 6] = q64; }
    __syncthreads();
    if (tid == 0) {
        long long s = 0, q = 0;
        for (int i = 0; i < kThreads / 64; i++) { s += red[0][i]; q += red[1][i]; }
        atomicAdd(&lap_acc[2 * f], (unsigned long long)s);
        atomicAdd(&lap_acc[2 * f + 1], (unsigned long long)q);
    }
}

// 32x32 INTER_AREA cells from the per-row partials (vertical accumulation in cv2's row
// order), then aHash bits: g >= mean(g)  <=>  1024*g >= sum(g)   (video.py:7-8)
__global__ __launch_bounds__(1024) void k_hash(const float* __restrict__ rowbuf, HashParams P,
                                              uint8_t* __restrict__ area, uint8_t* __restrict__ bits)
{
    __shared__ int wsum[16];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int dy = tid >> 5, dx = tid & 31;
    const float* rb = rowbuf + (int64_t)f * P.h * AVD_HASH + dx;
    const int y0 = P.ay_begin[dy], cnt = P.ay_count[dy];
    int cell;
    if (P.area_fast) {
        int acc = 0;
        for (int k = 0; k < cnt; k++) acc += __float_as_int(rb[(int64_t)(y0 + k) * AVD_HASH]);
        if (dx < P.fast_simd_w) cell = (acc + 2) >> 2;
        else {
            const float scale = 1.f / (float)P.fast_area;
            cell = (int)rintf(__fmul_rn((float)acc, scale));
        }
    } else {
        const float wf = P.ay_first[dy], wm = P.ay_mid[dy], wl = P.ay_last[dy];
        float acc = 0.f;
        for (int k = 0; k < cnt; k++) {
            const float beta = k == 0 ? wf : (k == cnt - 1 ? wl : wm);
            const float t = __fmul_rn(beta, rb[(int64_t)(y0 + k) * AVD_HASH]);
            acc = k == 0 ? t : __fadd_rn(acc, t);
        }
        cell = (int)rintf(acc);             // saturate_cast<uchar>: round-half-even, clamp
    }
    cell = min(255, max(0, cell));
    int s = wave_sum(cell);
    if ((tid & 63) == 0) wsum[tid >> 6] = s;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) total += wsum[i];
    area[(int64_t)f * 1024 + tid] = (uint8_t)cell;
    bits[(int64_t)f * 1024 + tid] = (uint8_t)(cell * 1024 >= total);
}

// ham[f] = popcount(hash[f] ^ hash[f-1]) (video.py:38); ham[0] = -1
__global__ __launch_bounds__(256) void k_hamming(const uint8_t* __restrict__ bits, int* __restrict__ ham)
{
    __shared__ int wsum[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    if (f == 0) { if (tid == 0) ham[0] = -1; return; }
    const unsigned a = reinterpret_cast<const unsigned*>(bits + (int64_t)f * 1024)[tid];
    const unsigned b = reinterpret_cast<const unsigned*>(bits + (int64_t)(f - 1) * 1024)[tid];
    int c = __popc(a ^ b);
    c = wave_sum(c);
    if ((tid & 63) == 0) wsum[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) ham[f] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

}  // namespace

int launch_preprocess(avd_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w,
                      int64_t row_stride, int64_t frame_stride)
{
    Workspace& ws = ctx->ws;
    PreParams P = ws.pre;
    P.row_stride = row_stride;
    P.frame_stride = frame_stride;
    HIP_TRY(ctx, hipMemsetAsync(ws.d_lap, 0, sizeof(unsigned long long) * 2 * n, ctx->stream));
    const int total = n * P.nbands;
    const int grid = ((total + 7) / 8) * 8;
    const size_t lds = (size_t)(P.rows_per_band + 2) * P.pitch;
    const bool vec = (w % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(d_bgr) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(k_preprocess<true>, dim3(grid), dim3(kThreads), lds, ctx->stream,
                           d_bgr, n, P, ws.d_small, ws.d_rowbuf, ws.d_lap);
    else
        hipLaunchKernelGGL(k_preprocess<false>, dim3(grid), dim3(kThreads), lds, ctx->stream,
                           d_bgr, n, P, ws.d_small, ws.d_rowbuf, ws.d_lap);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int launch_hash(avd_ctx* ctx, int n)
{
    Workspace& ws = ctx->ws;
    hipLaunchKernelGGL(k_hash, dim3(n), dim3(1024), 0, ctx->stream, ws.d_rowbuf, ws.hsh, ws.d_area, ws.d_hash);
    hipLaunchKernelGGL(k_hamming, dim3(n), dim3(256), 0, ctx->stream, ws.d_hash, ws.d_ham);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
