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
//
// Measured design notes (profiles/r01_preprocess_ablation.md): byte ops (v_dot4_u32_u8,
// v_alignbyte, v_perm) issue at half rate on gfx950, the kernel is balanced between the load
// path (0.13 ms alone) and the arithmetic (0.14 ms alone); persistent software-pipelined
// variants (next band's loads in flight in registers during the arithmetic, 256 or 512 threads)
// were built and were 30-40 % SLOWER than many small independent workgroups (4 per CU): the
// dependent float chains of INTER_AREA want thread-level parallelism more than prefetch.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "avd_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPad = 16;     // bytes of padding left of pixel 0 in every LDS tile row
constexpr int kLapSlots = 8; // per-band slots for per-wave Laplacian partial moments (<= 8 waves/workgroup)

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// 15-bit fixed-point luma, split into byte coefficients for v_dot4_u32_u8:
//   3735 = 14*256+151 (B)   19235 = 75*256+35 (G)   9798 = 38*256+70 (R)
// gray = (256 h + l) >> 15 with h = px . hi, l = px . lo + 2^14.  Since 256 h + l = 256 (h + (l >> 8)) + (l & 255),
// gray = (h + (l >> 8)) >> 7 = byte 1 of 2 h + 2 (l >> 8); 2 (l >> 8) and l >> 7 differ in bit 0 only and 2 h is even,
// so byte 1 of  v = px . (2 hi) + (l >> 7)  is the gray value: dot4, shift, dot4 -- and the four results of a
// quad are packed by two v_perm picking byte 1 of each.
constexpr unsigned kHi2 = 28u | (150u << 8) | (76u << 16);
constexpr unsigned kLo = 151u | (35u << 8) | (70u << 16);

__device__ __forceinline__ unsigned gray_v(unsigned px, unsigned hi2, unsigned lo)
{
    const unsigned l = __builtin_amdgcn_udot4(px, lo, 1u << 14, false);
    return __builtin_amdgcn_udot4(px, hi2, l >> 7, false);       // gray in bits 8..15
}

// 12 bytes (4 BGR pixels) -> 4 gray bytes packed little-endian
__device__ __forceinline__ unsigned gray4(unsigned w0, unsigned w1, unsigned w2)
{
    const unsigned p1 = __builtin_amdgcn_alignbyte(w1, w0, 3);
    const unsigned p2 = __builtin_amdgcn_alignbyte(w2, w1, 2);
    const unsigned v0 = gray_v(w0, kHi2, kLo);
    const unsigned v1 = gray_v(p1, kHi2, kLo);
    const unsigned v2 = gray_v(p2, kHi2, kLo);
    const unsigned v3 = gray_v(w2, kHi2 << 8, kLo << 8);
    // v_perm_b32(hi, lo, sel): byte k of the result = byte sel[k] of {hi:lo} (0..3 = lo, 4..7 = hi)
    const unsigned g01 = __builtin_amdgcn_perm(v1, v0, 0x0c0c0501u);     // [v0.b1, v1.b1, 0, 0]
    const unsigned g23 = __builtin_amdgcn_perm(v3, v2, 0x05010c0cu);     // [0, 0, v2.b1, v3.b1]
    return g01 | g23;
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

__device__ __forceinline__ int reflect_once(int p, int len)      // valid for -len < p < 2*len-1
{
    p = p < 0 ? -p : p;
    return p >= len ? 2 * len - 2 - p : p;
}

// 16 BGR pixels (three 16-byte words) -> 16 gray bytes
__device__ __forceinline__ uint4 gray16(const uint4& a, const uint4& b, const uint4& d)
{
    uint4 g;
    g.x = gray4(a.x, a.y, a.z);
    g.y = gray4(a.w, b.x, b.y);
    g.z = gray4(b.z, b.w, d.x);
    g.w = gray4(d.y, d.z, d.w);
    return g;
}

struct Moments { long long s, q; };

// Phases 2-4 of a band whose gray rows [r0-1, r0+rows] (incl. column halo) sit in `tile`:
// exact Laplacian moments, INTER_AREA horizontal partials, INTER_LINEAR 320x320 rows.
// NT = workgroup size.  LTAB: the resampling tables were copied to LDS (layout LdsTabs) so that the
// compute phases issue no global LOADS (the pipelined kernel keeps the next band's loads in flight
// here, and vmcnt retires in order).
struct LdsTabs {
    LinTap lxt[AVD_SMALL], lyt[AVD_SMALL];
    int ax_begin[AVD_HASH], ax_count[AVD_HASH];
    float ax_first[AVD_HASH], ax_mid[AVD_HASH], ax_last[AVD_HASH];
};

template <int NT>
__device__ __forceinline__ void fill_lds_tabs(LdsTabs* lt, const PreParams& P, int tid)
{
    const uint2* sx = reinterpret_cast<const uint2*>(P.lxt);
    const uint2* sy = reinterpret_cast<const uint2*>(P.lyt);
    for (int i = tid; i < AVD_SMALL; i += NT) {
        reinterpret_cast<uint2*>(lt->lxt)[i] = sx[i];
        reinterpret_cast<uint2*>(lt->lyt)[i] = sy[i];
    }
    if (tid < AVD_HASH) {
        lt->ax_begin[tid] = P.ax_begin[tid]; lt->ax_count[tid] = P.ax_count[tid];
        lt->ax_first[tid] = P.ax_first[tid]; lt->ax_mid[tid] = P.ax_mid[tid]; lt->ax_last[tid] = P.ax_last[tid];
    }
}

template <int NT, bool LTAB>
__device__ __forceinline__ Moments band_phases(const uint8_t* tile, const LdsTabs* lt, const PreParams& P, int f,
                                               int band, int r0, int rows, int tid, uint8_t* __restrict__ small,
                                               float* __restrict__ rowbuf)
{
    constexpr int kThreads = NT;
    const int w = P.w, h = P.h, pitch = P.pitch;
    // ---- Laplacian moments, exact, from row-pair products -----------------------------------
    // lap(t,x) = T(t-1,x) + T(t+1,x) + T(t,x-1) + T(t,x+1) - 4 T(t,x) on the haloed tile T
    // (tile row t = 1..rows are the band's rows).  Expanding sum lap^2 gives 15 sums of byte
    // products per band row; products that involve the same two tile rows are shared between
    // neighbouring band rows, so it is enough to form, when tile row n "enters" a lane's column
    // walk, seven v_dot4_u32_u8 with the rows above it:
    //   A(n)=C.C  HR(n)=C.R  H2(n)=L.R  V1(n-1)=C'.C  V2(n-2)=C''.C  Dp(n-1)=C'.R  Dm(n-1)=C'.L
    // (C = 4 gray bytes of row n, L/R = the same shifted by -1/+1 pixel, ' = previous row).
    // Row n contributes with weights that depend only on whether n-1, n, n+1 are band rows:
    //   A: [n+1 in B] + [n-1 in B] + 18 [n in B]      V1, Dp, Dm: (-8|+2|+2) ([n in B] + [n-1 in B])
    //   HR: -16 [n in B]   H2: +2 [n in B]   V2: +2 [n-1 in B]    row sum: [n+1 in B]+[n-1 in B]-2[n in B]
    // Interior rows (2 <= n <= rows-1) have constant weights (20,-16,2,-16,2,4,4; row sum 0) and
    // accumulate in place; the <= 4 boundary rows use a multiply-add per product.  What the
    // column shifts miss at x = 0 / w-1 is added per band row as exact scalar edge terms.
    long long s_acc = 0, q_acc = 0;
    if (!(P.dbg_skip & 1)) {
        const int quads = (w + 3) >> 2;
        const int trows = rows + 2;
        const unsigned ones = 0x01010101u;
        unsigned iA = 0, iHR = 0, iH2 = 0, iV1 = 0, iV2 = 0, iDp = 0, iDm = 0;   // products with the interior weights
        int xA = 0;                                                                // A with weight 1 (signed: also corrects 20 -> 19)
        unsigned xV1 = 0, xD = 0;                                                  // V1 with weight -8, Dp + Dm with weight 2
        int bq = 0, bs = 0;                                                        // generic path (rows < 2) and row sums
        // two compiled versions: only widths that are not a multiple of 4 need the mask of the last quad
        auto walk = [&](auto ragged_tag) {
        constexpr bool ragged = decltype(ragged_tag)::value;
        for (int qx = tid; qx < quads; qx += kThreads) {
            const uint8_t* col = tile + kPad + qx * 4;
            unsigned mk = 0xFFFFFFFFu;
            if (ragged) {                                  // ragged last quad: drop pixels >= w
                const int valid = w - qx * 4;
                if (valid < 4) mk = (1u << (8 * valid)) - 1u;
            }
            auto ld = [&](int nrow, unsigned& C, unsigned& L, unsigned& R) {
                const uint8_t* cp = col + nrow * pitch;
                const unsigned Cw = *reinterpret_cast<const unsigned*>(cp);
                const unsigned Lw = *reinterpret_cast<const unsigned*>(cp - 4);
                const unsigned Rw = *reinterpret_cast<const unsigned*>(cp + 4);
                C = Cw; L = __builtin_amdgcn_alignbyte(Cw, Lw, 3); R = __builtin_amdgcn_alignbyte(Rw, Cw, 1);
                if (ragged) { C &= mk; L &= mk; R &= mk; }
            };
            unsigned C, L, R, C1, C2;
            if (rows >= 2) {
                // The four boundary rows are peeled (their weights differ from the interior ones only in a few
                // places, which go to the x* accumulators), so the interior loop is seven dot4 and no branch:
                //   row 0      : A x1, row sum +1
                //   row 1      : as interior but A x19, V1 x-8, no V2, Dp/Dm x2, row sum -1
                //   row rows   : as interior but A x19, row sum -1
                //   row rows+1 : A x1, V1 x-8, V2 x2, Dp/Dm x2, row sum +1
                ld(0, C, L, R);
                xA += (int)__builtin_amdgcn_udot4(C, C, 0u, false);
                bs += (int)__builtin_amdgcn_udot4(C, ones, 0u, false);
                C1 = C;
                ld(1, C, L, R);
                {
                    const unsigned t = __builtin_amdgcn_udot4(C, C, 0u, false);
                    iA += t; xA -= (int)t;
                    iHR = __builtin_amdgcn_udot4(C, R, iHR, false);
                    iH2 = __builtin_amdgcn_udot4(L, R, iH2, false);
                    xV1 = __builtin_amdgcn_udot4(C1, C, xV1, false);
                    xD = __builtin_amdgcn_udot4(C1, R, xD, false);
                    xD = __builtin_amdgcn_udot4(C1, L, xD, false);
                    bs -= (int)__builtin_amdgcn_udot4(C, ones, 0u, false);
                }
                C2 = C1; C1 = C;
                for (int nrow = 2; nrow <= rows - 1; nrow++) {
                    ld(nrow, C, L, R);
                    iA = __builtin_amdgcn_udot4(C, C, iA, false);
                    iHR = __builtin_amdgcn_udot4(C, R, iHR, false);
                    iH2 = __builtin_amdgcn_udot4(L, R, iH2, false);
                    iV1 = __builtin_amdgcn_udot4(C1, C, iV1, false);
                    iV2 = __builtin_amdgcn_udot4(C2, C, iV2, false);
                    iDp = __builtin_amdgcn_udot4(C1, R, iDp, false);
                    iDm = __builtin_amdgcn_udot4(C1, L, iDm, false);
                    C2 = C1; C1 = C;
                }
                ld(rows, C, L, R);
                {
                    const unsigned t = __builtin_amdgcn_udot4(C, C, 0u, false);
                    iA += t; xA -= (int)t;
                    iHR = __builtin_amdgcn_udot4(C, R, iHR, false);
                    iH2 = __builtin_amdgcn_udot4(L, R, iH2, false);
                    iV1 = __builtin_amdgcn_udot4(C1, C, iV1, false);
                    iV2 = __builtin_amdgcn_udot4(C2, C, iV2, false);
                    iDp = __builtin_amdgcn_udot4(C1, R, iDp, false);
                    iDm = __builtin_amdgcn_udot4(C1, L, iDm, false);
                    bs -= (int)__builtin_amdgcn_udot4(C, ones, 0u, false);
                }
                C2 = C1; C1 = C;
                ld(rows + 1, C, L, R);
                xA += (int)__builtin_amdgcn_udot4(C, C, 0u, false);
                xV1 = __builtin_amdgcn_udot4(C1, C, xV1, false);
                iV2 = __builtin_amdgcn_udot4(C2, C, iV2, false);
                xD = __builtin_amdgcn_udot4(C1, R, xD, false);
                xD = __builtin_amdgcn_udot4(C1, L, xD, false);
                bs += (int)__builtin_amdgcn_udot4(C, ones, 0u, false);
            } else {
                // a one-row band (last band of some geometries): every tile row is a boundary row
                C1 = 0; C2 = 0;
                for (int nrow = 0; nrow < trows; nrow++) {
                    ld(nrow, C, L, R);
                    const int in0 = nrow >= 1 && nrow <= rows;          // n   in B
                    const int inm = nrow >= 2 && nrow <= rows + 1;      // n-1 in B
                    const int inp = nrow + 1 <= rows;                   // n+1 in B  (n >= 0 always)
                    const int wA = inp + inm + 18 * in0, wP = in0 + inm;
                    bq += wA * (int)__builtin_amdgcn_udot4(C, C, 0u, false);
                    bq -= 16 * in0 * (int)__builtin_amdgcn_udot4(C, R, 0u, false);
                    bq += 2 * in0 * (int)__builtin_amdgcn_udot4(L, R, 0u, false);
                    bq -= 8 * wP * (int)__builtin_amdgcn_udot4(C1, C, 0u, false);
                    bq += 2 * inm * (int)__builtin_amdgcn_udot4(C2, C, 0u, false);
                    bq += 2 * wP * (int)__builtin_amdgcn_udot4(C1, R, 0u, false);
                    bq += 2 * wP * (int)__builtin_amdgcn_udot4(C1, L, 0u, false);
                    bs += (inp + inm - 2 * in0) * (int)__builtin_amdgcn_udot4(C, ones, 0u, false);
                    C2 = C1; C1 = C;
                }
            }
        }
        };
        if (w & 3) walk(std::true_type{}); else walk(std::false_type{});
        q_acc = 20ll * iA - 16ll * iHR + 2ll * iH2 - 16ll * iV1 + 2ll * iV2 + 4ll * iDp + 4ll * iDm
                + (long long)xA - 8ll * xV1 + 2ll * xD + bq;
        s_acc = bs;
        // edge terms of band row t (tile row t = tid + 1): the l/r shifted sums run over x-1 / x+1
        if (tid < rows) {
            const uint8_t* r = tile + (tid + 1) * pitch + kPad;
            const uint8_t* dn = r + pitch;
            const int a = r[-1], b0 = r[0], e = r[w - 1], z = r[w];      // T(t,-1), T(t,0), T(t,w-1), T(t,w)
            const int da = dn[-1], db = dn[0], de = dn[w - 1], dz = dn[w];
            s_acc += a - e - b0 + z;
            q_acc += (a * a - e * e) + (z * z - b0 * b0)                // l^2 + r^2
                     - 8 * (a * b0 - e * z)                            // m*l
                     + 2 * (a * db - e * dz)                           // d*l
                     + 2 * (z * de - b0 * da);                         // d*r
        }
    }

    // ---- INTER_AREA horizontal partials, one float chain per (row, cell), cv2's order ----
    // The chain of a cell is sequential by definition (float adds in cv2's order); a lane runs the
    // chains of two different rows interleaved so that dependent adds of one hide behind the other.
    if (!(P.dbg_skip & 2)) {
        float* out = rowbuf + ((int64_t)f * h + r0) * AVD_HASH;
        const int dx = tid & 31;
        const int xb = LTAB ? lt->ax_begin[dx] : P.ax_begin[dx];
        const int cnt = LTAB ? lt->ax_count[dx] : P.ax_count[dx];
        const float wf = LTAB ? lt->ax_first[dx] : P.ax_first[dx], wm = LTAB ? lt->ax_mid[dx] : P.ax_mid[dx],
                    wl = LTAB ? lt->ax_last[dx] : P.ax_last[dx];
        constexpr int RSTEP = kThreads / AVD_HASH;             // rows covered per pass
        for (int ra = tid >> 5; ra < rows; ra += 2 * RSTEP) {
            const int rb = ra + RSTEP;
            const bool two = rb < rows;
            const uint8_t* sa = tile + (ra + 1) * pitch + kPad + xb;
            const uint8_t* sb = tile + ((two ? rb : ra) + 1) * pitch + kPad + xb;
            if (P.area_fast) {
                int a0 = 0, a1 = 0;
                for (int k = 0; k < cnt; k++) { a0 += sa[k]; a1 += sb[k]; }
                out[ra * AVD_HASH + dx] = __int_as_float(a0);
                if (two) out[rb * AVD_HASH + dx] = __int_as_float(a1);
            } else if (P.area_x_uniform4) {
                // every cell starts on a 4-byte boundary, spans a multiple of 4 pixels, one weight
                const unsigned* a4 = reinterpret_cast<const unsigned*>(sa);
                const unsigned* b4 = reinterpret_cast<const unsigned*>(sb);
                float a0 = 0.f, a1 = 0.f;
                for (int k = 0; k < (cnt >> 2); k++) {
                    const unsigned va = a4[k], vb = b4[k];
                    a0 = __fadd_rn(a0, __fmul_rn((float)(va & 0xFF), wm));
                    a1 = __fadd_rn(a1, __fmul_rn((float)(vb & 0xFF), wm));
                    a0 = __fadd_rn(a0, __fmul_rn((float)((va >> 8) & 0xFF), wm));
                    a1 = __fadd_rn(a1, __fmul_rn((float)((vb >> 8) & 0xFF), wm));
                    a0 = __fadd_rn(a0, __fmul_rn((float)((va >> 16) & 0xFF), wm));
                    a1 = __fadd_rn(a1, __fmul_rn((float)((vb >> 16) & 0xFF), wm));
                    a0 = __fadd_rn(a0, __fmul_rn((float)(va >> 24), wm));
                    a1 = __fadd_rn(a1, __fmul_rn((float)(vb >> 24), wm));
                }
                out[ra * AVD_HASH + dx] = a0;
                if (two) out[rb * AVD_HASH + dx] = a1;
            } else {
                float a0 = 0.f, a1 = 0.f;
                for (int k = 0; k < cnt; k++) {
                    const float wgt = k == 0 ? wf : (k == cnt - 1 ? wl : wm);
                    a0 = __fadd_rn(a0, __fmul_rn((float)sa[k], wgt));
                    a1 = __fadd_rn(a1, __fmul_rn((float)sb[k], wgt));
                }
                out[ra * AVD_HASH + dx] = a0;
                if (two) out[rb * AVD_HASH + dx] = a1;
            }
        }
    }

    // ---- INTER_LINEAR 320x320 rows whose upper source row lies in this band ----------------
    if (!(P.dbg_skip & 4)) {
        const int d0 = P.band_dy[band], d1 = P.band_dy[band + 1];        // wave-uniform scalar loads
        uint8_t* dst = small + (int64_t)f * AVD_NPIX;
        // a lane owns output COLUMNS (its x taps are unpacked once); the rows of the band are walked with
        // wave-uniform y taps held in scalar registers
        for (int dx = tid; dx < AVD_SMALL; dx += kThreads) {
            const LinTap tx = LTAB ? lt->lxt[dx] : P.lxt[dx];
            const int x0 = tx.i0, x1 = tx.i1, wx0 = tx.w0, wx1 = tx.w1;
            for (int dy = d0; dy < d1; dy++) {
                const uint2 tyw = *reinterpret_cast<const uint2*>(LTAB ? &lt->lyt[dy] : &P.lyt[dy]);
                const unsigned ta = __builtin_amdgcn_readfirstlane(tyw.x), tb = __builtin_amdgcn_readfirstlane(tyw.y);
                const int yi0 = (short)(ta & 0xffffu), yi1 = (short)(ta >> 16);
                const int wy0 = (short)(tb & 0xffffu), wy1 = (short)(tb >> 16);
                const uint8_t* ra = tile + (yi0 - r0 + 1) * pitch + kPad;
                const uint8_t* rb = tile + (yi1 - r0 + 1) * pitch + kPad;
                const int ha = ra[x0] * wx0 + ra[x1] * wx1;
                const int hb = rb[x0] * wx0 + rb[x1] * wx1;
                dst[dy * AVD_SMALL + dx] = (uint8_t)((((wy0 * (ha >> 4)) >> 16) + ((wy1 * (hb >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    Moments m;
    m.s = s_acc;
    m.q = q_acc;
    return m;
}

// Generic kernel: one workgroup per band, any geometry / alignment (scalar loads if needed).
template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_preprocess(const uint8_t* __restrict__ bgr, int n,
                                                        PreParams P, uint8_t* __restrict__ small,
                                                        float* __restrict__ rowbuf,
                                                        long long* __restrict__ lap_part)
{
    extern __shared__ __align__(16) uint8_t tile[];
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
    const int trows = rows + 2;                            // tile row 0 = image row r0-1
    if (VEC) {
        const int chunks = w >> 4;
        for (int it = tid; it < trows * chunks; it += kThreads) {
            const int tr = it / chunks, c = it - tr * chunks;
            const int y = reflect_once(r0 - 1 + tr, h);
            const uint4* src = reinterpret_cast<const uint4*>(frame + (int64_t)y * P.row_stride + c * 48);
            *reinterpret_cast<uint4*>(tile + tr * pitch + kPad + c * 16) = gray16(src[0], src[1], src[2]);
        }
    } else {
        for (int it = tid; it < trows * w; it += kThreads) {
            const int tr = it / w, x = it - tr * w;
            const int y = reflect_once(r0 - 1 + tr, h);
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
    const Moments m = band_phases<kThreads, false>(tile, nullptr, P, f, band, r0, rows, tid, small, rowbuf);
    // per-wave partial moments, summed per frame in k_hash (atomics on the 16 B/frame accumulators
    // serialise in L2: 65 k same-line atomics cost ~50 us per launch)
    const long long s64 = wave_sum(m.s), q64 = wave_sum(m.q);
    if ((tid & 63) == 0) {
        long long* slot = lap_part + ((int64_t)lid * kLapSlots + (tid >> 6)) * 2;
        slot[0] = s64; slot[1] = q64;
    }
}

// Aligned fast path (w % 16 == 0, <= 4096 px): one workgroup per band like the generic kernel, but
// every lane owns one 16-pixel column chunk and issues ALL its NI row loads (3 x 16 B each) before
// the first conversion, so a workgroup has its whole band in flight at once; conversions start as
// the words arrive (vmcnt counts down in issue order).  The lanes that convert the first / last
// chunk also write the reflected halo bytes, which saves a barrier.
template <int NI, int NT>
__global__ __launch_bounds__(NT, 4) void k_preprocess_vec(const uint8_t* __restrict__ bgr, int n,
                                                               PreParams P, uint8_t* __restrict__ small,
                                                               float* __restrict__ rowbuf,
                                                               long long* __restrict__ lap_part)
{
    extern __shared__ __align__(16) uint8_t tile[];
    const int total = n * P.nbands;
    const int lid = xcd_remap(blockIdx.x, total);
    if (lid >= total) return;
    const int f = lid / P.nbands, band = lid - f * P.nbands;
    const int h = P.h, w = P.w, pitch = P.pitch;
    const int r0 = band * P.rows_per_band;
    const int rows = min(P.rows_per_band, h - r0);
    const int trows = rows + 2;
    const int tid = threadIdx.x;
    const int chunks = w >> 4;
    const int rpp = NT / chunks;                     // tile rows covered per pass of the workgroup
    const int rsub = tid / chunks, c = tid - rsub * chunks;
    LdsTabs* lt = reinterpret_cast<LdsTabs*>(tile + (P.rows_per_band + 2) * pitch);
    fill_lds_tabs<NT>(lt, P, tid);
    if (rsub < rpp) {
        const uint8_t* col = bgr + (int64_t)f * P.frame_stride + c * 48;
        uint4 q[NI][3];
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int t = min(rsub + k * rpp, trows - 1);  // surplus items re-read the last row (same bytes)
            const int y = reflect_once(r0 - 1 + t, h);
            const uint4* src = reinterpret_cast<const uint4*>(col + (int64_t)y * P.row_stride);
            if (P.dbg_skip & 16) { q[k][0] = q[k][1] = q[k][2] = make_uint4(t, y, k, c); continue; }
            q[k][0] = src[0]; q[k][1] = src[1]; q[k][2] = src[2];
        }
        uint8_t* dst = tile + kPad + c * 16;
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int t = min(rsub + k * rpp, trows - 1);
            const uint4 g = (P.dbg_skip & 8) ? make_uint4(q[k][0].x ^ q[k][1].y, q[k][0].y ^ q[k][2].x, q[k][1].z ^ q[k][2].w, q[k][0].w ^ q[k][1].x ^ q[k][2].z)
                                             : gray16(q[k][0], q[k][1], q[k][2]);
            uint8_t* d = dst + t * pitch;
            *reinterpret_cast<uint4*>(d) = g;
            if (c == 0) d[-1] = (uint8_t)(g.x >> 8);                 // pixel -1 := pixel 1
            if (c == chunks - 1) d[16] = (uint8_t)(g.w >> 16);       // pixel w  := pixel w-2
        }
    }
    __syncthreads();
    const Moments m = band_phases<NT, true>(tile, lt, P, f, band, r0, rows, tid, small, rowbuf);
    // per-wave partial moments, summed per frame in k_hash (atomics on the 16 B/frame accumulators
    // serialise in L2: 65 k same-line atomics cost ~50 us per launch)
    const long long s64 = wave_sum(m.s), q64 = wave_sum(m.q);
    if ((tid & 63) == 0) {
        long long* slot = lap_part + ((int64_t)lid * kLapSlots + (tid >> 6)) * 2;
        slot[0] = s64; slot[1] = q64;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// NV12 ingest: the decoder's surface goes straight into the fused pass.  Per pixel the BGR triple
// that cv2.VideoCapture.retrieve() would have produced (libswscale's table-driven yuv2rgb.c, BT.601 limited range,
// nearest chroma; reference app/analyzers/video.py:28-32) is formed in registers, reduced to cv2's 15-bit gray at
// once and written to the LDS tile: neither BGR nor gray ever reaches HBM, and the frame costs 1.5 bytes per pixel
// of HBM reads instead of 3.  Phases 2-4 are the BGR kernel's.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int clip8(int v) { return min(max(v, 0), 255); }

// chroma part of the three table lookups of one U,V pair: c0 + off * cy per channel
struct ChromaTerms { int r, g, b; };
// every product below has operands inside 24 bits (8-bit samples, 17-bit coefficients, table indices of a few hundred):
// __mul24 / __umul24 are full-rate instructions where a 32-bit multiply is quarter rate
__device__ __forceinline__ ChromaTerms chroma_terms(int U, int V, const YuvConsts& k)
{
    ChromaTerms t;
    t.r = k.c0 + __mul24((__mul24(V, k.crv) >> 16) + k.kr, k.cy);
    t.b = k.c0 + __mul24((__mul24(U, k.cbu) >> 16) + k.kb, k.cy);
    t.g = k.c0 + __mul24((__mul24(U, k.cgu) >> 16) + (__mul24(V, k.cgv) >> 16) + k.kg, k.cy);
    return t;
}

// the table form: the three indices' chroma parts (libswscale's per-U / per-V table offsets), biased so that Y + offset >= 0
constexpr int kNvBias = 224, kNvTab = 704;     // Y + offset in [-221, 475]
__device__ __forceinline__ ChromaTerms chroma_offsets(int U, int V, const YuvConsts& k)
{
    ChromaTerms t;
    t.r = (__mul24(V, k.crv) >> 16) + (k.kr + kNvBias);
    t.b = (__mul24(U, k.cbu) >> 16) + (k.kb + kNvBias);
    t.g = (__mul24(U, k.cgu) >> 16) + (__mul24(V, k.cgv) >> 16) + (k.kg + kNvBias);
    return t;
}

__device__ __forceinline__ unsigned gray_from_tables(int Y, const ChromaTerms& t, const unsigned* tabB, const unsigned* tabG, const unsigned* tabR)
{
    return (tabB[Y + t.b] + tabG[Y + t.g] + tabR[Y + t.r]) >> 15;
}

__device__ __forceinline__ unsigned gray_from_yuv(int Y, const ChromaTerms& t, int cy)
{
    // Y * cy + term: one v_mad_i32_i24 per channel
    const unsigned B = (unsigned)clip8((__mul24(Y, cy) + t.b) >> 16), G = (unsigned)clip8((__mul24(Y, cy) + t.g) >> 16),
                   R = (unsigned)clip8((__mul24(Y, cy) + t.r) >> 16);
    return (__umul24(B, 3735u) + __umul24(G, 19235u) + __umul24(R, 9798u) + (1u << 14)) >> 15;
}

// 4 luma bytes + 2 chroma pairs (U0 V0 U1 V1) -> 4 gray bytes
__device__ __forceinline__ unsigned gray4_nv12(unsigned yw, unsigned cw, const YuvConsts& k)
{
    const ChromaTerms t0 = chroma_terms(cw & 0xFF, (cw >> 8) & 0xFF, k);
    const ChromaTerms t1 = chroma_terms((cw >> 16) & 0xFF, cw >> 24, k);
    return gray_from_yuv(yw & 0xFF, t0, k.cy) | (gray_from_yuv((yw >> 8) & 0xFF, t0, k.cy) << 8) |
           (gray_from_yuv((yw >> 16) & 0xFF, t1, k.cy) << 16) | (gray_from_yuv(yw >> 24, t1, k.cy) << 24);
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_preprocess_nv12(const uint8_t* __restrict__ yplane, Nv12Params nv, int n,
                                                             PreParams P, uint8_t* __restrict__ small,
                                                             float* __restrict__ rowbuf, long long* __restrict__ lap_part)
{
    extern __shared__ __align__(16) uint8_t tile[];
    const int total = n * P.nbands;
    const int lid = xcd_remap(blockIdx.x, total);
    if (lid >= total) return;
    const int f = lid / P.nbands;
    const int band = lid - f * P.nbands;
    const int h = P.h, w = P.w, pitch = P.pitch;
    const int r0 = band * P.rows_per_band;
    const int rows = min(P.rows_per_band, h - r0);
    const int tid = threadIdx.x;
    const uint8_t* yfr = yplane + (int64_t)f * P.frame_stride;
    const uint8_t* cfr = nv.uv + (int64_t)f * nv.uv_frame_stride;
    const int trows = rows + 2;                            // tile row 0 = image row r0-1
    if (VEC) {
#ifndef AVD_NV12_ARITH
        // libswscale's converter IS a table lookup: B = T[Y + ob(U)], G = T[Y + og(U, V)], R = T[Y + or(V)] with one clip table T(i) = clip8((c0 + i cy) >> 16).
        // Round 5: three LDS tables of cv2's gray weight times T (3735 T, 19235 T + the rounding 2^14, 9798 T; 32-bit entries, index bias kNvBias), so a pixel is
        // three index additions, three ds_read_b32, one three-operand add and a shift -- instead of three multiply-adds, three shifts, three clamps and
        // three multiply-adds (12.3 -> 8.8 vector instructions per pixel; the LDS pipe does the lookups beside them).  Same integers by construction.
        unsigned* const tabB = reinterpret_cast<unsigned*>(tile + ((size_t)trows * pitch + 15) / 16 * 16);
        unsigned* const tabG = tabB + kNvTab;
        unsigned* const tabR = tabG + kNvTab;
        for (int i = tid; i < kNvTab; i += kThreads) {
            const unsigned v = (unsigned)clip8((nv.k.c0 + (i - kNvBias) * nv.k.cy) >> 16);
            tabB[i] = v * 3735u; tabG[i] = v * 19235u + (1u << 14); tabR[i] = v * 9798u;
        }
        __syncthreads();
#endif
        // one work item = one chroma row x one 16-pixel chunk: the eight chroma-term triples are formed once and serve the
        // two luma rows that share them (they are 8.5 of the ~27 integer operations a pixel costs otherwise)
        const int ylo = r0 - 1, yhi = r0 + rows;              // image rows of tile rows 0 and trows - 1, before reflection
        const int ya = max(ylo, 0), yb = min(yhi, h - 1);      // the ones that exist
        const int p0 = ya >> 1, np = (yb >> 1) - p0 + 1;
        const int chunks = w >> 4;
        for (int it = tid; it < np * chunks; it += kThreads) {
            const int pr = it / chunks, c = it - pr * chunks, p = p0 + pr;
            const uint4 cc = *reinterpret_cast<const uint4*>(cfr + (int64_t)p * nv.uv_row_stride + c * 16);
            const unsigned cw[4] = {cc.x, cc.y, cc.z, cc.w};
            ChromaTerms t[8];
#pragma unroll
            for (int j = 0; j < 4; j++) {
#ifndef AVD_NV12_ARITH
                t[2 * j] = chroma_offsets(cw[j] & 0xFF, (cw[j] >> 8) & 0xFF, nv.k);
                t[2 * j + 1] = chroma_offsets((cw[j] >> 16) & 0xFF, cw[j] >> 24, nv.k);
#else
                t[2 * j] = chroma_terms(cw[j] & 0xFF, (cw[j] >> 8) & 0xFF, nv.k);
                t[2 * j + 1] = chroma_terms((cw[j] >> 16) & 0xFF, cw[j] >> 24, nv.k);
#endif
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const int y = 2 * p + s2;
                if (y < ya || y > yb) continue;
                const uint4 yy = *reinterpret_cast<const uint4*>(yfr + (int64_t)y * P.row_stride + c * 16);
                const unsigned yw[4] = {yy.x, yy.y, yy.z, yy.w};
                unsigned g[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
#ifdef AVD_NV12_NOCONV      // timing experiment: no conversion arithmetic (results are wrong)
                    g[j] = yw[j] ^ cw[j];
#elif !defined(AVD_NV12_ARITH)
                    g[j] = gray_from_tables(yw[j] & 0xFF, t[2 * j], tabB, tabG, tabR) | (gray_from_tables((yw[j] >> 8) & 0xFF, t[2 * j], tabB, tabG, tabR) << 8) |
                           (gray_from_tables((yw[j] >> 16) & 0xFF, t[2 * j + 1], tabB, tabG, tabR) << 16) | (gray_from_tables(yw[j] >> 24, t[2 * j + 1], tabB, tabG, tabR) << 24);
#else
                    g[j] = gray_from_yuv(yw[j] & 0xFF, t[2 * j], nv.k.cy) | (gray_from_yuv((yw[j] >> 8) & 0xFF, t[2 * j], nv.k.cy) << 8) |
                           (gray_from_yuv((yw[j] >> 16) & 0xFF, t[2 * j + 1], nv.k.cy) << 16) | (gray_from_yuv(yw[j] >> 24, t[2 * j + 1], nv.k.cy) << 24);
#endif
                *reinterpret_cast<uint4*>(tile + (y - ylo) * pitch + kPad + c * 16) = make_uint4(g[0], g[1], g[2], g[3]);
            }
        }
        __syncthreads();
        // BORDER_REFLECT_101 rows: image row -1 is row 1, row h is row h - 2 (both already in the tile)
        if (ylo < 0)
            for (int c = tid; c < chunks; c += kThreads)
                *reinterpret_cast<uint4*>(tile + kPad + c * 16) = *reinterpret_cast<const uint4*>(tile + 2 * pitch + kPad + c * 16);
        if (yhi > h - 1)
            for (int c = tid; c < chunks; c += kThreads)
                *reinterpret_cast<uint4*>(tile + (trows - 1) * pitch + kPad + c * 16) =
                    *reinterpret_cast<const uint4*>(tile + (trows - 3) * pitch + kPad + c * 16);
    } else {
        for (int it = tid; it < trows * w; it += kThreads) {
            const int tr = it / w, x = it - tr * w;
            const int y = reflect_once(r0 - 1 + tr, h);
            const uint8_t* cp = cfr + (int64_t)(y >> 1) * nv.uv_row_stride + (x >> 1) * 2;
            const ChromaTerms t = chroma_terms(cp[0], cp[1], nv.k);
            tile[tr * pitch + kPad + x] = (uint8_t)gray_from_yuv(yfr[(int64_t)y * P.row_stride + x], t, nv.k.cy);
        }
    }
    __syncthreads();
    for (int tr = tid; tr < trows; tr += kThreads) {      // column halo (BORDER_REFLECT_101)
        uint8_t* row = tile + tr * pitch + kPad;
        row[-1] = row[reflect101(-1, w)];
        row[w] = row[reflect101(w, w)];
    }
    __syncthreads();
    const Moments m = band_phases<kThreads, false>(tile, nullptr, P, f, band, r0, rows, tid, small, rowbuf);
    const long long s64 = wave_sum(m.s), q64 = wave_sum(m.q);
    if ((tid & 63) == 0) {
        long long* slot = lap_part + ((int64_t)lid * kLapSlots + (tid >> 6)) * 2;
        slot[0] = s64; slot[1] = q64;
    }
}

// 32x32 INTER_AREA cells from the per-row partials (vertical accumulation in cv2's row
// order), then aHash bits: g >= mean(g)  <=>  1024*g >= sum(g)   (video.py:7-8)
__global__ __launch_bounds__(1024) void k_hash(const float* __restrict__ rowbuf, HashParams P,
                                              uint8_t* __restrict__ area, uint8_t* __restrict__ bits,
                                              const long long* __restrict__ lap_part, int nbands, int waves,
                                              unsigned long long* __restrict__ lap)
{
    __shared__ int wsum[16];
    __shared__ long long lsum[2][16];
    const int f = blockIdx.x, tid = threadIdx.x;
    {   // exact Laplacian moments of the frame = sum of the per-wave partials of its bands
        long long s = 0, q = 0;
        for (int i = tid; i < nbands * waves; i += 1024) {
            const long long* slot = lap_part + (((int64_t)f * nbands + i / waves) * kLapSlots + i % waves) * 2;
            s += slot[0]; q += slot[1];
        }
        s = wave_sum(s); q = wave_sum(q);
        if ((tid & 63) == 0) { lsum[0][tid >> 6] = s; lsum[1][tid >> 6] = q; }
    }
    const int dy = tid >> 5, dx = tid & 31;
    const float* rb = rowbuf + (int64_t)f * P.h * AVD_HASH + dx;
    const int y0 = P.ay_begin[dy], cnt = P.ay_count[dy];
    int cell;
    if (P.area_fast) {
        int acc = 0;
        for (int k0 = 0; k0 < cnt; k0 += 8) {
            int v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = __float_as_int(rb[(int64_t)(y0 + min(k0 + j, cnt - 1)) * AVD_HASH]);
#pragma unroll
            for (int j = 0; j < 8; j++) acc += k0 + j < cnt ? v[j] : 0;
        }
        if (dx < P.fast_simd_w) cell = (acc + 2) >> 2;
        else {
            const float scale = 1.f / (float)P.fast_area;
            cell = (int)rintf(__fmul_rn((float)acc, scale));
        }
    } else {
        const float wf = P.ay_first[dy], wm = P.ay_mid[dy], wl = P.ay_last[dy];
        float acc = 0.f;
        // the sum is a dependent chain in cv2's row order; the LOADS are not: eight rows are fetched at a time (one frame =
        // one workgroup, so the kernel's time is this chain's memory latency: 35 round trips before, 5 now)
        for (int k0 = 0; k0 < cnt; k0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = rb[(int64_t)(y0 + min(k0 + j, cnt - 1)) * AVD_HASH];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = k0 + j;
                if (k < cnt) {
                    const float beta = k == 0 ? wf : (k == cnt - 1 ? wl : wm);
                    const float t = __fmul_rn(beta, v[j]);
                    acc = k == 0 ? t : __fadd_rn(acc, t);
                }
            }
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
    if (tid == 0) {
        long long s = 0, q = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { s += lsum[0][i]; q += lsum[1][i]; }
        lap[2 * f] = (unsigned long long)s;
        lap[2 * f + 1] = (unsigned long long)q;
    }
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
#ifdef AVD_TIMING_EXPERIMENTS      // make EXTRA=-DAVD_TIMING_EXPERIMENTS: phase ablation (results are WRONG with a non-zero mask)
    { const char* e = std::getenv("AVD_DBG_SKIP"); P.dbg_skip = e ? std::atoi(e) : 0; }
#else
    P.dbg_skip = 0;
#endif
    const int total = n * P.nbands;
    const bool vec = (w % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(d_bgr) % 16 == 0);
    const int chunks = w >> 4;
    const size_t lds1 = (size_t)(P.rows_per_band + 2) * P.pitch;
    const size_t lds_vec = lds1 + sizeof(LdsTabs);
    const int grid1 = ((total + 7) / 8) * 8;
    static const int nt_pref = [] { const char* e = std::getenv("AVD_PRE_NT"); return e ? std::atoi(e) : 256; }();
    const int nt = (nt_pref == 512 && chunks <= 512) ? 512 : 256;
    const int ni_nt = (vec && chunks <= nt) ? (P.rows_per_band + 2 + nt / chunks - 1) / (nt / chunks) : 0;
    ws.lap_waves = kThreads / 64;
    if (ni_nt > 0 && ni_nt <= 9) {
        ws.lap_waves = nt / 64;
#define AVD_VEC_CASE(N)                                                                                              \
    do {                                                                                                             \
        if (nt == 512)                                                                                               \
            hipLaunchKernelGGL((k_preprocess_vec<N, 512>), dim3(grid1), dim3(512), lds_vec, ctx->stream, d_bgr, n, P, \
                               ws.d_small + (size_t)ws.f0 * AVD_NPIX, ws.d_rowbuf + ws.rowbuf_off, ws.d_lap_part + ws.lappart_off);                                              \
        else                                                                                                         \
            hipLaunchKernelGGL((k_preprocess_vec<N, 256>), dim3(grid1), dim3(256), lds_vec, ctx->stream, d_bgr, n, P, \
                               ws.d_small + (size_t)ws.f0 * AVD_NPIX, ws.d_rowbuf + ws.rowbuf_off, ws.d_lap_part + ws.lappart_off);                                              \
    } while (0)
        switch (ni_nt) {
        case 1: case 2: case 3: AVD_VEC_CASE(3); break;
        case 4: AVD_VEC_CASE(4); break;
        case 5: AVD_VEC_CASE(5); break;
        case 6: AVD_VEC_CASE(6); break;
        case 7: AVD_VEC_CASE(7); break;
        case 8: AVD_VEC_CASE(8); break;
        default: AVD_VEC_CASE(9); break;
        }
#undef AVD_VEC_CASE
    } else {
        const int grid = ((total + 7) / 8) * 8;
        const size_t lds = (size_t)(P.rows_per_band + 2) * P.pitch;
        if (vec)
            hipLaunchKernelGGL(k_preprocess<true>, dim3(grid), dim3(kThreads), lds, ctx->stream,
                               d_bgr, n, P, ws.d_small + (size_t)ws.f0 * AVD_NPIX, ws.d_rowbuf + ws.rowbuf_off, ws.d_lap_part + ws.lappart_off);
        else
            hipLaunchKernelGGL(k_preprocess<false>, dim3(grid), dim3(kThreads), lds, ctx->stream,
                               d_bgr, n, P, ws.d_small + (size_t)ws.f0 * AVD_NPIX, ws.d_rowbuf + ws.rowbuf_off, ws.d_lap_part + ws.lappart_off);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int launch_preprocess_nv12(avd_ctx* ctx, const uint8_t* d_y, const Nv12Params& nv, int n, int h, int w,
                           int64_t row_stride, int64_t frame_stride)
{
    Workspace& ws = ctx->ws;
    PreParams P = ws.pre;
    P.row_stride = row_stride;
    P.frame_stride = frame_stride;
    P.dbg_skip = 0;
    const int total = n * P.nbands;
    const bool vec = (w % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0) && (nv.uv_row_stride % 16 == 0) &&
                     (nv.uv_frame_stride % 16 == 0) && (reinterpret_cast<uintptr_t>(d_y) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(nv.uv) % 16 == 0);
    const int grid = ((total + 7) / 8) * 8;
    size_t lds = (size_t)(P.rows_per_band + 2) * P.pitch;
#ifndef AVD_NV12_ARITH
    if (vec) lds = (lds + 15) / 16 * 16 + 3 * sizeof(unsigned) * 704;     // the three conversion tables behind the tile (k_preprocess_nv12, kNvTab)
#endif
    // (a register-staged variant in the style of k_preprocess_vec measured no faster: the kernel is bound by the
    // conversion's integer arithmetic, not by how its loads are issued -- profiles/r02_experiments.md)
    ws.lap_waves = kThreads / 64;
    if (vec)
        hipLaunchKernelGGL(k_preprocess_nv12<true>, dim3(grid), dim3(kThreads), lds, ctx->stream, d_y, nv, n, P, ws.d_small + (size_t)ws.f0 * AVD_NPIX,
                           ws.d_rowbuf + ws.rowbuf_off, ws.d_lap_part + ws.lappart_off);
    else
        hipLaunchKernelGGL(k_preprocess_nv12<false>, dim3(grid), dim3(kThreads), lds, ctx->stream, d_y, nv, n, P, ws.d_small + (size_t)ws.f0 * AVD_NPIX,
                           ws.d_rowbuf + ws.rowbuf_off, ws.d_lap_part + ws.lappart_off);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// with_hamming = false: the caller forms the Hamming distances itself (k_records of the analyze entries does, from the bits)
int launch_hash(avd_ctx* ctx, int n, bool with_hamming)
{
    Workspace& ws = ctx->ws;
    // the clip's slice of the call's buffers: frame ws.f0 onwards (the first frame of a clip has no predecessor: ham = -1)
    const size_t f0 = (size_t)ws.f0;
    hipLaunchKernelGGL(k_hash, dim3(n), dim3(1024), 0, ctx->stream, ws.d_rowbuf + ws.rowbuf_off, ws.hsh, ws.d_area + f0 * 1024,
                       ws.d_hash + f0 * 1024, (const long long*)(ws.d_lap_part + ws.lappart_off), ws.pre.nbands, ws.lap_waves,
                       ws.d_lap + 2 * f0);
    if (with_hamming) hipLaunchKernelGGL(k_hamming, dim3(n), dim3(256), 0, ctx->stream, ws.d_hash + f0 * 1024, ws.d_ham + f0);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
