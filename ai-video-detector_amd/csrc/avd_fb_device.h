// avd_fb_device.h -- device helpers shared by the Farneback kernels (avd_farneback.hip, avd_fbfused.hip):
// FarnebackUpdateMatrices for one pixel (reference site: cv2.calcOpticalFlowFarneback, app/analyzers/video.py:45),
// split into the three steps of a software pipeline: inputs (flow, R0) -> bilinear gather of R1 at the warped
// position -> normal equations.  Float arithmetic in OpenCV's order, no contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int reflect101(int p, int len)
{
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int floor_f(float v) { int i = (int)v; return i - (i > v); }

struct NeIn { float dx, dy, r0[5]; };
struct NeG { float top[10], bot[10]; };            // (y1,x1..x1+1) and (y1+1,x1..x1+1), 5 coefficients each

struct __attribute__((packed, aligned(4))) F4 { float a, b, c, d; };
struct __attribute__((packed, aligned(4))) F2 { float a, b; };

// R is interleaved [frame][y][x][5]; flow planar [pair][2][y][x].  Addressing is "uniform base + unsigned
// 32-bit BYTE offset" (both buffers are smaller than 4 GiB), which the compiler turns into the
// saddr + voffset form of global_load: no 64-bit address arithmetic per load.  Wide loads on 4-byte-aligned
// addresses.
template <typename T>
__device__ __forceinline__ T ld_off(const void* __restrict__ base, unsigned byte_off)
{
    return *reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}

// A further piece of the same pixel: the constant is added AFTER the 32-bit offset has been widened, which is the form the compiler folds
// into the load's immediate offset field ("+ 16" on the 32-bit offset could wrap, so it would need a v_add_u32 per piece: 42 of the 585
// VALU instructions of the N waves' loop)
template <typename T, unsigned IMM>
__device__ __forceinline__ T ld_off_i(const void* __restrict__ base, unsigned byte_off)
{
    return *reinterpret_cast<const T*>(static_cast<const char*>(base) + (size_t)byte_off + IMM);
}

template <typename T>
__device__ __forceinline__ void st_off(void* __restrict__ base, unsigned byte_off, T v)
{
    *reinterpret_cast<T*>(static_cast<char*>(base) + byte_off) = v;
}

// non-temporal store: D is written once and read once by the next kernel.  Measured A/B on one box, 3 rounds
// each, D stores and D loads non-temporal vs plain: k_hscan<320> 0.146 -> 0.121 ms, k_uv<320> 0.288 -> 0.300 ms,
// 59.3 k -> 61.4 k frames/s with 3 clips in flight (loads alone: hscan 0.115 ms but 58.5 k frames/s; the same
// treatment of the preprocess input loads LOST 4 %)
template <typename T>
__device__ __forceinline__ void st_off_nt(void* __restrict__ base, unsigned byte_off, T v)
{
    __builtin_nontemporal_store(v, reinterpret_cast<T*>(static_cast<char*>(base) + byte_off));
}

template <bool SCALAR_ROW = false>
__device__ __forceinline__ void ne_load(const float* __restrict__ R, const float* __restrict__ flow, unsigned r0base,
                                        unsigned flbase, int x, int y, int w, int plane, NeIn& in)
{
    unsigned pb;
    if constexpr (!SCALAR_ROW) {
        const unsigned o = (unsigned)(y * w + x);
        in.dx = ld_off<float>(flow, (flbase + o) * 4u); in.dy = ld_off<float>(flow, (flbase + plane + o) * 4u);
        pb = (r0base + o * 5u) * 4u;
    } else {
        // the row's part of the offsets is a SCALAR the compiler cannot see through (the 14-wave latency shapes of the fast level kernels, 128
        // registers): left to itself it turns every one of these loads, in every unrolled step, into a 64-bit per-lane pointer that it advances
        // and spills (31 registers of the 80-px kernel, reloaded inside the loop); this way a load is "uniform base + (lane part + row part)",
        // one 32-bit add.  The 12-wave shape of the 320-px level has the registers and is five instructions per step better off without.
        const unsigned yo = (unsigned)__builtin_amdgcn_readfirstlane(y * w);
        const unsigned o4 = (yo + (unsigned)x) * 4u;
        in.dx = ld_off<float>(flow, flbase * 4u + o4); in.dy = ld_off<float>(flow, (flbase + plane) * 4u + o4);
        pb = r0base * 4u + o4 * 5u;
    }
    const F4 v = ld_off<F4>(R, pb);
    in.r0[0] = v.a; in.r0[1] = v.b; in.r0[2] = v.c; in.r0[3] = v.d; in.r0[4] = ld_off_i<float, 16>(R, pb);
}

// the R0 part alone (the flow comes from elsewhere)
__device__ __forceinline__ void ne_load_r0(const float* __restrict__ R, unsigned r0base, int x, int y, int w, NeIn& in)
{
    const unsigned pb = (r0base + (unsigned)(y * w + x) * 5u) * 4u;
    const F4 v = ld_off<F4>(R, pb);
    in.r0[0] = v.a; in.r0[1] = v.b; in.r0[2] = v.c; in.r0[3] = v.d; in.r0[4] = ld_off_i<float, 16>(R, pb);
}

// gather the four bilinear neighbours of the warped position (clamped address when outside:
// the values are discarded by ne_finish, exactly as cv2 takes the "else" branch there)
__device__ __forceinline__ void ne_gather(const float* __restrict__ R, unsigned r1base, const NeIn& in, int x, int y,
                                          int w, int h, int plane, NeG& g)
{
    const float fx = x + in.dx, fy = y + in.dy;
    const int x1 = clampi(floor_f(fx), 0, w - 2), y1 = clampi(floor_f(fy), 0, h - 2);
    const unsigned pb = (r1base + (unsigned)(y1 * w + x1) * 5u) * 4u, qb = pb + (unsigned)w * 20u;
    const F4 t0 = ld_off<F4>(R, pb), t1 = ld_off_i<F4, 16>(R, pb);
    const F2 t2 = ld_off_i<F2, 32>(R, pb);
    const F4 b0 = ld_off<F4>(R, qb), b1 = ld_off_i<F4, 16>(R, qb);
    const F2 b2 = ld_off_i<F2, 32>(R, qb);
    g.top[0] = t0.a; g.top[1] = t0.b; g.top[2] = t0.c; g.top[3] = t0.d; g.top[4] = t1.a;
    g.top[5] = t1.b; g.top[6] = t1.c; g.top[7] = t1.d; g.top[8] = t2.a; g.top[9] = t2.b;
    g.bot[0] = b0.a; g.bot[1] = b0.b; g.bot[2] = b0.c; g.bot[3] = b0.d; g.bot[4] = b1.a;
    g.bot[5] = b1.b; g.bot[6] = b1.c; g.bot[7] = b1.d; g.bot[8] = b2.a; g.bot[9] = b2.b;
}

__device__ __forceinline__ void ne_finish(const NeIn& in, const NeG& g, int x, int y, int w, int h, float (&M)[5])
{
    // Branch-free on purpose (selects, multiplication by an exact 1.0f): a conditional block here lets the
    // compiler sink the gathered loads into it, right in front of their use, and makes its vmcnt counts
    // conservative at the join -- either way the software pipeline of k_uv / k_uvp collapses.
    const float dx = in.dx, dy = in.dy;
    float fx = x + dx, fy = y + dy;
    const int x1 = floor_f(fx), y1 = floor_f(fy);
    fx -= x1; fy -= y1;
    const bool inside = (unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1);
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    const float b2 = a00 * g.top[0] + a01 * g.top[5] + a10 * g.bot[0] + a11 * g.bot[5];
    const float b3 = a00 * g.top[1] + a01 * g.top[6] + a10 * g.bot[1] + a11 * g.bot[6];
    const float b4 = a00 * g.top[2] + a01 * g.top[7] + a10 * g.bot[2] + a11 * g.bot[7];
    const float b5 = a00 * g.top[3] + a01 * g.top[8] + a10 * g.bot[3] + a11 * g.bot[8];
    const float b6 = a00 * g.top[4] + a01 * g.top[9] + a10 * g.bot[4] + a11 * g.bot[9];
    float r2 = inside ? b2 : 0.f, r3 = inside ? b3 : 0.f;
    float r4 = inside ? (in.r0[2] + b4) * 0.5f : in.r0[2];
    float r5 = inside ? (in.r0[3] + b5) * 0.5f : in.r0[3];
    float r6 = inside ? (in.r0[4] + b6) * 0.25f : in.r0[4] * 0.5f;
    r2 = (in.r0[0] - r2) * 0.5f;
    r3 = (in.r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    {
        auto border = [](int d) { return d < 2 ? 0.14f : 0.4472f; };      // {.14,.14,.4472,.4472,.4472}
        const bool edge = (unsigned)(x - 5) >= (unsigned)(w - 10) || (unsigned)(y - 5) >= (unsigned)(h - 10);
        const float sc = (x < 5 ? border(x) : 1.f) * (x >= w - 5 ? border(w - x - 1) : 1.f) *
                         (y < 5 ? border(y) : 1.f) * (y >= h - 5 ? border(h - y - 1) : 1.f);
        const float scale = edge ? sc : 1.f;             // interior: x * 1.0f == x exactly
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    M[0] = r4 * r4 + r6 * r6;
    M[1] = (r4 + r5) * r6;
    M[2] = r5 * r5 + r6 * r6;
    M[3] = r4 * r2 + r6 * r3;
    M[4] = r6 * r2 + r5 * r3;
}
// cv2.resize(prev, (W, W), INTER_LINEAR) * 2 (the initial flow of a pyramid level from the coarser level's result), four consecutive
// outputs dx = 4 q .. 4 q + 3 of row dy of one component plane: k_flow_up's arithmetic (avd_farneback.hip), shared by the kernels that fold
// that resize into themselves.  The destination is exactly twice the source, so the source coordinate d / 2 - 0.25 is exact in float:
// weights 0.75 / 0.25; horizontally the weights snap to the edge pixel when the source index falls outside, vertically the rows are
// clipped and the weights kept.
template <int W>
__device__ __forceinline__ void flow_up_chunk(const float* __restrict__ src, int dy, int q, float (&o)[4])
{
    constexpr int H = W, PW = W / 2, PH = H / 2;
    float fy = dy * 0.5f - 0.25f;
    const int sy = floor_f(fy);
    fy -= sy;
    const float* r0 = src + clampi(sy, 0, PH - 1) * PW;
    const float* r1 = src + clampi(sy + 1, 0, PH - 1) * PW;
    const float b0 = 1.f - fy, b1 = fy;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = q * 4 + i;
        float fx = dx * 0.5f - 0.25f;
        int sx = floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        bool edge = false;                                  // dx >= xmax: value copied, no weights
        if (sx + 1 >= PW) { edge = true; if (sx >= PW - 1) { fx = 0; sx = PW - 1; } }
        const int x1 = sx + 1 < PW - 1 ? sx + 1 : PW - 1;
        const float a0 = 1.f - fx, a1 = fx;
        float d0, d1;
        if (edge) { d0 = r0[sx] * 1.f; d1 = r1[sx] * 1.f; }
        else { d0 = r0[sx] * a0 + r0[x1] * a1; d1 = r1[sx] * a0 + r1[x1] * a1; }
        o[i] = (d0 * b0 + d1 * b1) * 2.f;
    }
}

// ---- leaner forms for the fused level kernel (avd_fbfused.hip): same arithmetic, fewer instructions per row ----------
// * the warped integer position is computed once (in the gather step) and carried to the finish step;
// * the 5-pixel border attenuation (x < 5 ? b[x] : 1) * (x >= w-5 ? b[w-x-1] : 1) * (y < 5 ? ...) * (y >= h-5 ? ...) is split
//   into a per-lane factor sx (constant for the whole kernel) and a per-row factor sy (wave-uniform).  For images of at
//   least 10 pixels at most one x factor and one y factor differ from 1, so cv2's left-to-right product equals sx * sy
//   bit for bit (multiplications by 1.0f are exact), and interior pixels multiply by exactly 1.
typedef float fv2 __attribute__((ext_vector_type(2)));
typedef float fv4 __attribute__((ext_vector_type(4)));
// the gathered rows as the three loads of a row deliver them: t0 = px0 c0..c3, t1 = (px0 c4, px1 c0, px1 c1, px1 c2), t2 = (px1 c3, px1 c4)
// fx, fy: fractional parts of the warped position; inside: cv2's test (unsigned)x1 < w - 1 && (unsigned)y1 < h - 1
struct NeG2 { fv4 t0, t1; fv2 t2; fv4 b0, b1; fv2 b2; float fx, fy; bool inside; };

// The bilinear sample of the five coefficients, cv2's operation order per channel -- ((a00 p00 + a01 p01) + a10 p10) + a11 p11 -- with the
// TEN products of a row formed as five packed multiplies on the register pairs the loads delivered (px1's coefficients start at an odd
// register: pairing by channel across the two pixels, or by pixel across channels, both need a move per pair -- 116 of the 585 VALU
// instructions of the N waves' loop were v_mov); the sums are scalar adds, which read any register.  Same IEEE operations: bit-identical.
__device__ __forceinline__ void ne_bilinear(const NeG2& g, float a00, float a01, float a10, float a11, float (&b)[5])
{
    const fv2 w00 = {a00, a00}, w0x = {a00, a01}, w01 = {a01, a01};
    const fv2 w10 = {a10, a10}, w1x = {a10, a11}, w11 = {a11, a11};
    const fv2 PA = g.t0.xy * w00, PB = g.t0.zw * w00, PC = g.t1.xy * w0x, PD = g.t1.zw * w01, PE = g.t2 * w01;
    const fv2 QA = g.b0.xy * w10, QB = g.b0.zw * w10, QC = g.b1.xy * w1x, QD = g.b1.zw * w11, QE = g.b2 * w11;
    b[0] = ((PA.x + PC.y) + QA.x) + QC.y;
    b[1] = ((PA.y + PD.x) + QA.y) + QD.x;
    b[2] = ((PB.x + PD.y) + QB.x) + QD.y;
    b[3] = ((PB.y + PE.x) + QB.y) + QE.x;
    b[4] = ((PC.x + PE.y) + QC.x) + QE.y;
}

__device__ __forceinline__ float border_factor(int p, int len)
{
    const float lo = p < 2 ? 0.14f : 0.4472f, hi = len - p - 1 < 2 ? 0.14f : 0.4472f;
    return (p < 5 ? lo : 1.f) * (p >= len - 5 ? hi : 1.f);
}

// zf (wave-uniform): the flow is known to be zero (first iteration of the coarsest level): whatever the flow buffer holds is
// ignored, which saves clearing it
__device__ __forceinline__ void ne_gather2(const float* __restrict__ R, unsigned r1base, const NeIn& in, int x, int y,
                                           int w, int h, NeG2& g, bool zf = false)
{
    // cvFloor and "fx -= x1" through v_floor_f32: floorf(v) IS (float)cvFloor(v) for |v| < 2^24 (and the same value beyond), three
    // instructions per coordinate (floor, convert, subtract) instead of six (truncate, convert back, compare, borrow, convert, subtract)
    const float fx = x + (zf ? 0.f : in.dx), fy = y + (zf ? 0.f : in.dy);
    const float flx = __builtin_floorf(fx), fly = __builtin_floorf(fy);
    g.fx = fx - flx; g.fy = fy - fly;
    g.inside = flx >= 0.f && flx <= (float)(w - 2) && fly >= 0.f && fly <= (float)(h - 2);
    const int x1 = clampi((int)flx, 0, w - 2), y1 = clampi((int)fly, 0, h - 2);
    const unsigned pb = (r1base + (unsigned)(y1 * w + x1) * 5u) * 4u, qb = pb + (unsigned)w * 20u;
    g.t0 = __builtin_bit_cast(fv4, ld_off<F4>(R, pb)); g.t1 = __builtin_bit_cast(fv4, ld_off_i<F4, 16>(R, pb));
    g.t2 = __builtin_bit_cast(fv2, ld_off_i<F2, 32>(R, pb));
    g.b0 = __builtin_bit_cast(fv4, ld_off<F4>(R, qb)); g.b1 = __builtin_bit_cast(fv4, ld_off_i<F4, 16>(R, qb));
    g.b2 = __builtin_bit_cast(fv2, ld_off_i<F2, 32>(R, qb));
}

__device__ __forceinline__ void ne_finish2(const NeIn& in, const NeG2& g, int x, int y, int w, int h, float sx, float sy,
                                           float (&M)[5], bool zf = false)
{
    const float dx = zf ? 0.f : in.dx, dy = zf ? 0.f : in.dy;
    const float fx = g.fx, fy = g.fy;
    const bool inside = g.inside;
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    float bb[5];
    ne_bilinear(g, a00, a01, a10, a11, bb);
    const float b2 = bb[0], b3 = bb[1], b4 = bb[2], b5 = bb[3], b6 = bb[4];
    float r2 = inside ? b2 : 0.f, r3 = inside ? b3 : 0.f;
    float r4 = inside ? (in.r0[2] + b4) * 0.5f : in.r0[2];
    float r5 = inside ? (in.r0[3] + b5) * 0.5f : in.r0[3];
    float r6 = inside ? (in.r0[4] + b6) * 0.25f : in.r0[4] * 0.5f;
    r2 = (in.r0[0] - r2) * 0.5f;
    r3 = (in.r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    const float scale = sx * sy;
    r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    M[0] = r4 * r4 + r6 * r6;
    M[1] = (r4 + r5) * r6;
    M[2] = r5 * r5 + r6 * r6;
    M[3] = r4 * r2 + r6 * r3;
    M[4] = r6 * r2 + r5 * r3;
}

// ---- the same split in two for the fast level kernel (avd_fbfast.hip): the normal-equation wave stops at r2 .. r6 (before
// the border attenuation), the chain wave -- which has issue slots to spare -- applies the attenuation and forms the five
// products.  Same operations in the same order as ne_finish2: bit-identical.
__device__ __forceinline__ void ne_finish_r(const NeIn& in, const NeG2& g, int x, int y, int w, int h, float (&r)[5], bool zf = false)
{
    const float dx = zf ? 0.f : in.dx, dy = zf ? 0.f : in.dy;
    const float fx = g.fx, fy = g.fy;
    const bool inside = g.inside;
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    float bb[5];
    ne_bilinear(g, a00, a01, a10, a11, bb);
    const float b2 = bb[0], b3 = bb[1], b4 = bb[2], b5 = bb[3], b6 = bb[4];
    float r2 = inside ? b2 : 0.f, r3 = inside ? b3 : 0.f;
    const float r4 = inside ? (in.r0[2] + b4) * 0.5f : in.r0[2];
    const float r5 = inside ? (in.r0[3] + b5) * 0.5f : in.r0[3];
    const float r6 = inside ? (in.r0[4] + b6) * 0.25f : in.r0[4] * 0.5f;
    r2 = (in.r0[0] - r2) * 0.5f;
    r3 = (in.r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    r[0] = r2; r[1] = r3; r[2] = r4; r[3] = r5; r[4] = r6;
}

__device__ __forceinline__ void ne_products(const float (&r)[5], float scale, float (&M)[5])
{
    const float r2 = r[0] * scale, r3 = r[1] * scale, r4 = r[2] * scale, r5 = r[3] * scale, r6 = r[4] * scale;
    M[0] = r4 * r4 + r6 * r6;
    M[1] = (r4 + r5) * r6;
    M[2] = r5 * r5 + r6 * r6;
    M[3] = r4 * r2 + r6 * r3;
    M[4] = r6 * r2 + r5 * r3;
}

// ---- shared by the fast level kernels (avd_fbfast.hip) -------------------------------------------------------------
// thresholds of the ill-posedness criteria (see role_solve and role_ne in avd_fbfast.hip)
constexpr double kCondMax = 2000.;
constexpr float kFlowMax = 0.3f;
// border-sign criterion (round 5, role_ne): a flow component below kTinyFlow in magnitude is of the size of cv2's own running-sum residue
// (<= ~1e-13 px), so its SIGN -- which decides "inside" / "outside" at the top / left border -- is not reproducible; kJumpMin: how much the
// two branches must differ at that pixel for the flip to matter
constexpr float kTinyFlow = 1e-12f;
constexpr float kJumpMin = 1e-6f;    // a non-zero component below kTinyFlow
constexpr float kJumpMinZero = 0.05f; // an exactly zero one (cv2's may be +-residue): two different FLAT frames, whose zero flow is structural, stay below
constexpr int kPairDiffTiles = 20;   // tiles per frame of the pyramid kernel's 160-px scale: each leaves "frame f differs from frame f + 1 here"

// What FarnebackUpdateMatrices' two branches disagree by at a pixel whose deciding flow component is (all but) zero: "outside" takes
// r2 = R0[0] / 2, r3 = R0[1] / 2, r4 .. r6 from R0 alone, "inside" (R0[0] - b[0]) / 2, .., (R0[2] + b[2]) / 2 .. with b the sample of R1, which is
// the top-left gathered pixel there -> max(|b0|, |b1|, |R0[2] - b2|, |R0[3] - b3|, |R0[4] - b4|).  Cold path of role_ne; the same definition in
// tools/experiments/fb_illposed_exp.c (border_ind).
__device__ __forceinline__ float ne_branch_jump(const NeIn& in, const NeG2& g)
{
    float j = fmaxf(fabsf(g.t0.x), fabsf(g.t0.y));
    j = fmaxf(j, fabsf(in.r0[2] - g.t0.z));
    j = fmaxf(j, fabsf(in.r0[3] - g.t0.w));
    return fmaxf(j, fabsf(in.r0[4] - g.t1.x));
}

// 1 / d as the compiler's IEEE division sequence computes it (v_rcp_f64, two Newton steps, a correction of the quotient)
// minus its scaling and fix-up instructions: they only act on denormal / huge / special operands, and d is a determinant
// plus 1e-3 in [1e-3, ~1e13].  Same result as 1. / d, bit for bit, on that range (tests/test_gpu_fbfast.py compares the
// flow of the exact kernels, which divide, against this one).
__device__ __forceinline__ double recip_exact(double d)
{
#ifdef AVD_FBF_PLAIN_DIV
    return 1. / d;
#else
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e0 = __builtin_fma(-d, r0, 1.);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-d, r1, 1.);
    const double r2 = __builtin_fma(r1, e1, r1);
    const double q = 1. * r2;
    const double e2 = __builtin_fma(-d, q, 1.);
    return __builtin_fma(e2, r2, q);
#endif
}

// the same without the final quotient step: two Newton steps on v_rcp_f64 (relative error ~1e-16 on [1e-3, 1e13]), for the fast level kernels
__device__ __forceinline__ double recip_newton2(double d)
{
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e0 = __builtin_fma(-d, r0, 1.);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-d, r1, 1.);
    return __builtin_fma(r1, e1, r1);
}

}  // namespace
