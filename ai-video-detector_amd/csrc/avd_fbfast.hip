// avd_fbfast.hip -- FarnebackUpdateFlow_Blur (winsize 15), ONE iteration per launch, a pair spread over several
// workgroups, no serial horizontal chain (gfx950).  The default level kernel since round 3 ("fb_mode" = fast).
//
// Reference site: cv2.calcOpticalFlowFarneback(prev, cur, None, 0.5, 3, 15, 3, 5, 1.2, 0), app/analyzers/video.py:45.
//
// What is literal and what is not.  cv2 forms the 15 x 15 box sums of the five normal-equation channels as RUNNING sums:
//   vertical    vsum[x] += (float)(M[y+7][x] - M[y-8][x])      a float difference added to a double, per column
//   horizontal  g       += (vsum[x+7] - vsum[x-8])             doubles
// The vertical chain rounds a FLOAT at every slide, so its value at row y depends on every row above: re-ordering it moves
// the flow by up to 8e-5 px on hard inputs (tools/experiments/fb_tolerance_exp.c, mode 2).  It is kept literal here: one
// "chain" wave per 64 columns walks down the rows exactly as cv2 does (lanes along x: the chain is sequential in y only).
// The horizontal chain only ever adds doubles; forming each window directly, sum of vsum[x-7 .. x+7] in double, differs from
// cv2's running value by a few double ulps (1e-16 relative), which the float cast of the flow swallows: on every input of
// that experiment (smooth clips, scene cuts, white noise, flat boxes) the flow comes out BIT-IDENTICAL to the oracle
// (mode 1: max |delta| = 0; 3e-14 px on a degenerate flat image).  tests/test_gpu_fbfast.py bounds it at 1e-5 px and
// ai_susp at 1e-6 on every geometry the parity suite covers; avd_fbfused.hip stays as fb_mode = exact.
//
// Structure.  The serial scanner wave of avd_fbfused.hip (one wave of 40 lanes walking a dependent double chain over 320
// columns, the pole of that kernel) is gone, and with it the 160 KB transpose buffer that held a workgroup to one per CU
// and a pair to one workgroup.  A workgroup owns a STRIP of a pair: OW output columns plus 7 halo columns on each side
// (recomputed, not exchanged: there is no cross-workgroup dependency inside a launch), all rows.  Columns outside the image
// clamp to the edge column, which is exactly cv2's replicated border of vsum.  Per 64-column block of the strip four waves:
//   N0, N1  normal equations (FarnebackUpdateMatrices up to r2 .. r6) of two image rows each per step -> ring in LDS (float)
//   C       border attenuation and the five products of M (the N waves are the pole of a step, this wave has issue slots to
//           spare), then the literal vertical chain: M rows in, vsum rows (double) out to LDS
//   X       horizontal window sums + 2 x 2 solve + flow stores.  A lane owns FOUR consecutive columns of one row: 18
//           doubles (nine 16-byte LDS reads) per channel give four windows with 21 additions, and a wave covers 4 rows x 64
//           columns per step
// (No touch / prefetch loads: the N waves' own three-entry lead covers the latency, and touching the next rows one dword
// per line from a fourth wave cost 0.16 ms per level: it is the texture-addresser / L1 path that limits this kernel.)
// One workgroup barrier per step of 4 rows; N works on group t, C on group t-1, X on group t-2.  A launch is one blur
// iteration: the flow is read from one buffer and written to another (strips of a pair read each other's columns), and the
// next iteration is the next launch.
//
// Bound: HBM.  Per iteration every frame's R is read once (20 B/px; pair p's R1 is pair p+1's R0 and neighbouring pairs
// share an XCD's L2), the flow is read and written (16 B/px per pair): 441 MB at 320 px x 119 pairs.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "avd_internal.h"
#include "avd_fb_device.h"

#pragma clang fp contract(off)

namespace {

constexpr int kM = 7;                 // (winsize - 1) / 2

typedef double dbl2 __attribute__((ext_vector_type(2)));

#ifdef AVD_FBF_NO_SCALAR_ROW       // A/B builds (tools/r05_ab_solve.sh)
constexpr bool kScalarRow = false;
#else
constexpr bool kScalarRow = true;
#endif

// W level size; NB 64-column blocks per strip; GD lead of the bilinear gather; NPB normal-equation waves per block (2: two
// entries each per step, 4: one each); XPB solver waves per block (1: four columns per lane, 2: two columns per lane).
// 2 + 1 + 1 waves per block is the throughput shape (320 px: the chip is full and total issue counts); 4 + 1 + 2 is the
// latency shape for the small levels, where a launch is steps x the slowest wave of a step and most CUs are idle anyway.
template <int W_, int NB_, int GD_ = 1, int NPB_ = 2, int XPB_ = 1>
struct FGeo {
    static constexpr int W = W_, NB = NB_, GD = GD_, NPB = NPB_, XPB = XPB_;
    static constexpr int H = W;
    static constexpr int SW = 64 * NB;               // lane columns of a strip
    static constexpr int WPB = NPB + 1 + XPB;        // waves per block
    static constexpr int NWAVES = WPB * NB;
    static constexpr int EPS = 4 / NPB;              // entries per normal-equation wave and step
    static constexpr int CPL = 4 / XPB;              // columns per solver lane
    // who applies the border attenuation and forms the five products of M: the chain wave where the N waves are the pole of a step
    // (throughput shape, 320 px), the N waves where the chain wave is (latency shape: four N waves per block, one entry each per step)
    static constexpr bool PN = NPB == 4;
    static constexpr int NE = H + kM;                // entries of the vertical chain: image row min(e, H-1)
    static constexpr int NG = (NE + 3) / 4;          // groups of four entries
    static constexpr int T = ((NG + 2 + 3) / 4) * 4; // steps (= barriers): N on group t, C on t-1, X on t-2; the loops unroll by 4
    // doubles per (row slot, channel) line.  Four columns per lane: lanes stride 32 bytes, ROWLEN % 4 == 2 puts the four
    // rows of a solver instruction on different halves of a 32-byte bank pair.  Two columns per lane: lanes stride 16
    // bytes (a row's 16 lanes = one 256-byte bank row), ROWLEN % 32 == 0 keeps the rows of an instruction on it
    static constexpr int ROWLEN = XPB == 1 ? SW + 18 : ((SW + 18 + 31) / 32) * 32;
    static constexpr int VS_SLOT = 5 * ROWLEN;
    static constexpr int VS_DOUBLES = 8 * VS_SLOT;   // ring of 8 image rows of vsum
    static constexpr int M_SLOT = 5 * 64;            // floats of one row of one block
    static constexpr int M_FLOATS = NB * 8 * M_SLOT; // ring of 8 rows per block
    static constexpr int LDS_DOUBLES = VS_DOUBLES + M_FLOATS / 2;
    // first iteration of a level below the coarsest (UP): the chain wave forms the level's initial flow -- the previous level's
    // flow resized x 2 -- a few rows ahead of the N waves, in a ring of 16 rows x 2 components per block
    static constexpr int F_SLOT = 2 * 64;            // floats of one row of one block
    static constexpr int F_FLOATS = NB * 16 * F_SLOT;
    static_assert((LDS_DOUBLES + F_FLOATS / 2) * 8 <= 163840 && NWAVES <= 16, "LDS layout, workgroup size");
    static_assert(H % 4 == 0 && (NPB == 2 || NPB == 4) && (XPB == 1 || XPB == 2), "whole groups of image rows");
};

// workgroup barrier; debug builds (-DAVD_FBF_DEBUG) account the cycles a wave spends waiting at it
#ifdef AVD_FBF_DEBUG
__device__ long long g_fbf_stamps[16][3];               // [wave][total cycles, cycles at barriers, role] of workgroup 0
#define fb_barrier()                                                      \
    do {                                                                  \
        const long long t0__ = __builtin_amdgcn_s_memtime();              \
        __syncthreads();                                                  \
        fbf_wait += __builtin_amdgcn_s_memtime() - t0__;                  \
    } while (0)
#define FBF_WAIT_DECL long long fbf_wait = 0;
#define FBF_WAIT_OUT(w, role, t0)                                                           \
    if (blockIdx.x == 8 && (threadIdx.x & 63) == 0) {                                       \
        g_fbf_stamps[w][0] = __builtin_amdgcn_s_memtime() - (t0);                           \
        g_fbf_stamps[w][1] = fbf_wait;                                                      \
        g_fbf_stamps[w][2] = role;                                                          \
    }
#else
__device__ __forceinline__ void fb_barrier() { __syncthreads(); }
#define FBF_WAIT_DECL
#define FBF_WAIT_OUT(w, role, t0)
#endif

// ------------------------------------------------------------------------------------------------------------------
// N: normal equations of entries 4t + 2k, 4t + 2k + 1 at step t (k = 0, 1), columns of one block.
// ------------------------------------------------------------------------------------------------------------------
// GD = entries of lead of the bilinear gather of R1 (its address needs the flow, so it cannot be issued arbitrarily early):
// the gather of entry i + GD is issued while entry i is evaluated, the inputs (flow, R0) of entry i + 2 GD + 1 likewise.
// Slots are statically indexed: the loop body is GD + 1 steps = 2 (GD + 1) entries.
// UP: the flow of this launch is the previous level's, resized: the chain wave leaves row y of it in fring[y & 15] at least
// one step before it is wanted here (see role_chain); it is fetched from there one entry before the gather that needs it.
// ZF (compile time: only the coarsest level's first iteration has it): the flow is known to be zero whatever the buffer holds
// Border-sign criterion (round 5; the second ill-posedness criterion, the first is in role_solve).  cv2's warp is discontinuous at the top / left
// border: at x = 0 a flow dx = -tiny gives x1 = -1, "outside" (the normal equations are formed from R0 alone), dx = +tiny gives x1 = 0, "inside"
// (average with R1); likewise dy at y = 0.  A component below kTinyFlow is of the size of the rounding residue of cv2's own running sums -- its
// sign is an artefact of their summation order, which this kernel does not share -- so where the two branches differ at such a pixel (ne_branch_jump)
// the pair can only be reproduced by the exact kernels: bit 4 + level of flags[p].  Derived on the CPU (tools/experiments/fb_illposed_exp.c,
// border_ind; profiles/r05_experiments.md section 1): it closes the checkerboard residual of round 4 (exactly periodic or static content, where
// the true flow is zero and everything is residue) and fires on no natural / noisy / letterboxed pair of the 3 120-pair experiment.
// chk (wave-uniform): flags are kept and the pair's two frames are not bit-identical (an exact duplicate's zero flow is structural in cv2 too).
// skip (may be null): an LDS word that says, after the workgroup's FIRST barrier, that the pair is flagged already (k_fb_fast) -- every role then returns true
template <typename Ge, bool UP, bool ZF = false>
__device__ __forceinline__ bool role_ne(const float* __restrict__ R, const float* __restrict__ flow, float* __restrict__ mring,
                                        const float* __restrict__ fring, int p, int x, int k, int lane, int* __restrict__ flags, bool chk,
                                        const volatile int* skip)
{
    constexpr bool zf = ZF;
    constexpr int W = Ge::W, GD = Ge::GD, EPS = Ge::EPS;
    FBF_WAIT_DECL
#ifdef AVD_FBF_DEBUG
    const long long fbf_t0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int H = W, plane = W * H;
    constexpr int NGS = GD + 1, NIS = 2 * NGS, U = NIS / EPS;   // gather slots, input slots, steps per loop body
    const unsigned r0base = (unsigned)p * 5u * plane, r1base = r0base + 5u * plane, flbase = (unsigned)p * 2u * plane;
    const float sxn = border_factor(x, W);                 // PN: x part of the border attenuation, a per-lane constant
    auto ent = [&](int i) { return 4 * (i / EPS) + EPS * k + (i % EPS); };  // this wave's i-th entry
    auto row_of = [](int e) { return e < H - 1 ? e : H - 1; };
    NeIn in[NIS];
    NeG2 g[NGS];
    const bool xz = x == 0;                                // this lane's column decides inside / outside by the sign of dx
    const bool chkx = chk && __builtin_amdgcn_ballot_w64(xz) != 0;   // wave-uniform: only the first block of the first strip holds column 0
    bool ill = false;
    auto gather = [&](const NeIn& s, int row, NeG2& gs) __attribute__((always_inline)) { ne_gather2(R, r1base, s, x, row, W, H, gs, zf); };
    auto flow_of = [&](int row, NeIn& s) { const float* f = fring + (row & 15) * Ge::F_SLOT + lane; s.dx = f[0]; s.dy = f[64]; };
    auto load_in = [&](int row, NeIn& s) {
        if (UP) ne_load_r0(R, r0base, x, row, W, s);
        else ne_load<Ge::PN && kScalarRow>(R, flow, r0base, flbase, x, row, W, plane, s);
    };
    bool first = skip != nullptr;                          // the first barrier of this role has not been passed yet
    if (UP) {
        fb_barrier();                                      // the chain wave has filled rows 0 .. 11 of the flow ring
        if (first && *skip) return true;
        first = false;
    }
#pragma unroll
    for (int i = 0; i < NIS - 1; i++) load_in(row_of(ent(i)), in[i]);
    if (UP) {
#pragma unroll
        for (int i = 0; i <= GD; i++) flow_of(row_of(ent(i)), in[i]);
    }
#pragma unroll
    for (int i = 0; i < GD; i++) gather(in[i], row_of(ent(i)), g[i]);
    auto work = [&](int t, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < EPS; j++) {
            const int i = EPS * t + j, ii = EPS * q + j;                     // ii = i mod NIS, static
            const int e = ent(i);
            // refills first: the gather GD entries ahead goes into the slot the previous entry released, the inputs
            // NIS - 1 entries ahead into the slot of the entry before this one (rows beyond the image clamp to the last)
#ifndef AVD_FBF_NOGATHER          // timing-only ablation builds (results are wrong)
            gather(in[(ii + GD) % NIS], row_of(ent(i + GD)), g[(ii + GD) % NGS]);
#endif
#ifndef AVD_FBF_NOINLOAD
            load_in(row_of(ent(i + NIS - 1)), in[(ii + NIS - 1) % NIS]);
#endif
            if (UP) flow_of(row_of(ent(i + GD + 1)), in[(ii + GD + 1) % NIS]);   // for the gather issued with the next entry
            __builtin_amdgcn_sched_barrier(0);
            float a[5];                                                      // r2 .. r6: the chain wave attenuates and multiplies (PN: done here)
            ne_finish_r(in[ii % NIS], g[ii % NGS], x, e, W, H, a, zf);
            if constexpr (!ZF) {                                             // (ZF: the coarsest level's first flow is zero in cv2 as well)
                if (chkx || (chk && e == 0)) {
                    const NeIn& s = in[ii % NIS];
                    const bool tx = chkx && xz && fabsf(s.dx) < kTinyFlow, ty = e == 0 && fabsf(s.dy) < kTinyFlow;
                    if (__builtin_amdgcn_ballot_w64(tx | ty) != 0) {         // cold: rows / columns of residue-sized flow
                        const float jump = ne_branch_jump(s, g[ii % NGS]);
                        ill |= (tx && jump > (s.dx == 0.f ? kJumpMinZero : kJumpMin)) | (ty && jump > (s.dy == 0.f ? kJumpMinZero : kJumpMin));
                    }
                }
            }
            float* dst = mring + (e & 7) * Ge::M_SLOT + lane;
            if constexpr (Ge::PN) {
                float mm[5];
                const int er = row_of(e);
                ne_products(a, sxn * border_factor(er, H), mm);
#pragma unroll
                for (int c = 0; c < 5; c++) dst[c * 64] = mm[c];
            } else {
#pragma unroll
                for (int c = 0; c < 5; c++) dst[c * 64] = a[c];
            }
        }
    };
    constexpr int TN = H / 4;                              // steps that bring image rows in; the rest only keep the barrier count
    int t0 = 0;
    for (; t0 + U <= TN; t0 += U) {
#pragma unroll
        for (int q = 0; q < U; q++) {
            fb_barrier();
            if (first) { first = false; if (*skip) return true; }
            work(t0 + q, q);
        }
    }
#pragma unroll
    for (int q = 0; q < TN % U; q++) {
        fb_barrier();
        work(t0 + q, q);
    }
    for (int t = TN; t < Ge::T; t++) fb_barrier();
    if (chk && __builtin_amdgcn_ballot_w64(ill) != 0 && lane == 0) atomicOr(flags + p, 16 << (Ge::W == 320 ? 0 : Ge::W == 160 ? 1 : Ge::W == 80 ? 2 : 3));
    FBF_WAIT_OUT(threadIdx.x >> 6, k, fbf_t0)
    return false;
}

// ------------------------------------------------------------------------------------------------------------------
// C: cv2's vertical running sums, literally.  Entry e brings image row min(e, H-1) in; from e = 7 on, row e - 15 (row 0
// while the window still touches the top edge) leaves and the vsum row of image row e - 7 is published.
// ------------------------------------------------------------------------------------------------------------------
// UP: the same wave (it has issue slots and registers to spare) also forms the launch's input flow, cv2's
// resize(prev_flow, INTER_LINEAR) * 2 as k_flow_up (avd_farneback.hip) computes it: source coordinate d / 2 - 0.25, weights
// 0.25 / 0.75, the first column with weights (1, 0), the last one copied.  Group g = rows 4g .. 4g+3 needs the four coarse
// rows 2g-1 .. 2g+2 (clamped); it is written during step g - 3 (groups 0 .. 2 before the first step), read by the N waves
// from step g - 2 on, and its ring slot is reused four steps later.
template <typename Ge>
struct UpRows { F2 v[4][2]; };

template <typename Ge>
__device__ __forceinline__ void up_issue(const float* __restrict__ prev, unsigned pbase, int g, int x, UpRows<Ge>& u)
{
    constexpr int PW = Ge::W / 2, PH = Ge::H / 2;
    const int sx = x == 0 ? 0 : (x - 1) >> 1;            // floor(x / 2 - 0.25), the first column snapped to 0
    const int xs = sx < PW - 2 ? sx : PW - 2;            // an 8-byte load never leaves the row
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int r = 2 * g - 1 + q;
        r = r < 0 ? 0 : (r > PH - 1 ? PH - 1 : r);
#pragma unroll
        for (int c = 0; c < 2; c++) u.v[q][c] = ld_off<F2>(prev, (pbase + (unsigned)(c * PW * PH + r * PW + xs)) * 4u);
    }
}

template <typename Ge>
__device__ __forceinline__ void up_finish(float* __restrict__ fring, int g, int x, int lane, const UpRows<Ge>& u)
{
    constexpr int W = Ge::W, PW = W / 2;
    const int sx = x == 0 ? 0 : (x - 1) >> 1;
    const bool edge = sx + 1 >= PW;                      // the last column: the value is copied, no weights
    const float a1 = x == 0 ? 0.f : ((x & 1) ? 0.25f : 0.75f), a0 = 1.f - a1;
    float d[4][2];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float lo = sx > PW - 2 ? u.v[q][c].b : u.v[q][c].a, hi = u.v[q][c].b;
            d[q][c] = edge ? lo * 1.f : lo * a0 + hi * a1;
        }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int y = 4 * g + j;                          // < H: only whole groups of image rows are formed
        const int q0 = j == 0 ? 0 : (j == 3 ? 2 : 1);     // floor(y / 2 - 0.25) - (2g - 1)
        const float b1 = (j & 1) ? 0.25f : 0.75f, b0 = 1.f - b1;
        float* dst = fring + (y & 15) * Ge::F_SLOT + lane;
#pragma unroll
        for (int c = 0; c < 2; c++) dst[c * 64] = (d[q0][c] * b0 + d[q0 + 1][c] * b1) * 2.f;
    }
}

template <typename Ge, bool UP>
__device__ __forceinline__ bool role_chain(const float* __restrict__ mring, double* __restrict__ vsring, float* __restrict__ fring,
                                           const float* __restrict__ prev, int p, int b, int x, int lane, const volatile int* skip)
{
    constexpr int W = Ge::W;
    FBF_WAIT_DECL
#ifdef AVD_FBF_DEBUG
    const long long fbf_t0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int H = W;
    const float sx = border_factor(x, W);                 // x part of the border attenuation: a per-lane constant
    float ring[16][5];                                   // ring[e & 15] = M row of entry e (statically indexed)
    double vs[5] = {0., 0., 0., 0., 0.};
    double* vdst = vsring + 8 + 64 * b + lane;
    const unsigned pbase = (unsigned)p * 2u * (W / 2) * (H / 2);
    UpRows<Ge> up;
    bool first = skip != nullptr;
    if (UP) {
#pragma unroll
        for (int g = 0; g < 3; g++) { up_issue<Ge>(prev, pbase, g, x, up); up_finish<Ge>(fring, g, x, lane, up); }
        fb_barrier();
        if (first && *skip) return true;
        first = false;
    }
    for (int t4 = 0; t4 < Ge::T; t4 += 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int t = t4 + q;
            fb_barrier();
            if (first) { first = false; if (*skip) return true; }
            const bool up_now = UP && 4 * (t + 3) < H;                       // wave-uniform
            if (up_now) up_issue<Ge>(prev, pbase, t + 3, x, up);             // consumed after the chain work of this step
            if (t >= 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int e = 4 * (t - 1) + j;
                    const int kk = 4 * ((q + 3) & 3) + j;                    // e & 15, static
                    if (e < Ge::NE) {                                        // wave-uniform
                        const int er = e < H - 1 ? e : H - 1;
                        const float* src = mring + (er & 7) * Ge::M_SLOT + lane;
                        float rr[5], a[5];
#pragma unroll
                        for (int c = 0; c < 5; c++) rr[c] = src[c * 64];
                        if constexpr (Ge::PN) {
#pragma unroll
                            for (int c = 0; c < 5; c++) a[c] = rr[c];          // the N waves left the products themselves
                        } else {
                            ne_products(rr, sx * border_factor(er, H), a);   // FarnebackUpdateMatrices' last lines, moved here from the N waves
                        }
                        if (e == 0) {
#pragma unroll
                            for (int c = 0; c < 5; c++) vs[c] = (double)(a[c] * (float)(kM + 2));
                        } else if (e < kM) {
#pragma unroll
                            for (int c = 0; c < 5; c++) vs[c] += (double)a[c];
                        } else {
                            const bool top = e < 16;                         // the leaving row is still row 0
#pragma unroll
                            for (int c = 0; c < 5; c++) {
                                const float lv = top ? ring[0][c] : ring[(kk + 1) & 15][c];
                                vs[c] += (double)(a[c] - lv);
                            }
                            double* d = vdst + ((e - kM) & 7) * Ge::VS_SLOT;
#pragma unroll
                            for (int c = 0; c < 5; c++) d[c * Ge::ROWLEN] = vs[c];
                        }
#pragma unroll
                        for (int c = 0; c < 5; c++) ring[kk][c] = a[c];
                    }
                }
            }
            if (up_now) up_finish<Ge>(fring, t + 3, x, lane, up);
        }
    }
    FBF_WAIT_OUT(threadIdx.x >> 6, 2, fbf_t0)
    return false;
}

// ------------------------------------------------------------------------------------------------------------------
// X: window sums of 15 columns, 2 x 2 solve (double, cv2's operation order), flow stores.  Lane = (row r of the group,
// chunk of four output columns).  The strip's lane column u holds image column clamp(xlo + u) with xlo = o0 - 7, and vsum
// of lane column u sits at index u + 8 of its line: the window of output column o0 + i is lane columns i .. i + 14.
// ------------------------------------------------------------------------------------------------------------------
// Ill-posedness flag (round 4).  Where the 2 x 2 normal equations are singular over whole regions the flow is chaotic: the
// ORACLE's own result moves by tens to hundreds of pixels under one ulp of noise on its inputs, and a kernel that re-orders a
// single double addition (this one: the horizontal window sums) cannot follow it.  Such pairs are recognised here and re-run
// by the exact kernels (avd_fbfused.hip, k_fb_rerun) before anything reads their flow.  Criterion, derived on the CPU over
// 1 440 pairs of 24 content families (tools/experiments/fb_illposed_run.py; profiles/r04_experiments.md section 1):
//   (g11 g22 + g12^2) > kCondMax (g11 g22 - g12^2 + 1e-3)   cancellation in the determinant: 1-D structure (ramps, stripes:
//                                                            3e5 .. 1e7; every natural / noise / scene-cut / letterboxed /
//                                                            saturated pair of the experiment: <= 975)
//   max(|fx|, |fy|) > kFlowMax * W                           a displacement beyond what a 15-px window can estimate at this
//                                                            level (the same experiment: <= 0.25 W on well-posed pairs)
// written so that a NaN or a negative determinant also fires.
template <typename Ge, bool UP>
__device__ __forceinline__ bool role_solve(const double* __restrict__ vsring, float* __restrict__ flow_out, float* __restrict__ mag_out,
                                           int* __restrict__ flags, int p, int b, int xi, int lane, int o0, int ow, const volatile int* skip)
{
    FBF_WAIT_DECL
#ifdef AVD_FBF_DEBUG
    const long long fbf_t0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int W = Ge::W, H = W, plane = W * H, CPL = Ge::CPL, NV = 14 + CPL;
    [[maybe_unused]] const double scale = 1. / (15 * 15);
    const int r = lane >> 4, j = 16 * (Ge::XPB * b + xi) + (lane & 15);      // chunk of CPL columns: output columns CPL j ..
    const bool colok = CPL * j < ow;
    float* fl = flow_out + (size_t)p * 2 * plane + o0 + CPL * j;
    const double* vsrc = vsring + 8 + CPL * j;
    bool first = skip != nullptr;
    if (UP) {
        fb_barrier();                                      // the chain wave's fill of the flow ring
        if (first && *skip) return true;
        first = false;
    }
    bool ill = false;
    for (int t4 = 0; t4 < Ge::T; t4 += 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int t = t4 + q;
            fb_barrier();
            if (first) { first = false; if (*skip) return true; }
            const int y = 4 * t - 15 + r;
#ifdef AVD_FBF_NOSOLVE
            if (false) {
#else
            if (t >= 2 && colok && y >= 0 && y < H) {
#endif
                const dbl2* s = reinterpret_cast<const dbl2*>(vsrc + (y & 7) * Ge::VS_SLOT);
                double o[5][CPL];
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    double v[NV];
#pragma unroll
                    for (int i = 0; i < NV / 2; i++) {
                        const dbl2 w2 = s[c * (Ge::ROWLEN / 2) + i];
                        v[2 * i] = w2.x; v[2 * i + 1] = w2.y;
                    }
                    if (CPL == 4) {
                        double A = v[3];
#pragma unroll
                        for (int i = 4; i < 15; i++) A += v[i];
#ifdef AVD_FBF_SOLVE_R4
                        const double p12 = v[1] + v[2], q2 = v[15] + v[16];
                        o[c][0] = A + (v[0] + p12);
                        o[c][1] = A + (p12 + v[15]);
                        o[c][2] = A + (v[2] + q2);
                        o[c][CPL - 1] = A + (q2 + v[NV - 1]);
#else
                        // four windows of fifteen from eighteen values in 19 additions: the twelve they share, then the two pairs next to them
                        const double Bl = A + (v[1] + v[2]), Br = A + (v[15] + v[16]);
                        o[c][0] = Bl + v[0];
                        o[c][1] = Bl + v[15];
                        o[c][2] = Br + v[2];
                        o[c][CPL - 1] = Br + v[NV - 1];
#endif
                    } else {
                        double A = v[1];
#pragma unroll
                        for (int i = 2; i < 15; i++) A += v[i];
                        o[c][0] = A + v[0];
                        o[c][CPL - 1] = A + v[15];
                    }
                }
                float fx[CPL], fy[CPL];
#pragma unroll
                for (int i = 0; i < CPL; i++) {
#ifdef AVD_FBF_SOLVE_R4
                    const double g11 = o[0][i] * scale, g12 = o[1][i] * scale, g22 = o[2][i] * scale;
                    const double h1 = o[3][i] * scale, h2 = o[4][i] * scale;
                    const double t1 = g11 * g22, t2 = g12 * g12, den = t1 - t2 + 1e-3;      // cv2's determinant, its operation order
                    const double idet = recip_exact(den);
                    fx[i] = (float)((g11 * h2 - g12 * h1) * idet);
                    fy[i] = (float)((g22 * h1 - g12 * h2) * idet);
                    ill |= !(t1 + t2 <= kCondMax * den) | !(fmaxf(fabsf(fx[i]), fabsf(fy[i])) <= kFlowMax * (float)W);
#else
                    // cv2 scales the five sums by 1 / 225 and adds 1e-3 to the determinant; the scale cancels in the quotient, so the sums stay as
                    // they are and the 1e-3 becomes 1e-3 * 225^2 (round 5: 14 double operations per column instead of 28; this mode's flow is held
                    // to a tolerance, not to cv2's rounding -- the exact kernels keep cv2's order).  The criterion t1 + t2 <= kCondMax den with
                    // t1 = den - c + t2 reads t2 <= (kCondMax - 1) / 2 den + c / 2.
                    constexpr double kC = 1e-3 * 225. * 225.;
                    const double a = o[0][i], b = o[1][i], d = o[2][i], h1 = o[3][i], h2 = o[4][i];
                    const double t2 = b * b, den = __builtin_fma(a, d, kC - t2);
                    const double idet = recip_newton2(den);
                    fx[i] = (float)(__builtin_fma(a, h2, -(b * h1)) * idet);
                    fy[i] = (float)(__builtin_fma(d, h1, -(b * h2)) * idet);
                    ill |= !(t2 <= __builtin_fma(0.5 * (kCondMax - 1.), den, 0.5 * kC)) | !(fmaxf(fabsf(fx[i]), fabsf(fy[i])) <= kFlowMax * (float)W);
#endif
                }
                // last iteration of the 320-px level: |flow| as np.sqrt(fx * fx + fy * fy) forms it in float32 (video.py:46), for the
                // statistics kernels -- they then read 4 bytes per pixel twice instead of 8, and the flow only once, here
                if (mag_out) {
                    float mg[CPL];
#pragma unroll
                    for (int i = 0; i < CPL; i++) { const float a2 = fx[i] * fx[i], b2 = fy[i] * fy[i]; mg[i] = sqrtf(a2 + b2); }
                    float* md = mag_out + (size_t)p * plane + y * W + o0 + CPL * j;
                    if (CPL == 4) *reinterpret_cast<float4*>(md) = make_float4(mg[0], mg[1], mg[2 % CPL], mg[3 % CPL]);
                    else *reinterpret_cast<float2*>(md) = make_float2(mg[0], mg[1]);
                }
                float* dst = fl + y * W;
                if (CPL == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(fx[0], fx[1], fx[2 % CPL], fx[3 % CPL]);
                    *reinterpret_cast<float4*>(dst + plane) = make_float4(fy[0], fy[1], fy[2 % CPL], fy[3 % CPL]);
                } else {
                    *reinterpret_cast<float2*>(dst) = make_float2(fx[0], fx[1]);
                    *reinterpret_cast<float2*>(dst + plane) = make_float2(fy[0], fy[1]);
                }
            }
        }
    }
    if (flags && __builtin_amdgcn_ballot_w64(ill) != 0 && lane == 0) atomicOr(flags + p, 1 << (Ge::W == 320 ? 0 : Ge::W == 160 ? 1 : Ge::W == 80 ? 2 : 3));
    FBF_WAIT_OUT(threadIdx.x >> 6, 3, fbf_t0)
    return false;
}

// between two phases of a launch that hand the flow over through global memory (prologue -> first iteration, iteration -> next
// iteration) INSIDE one workgroup: its waves share the CU's write-through L1, so workgroup scope is all it takes (the stores have
// left the wave, every wave has arrived).  Agent scope here costs a write-back of the L2 per phase on this chip (the XCDs' L2s are
// not coherent with each other): the 80-px level took 158 instead of 79 us with it.
__device__ __forceinline__ void phase_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// UP  (320 px): flow_in is the PREVIOUS level's flow ([pair][2][H/2][W/2]); the chain wave forms the launch's input flow from it on the
//     fly, a few rows ahead of the normal-equation waves (LDS ring).
// PRO (160 / 80 px): flow_in is the previous level's flow as well, but the whole workgroup resizes it in a PROLOGUE into flow_tmp (the
//     columns its own lanes read; strips of one pair write identical values where they overlap) -- at these sizes the chain wave is the
//     pole of a step and cannot take the resize on, while a prologue costs ~1 us and saves k_flow_up's launch.
// IT  iterations inside the launch (levels whose pairs are ONE strip: no other workgroup reads this one's flow): the flow ping-pongs
//     between flow_out and flow_tmp through L2, one workgroup barrier between iterations; the result is in flow_out (IT odd).
template <typename Ge, bool UP, int IT = 1, bool PRO = false>
__global__ __launch_bounds__((64 * Ge::NWAVES)) void k_fb_fast(const float* __restrict__ R, const float* flow_in, float* flow_out, float* flow_tmp,
                                                              float* __restrict__ mag_out, int* __restrict__ flags, const int* __restrict__ pairdiff,
                                                              int npairs, int nstrips, int ow, int zero_first, int dbg)
{
    constexpr int W = Ge::W, NB = Ge::NB;
    static_assert(IT == 1 || IT == 3, "one iteration per launch, or all three");
    static_assert(!(UP && (PRO || IT > 1)), "the chain wave's resize is the 320-px launch's");
    __shared__ __align__(16) double lds[Ge::LDS_DOUBLES + (UP ? Ge::F_FLOATS / 2 : 0)];
    double* vsring = lds;
    float* mrings = reinterpret_cast<float*>(lds + Ge::VS_DOUBLES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (XCD group, pair inside the group, strip): an XCD's L2 serves a contiguous run of pairs -- pair p gathers as R1 the
    // frame pair p + 1 reads as R0 -- and all strips of a pair; workgroups are dealt round-robin to the 8 XCDs
    const int ppx = (npairs + 7) >> 3;
    const int local = blockIdx.x >> 3;
    const int p = (blockIdx.x & 7) * ppx + local / nstrips;
    const int s = local % nstrips;
    if (p >= npairs) return;                              // whole workgroup
    const int o0 = s * ow;
    const int width = o0 + ow <= W ? ow : W - o0;         // output columns of this strip (multiples of 4)
    // Waves w, w + 4, w + 8 of a workgroup share a SIMD.  Issue cycles per step: X ~ 900 (double), N ~ 550, C ~ 230: with
    // NB = 3 the SIMDs get {X, N1, C} of block 0, 1, 2 and {N0, N0, N0} -- 1700 / 1700 / 1700 / 1650 cycles
    // role: 0 .. NPB-1 normal equations, NPB the chain, NPB+1 .. solvers
    int b, role;
    if (NB == 3 && Ge::WPB == 4) {
        const int pos = wave & 3, grp = wave >> 2;
        b = pos == 3 ? grp : pos;
        role = pos == 3 ? 0 : (grp == 0 ? 3 : (grp == 1 ? 1 : 2));
    } else {
        b = wave / Ge::WPB;
        role = (wave % Ge::WPB + b) % Ge::WPB;            // rotate the roles over a block's waves: every SIMD gets a mix
    }
    const int xu = o0 - kM + 64 * b + lane;               // image column of this lane (clamped: replicated border)
    const int x = xu < 0 ? 0 : (xu > W - 1 ? W - 1 : xu);
    float* mring = mrings + b * 8 * Ge::M_SLOT;
    float* fring = reinterpret_cast<float*>(lds + Ge::LDS_DOUBLES) + b * 16 * Ge::F_SLOT;   // UP only
    // ONE lane reads the pair's flag word (every thread reading it for itself could see different values while another strip of the pair raises it:
    // waves of one workgroup would part ways at a barrier)
    __shared__ int s_skip;
    if (flags && role == Ge::NPB + 1 && b == 0 && lane == 0) s_skip = __builtin_nontemporal_load(flags + p);
    if (PRO) {
        // columns any lane of this strip reads: [o0 - 7, o0 - 7 + 64 NB) clamped to the image, widened to whole chunks of four
        constexpr int H = W, plane = W * H, pplane = (W / 2) * (H / 2);
        const int c_lo = (o0 - kM < 0 ? 0 : o0 - kM) >> 2, c_hi = (o0 - kM + 64 * NB - 1 > W - 1 ? W - 1 : o0 - kM + 64 * NB - 1) >> 2;
        const int nq = c_hi - c_lo + 1;
        for (int item = threadIdx.x; item < 2 * H * nq; item += 64 * Ge::NWAVES) {
            const int q = c_lo + item % nq, dy = (item / nq) % H, c = item / (nq * H);
            float o[4];
            flow_up_chunk<W>(flow_in + ((size_t)p * 2 + c) * pplane, dy, q, o);
            *reinterpret_cast<float4*>(flow_tmp + ((size_t)p * 2 + c) * plane + dy * W + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
        phase_sync();
        if (flags && s_skip) return;                       // (behind the prologue's own barrier: workgroup-uniform)
    }
    // A pair that an earlier launch (or another strip, a moment ago) has flagged is re-run by the exact kernels whatever happens here: the workgroup
    // leaves (a clip of nothing but such pairs then costs the exact mode's time, not the sum of both).  The flag word is read by ONE lane of a solver wave
    // -- which has nothing to do during the first two steps -- into LDS, and every role looks at it after the workgroup's first barrier: the read's
    // latency hides behind the normal-equation waves' first loads instead of standing in front of them (at the head of the kernel it cost 1.5-2.5 % of the launch)
    const volatile int* skip = (flags && !PRO) ? &s_skip : nullptr;
    const float* fin0 = PRO ? flow_tmp : flow_in;
    if (role < Ge::NPB) {
        // border-sign criterion: not for a pair of bit-identical frames (the pyramid kernel left "frame p differs from frame p + 1" per tile)
        bool chk = flags != nullptr;
        if (chk && pairdiff) chk = __builtin_amdgcn_ballot_w64(lane < kPairDiffTiles && pairdiff[p * kPairDiffTiles + (lane < kPairDiffTiles ? lane : 0)] != 0) != 0;
#pragma unroll 1
        for (int it = 0; it < IT; it++) {
            if (it > 0) phase_sync();
            const float* fin = it == 0 ? fin0 : ((it & 1) ? flow_out : flow_tmp);
            const volatile int* sk = it == 0 ? skip : nullptr;
            if (Ge::W == 40 && zero_first != 0 && it == 0) { if (role_ne<Ge, UP, Ge::W == 40>(R, fin, mring, fring, p, x, role, lane, flags, chk, sk)) return; }
            else if (role_ne<Ge, UP, false>(R, fin, mring, fring, p, x, role, lane, flags, chk, sk)) return;
        }
    } else if (role == Ge::NPB) {
        if (!(dbg & 4)) __builtin_amdgcn_s_setprio(3);    // the only sequential part: take the issue slot whenever ready
#pragma unroll 1
        for (int it = 0; it < IT; it++) {
            if (it > 0) phase_sync();
            if (role_chain<Ge, UP>(mring, vsring, fring, flow_in, p, b, x, lane, it == 0 ? skip : nullptr)) return;
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < IT; it++) {
            if (it > 0) phase_sync();
            if (role_solve<Ge, UP>(vsring, (it & 1) ? flow_tmp : flow_out, it == IT - 1 ? mag_out : nullptr, flags, p, b, role - Ge::NPB - 1, lane, o0, width,
                                   it == 0 ? skip : nullptr)) return;
        }
    }
}

// mode: 0 = one iteration flow_in -> flow_out; 1 (320 px) = the same with the chain wave's resize of the previous level's flow; 2 = one
// iteration behind a prologue that resizes the previous level's flow into flow_tmp; 3 = all three iterations (flow_in ignored when
// zero_first, result in flow_out); 4 = prologue + all three iterations
template <typename Ge>
void launch_fast(hipStream_t stream, const float* R, const float* fin, float* fout, float* ftmp, float* mag, int* flags, const int* pairdiff, int np,
                 int nstrips, int ow, int zero_first, int mode)
{
    const int grid = 8 * ((np + 7) / 8) * nstrips;
    static const int dbg = [] { const char* e = std::getenv("AVD_FBF_DBG"); return e ? std::atoi(e) : 0; }();   // tuning experiments
    const dim3 g(grid), t(64 * Ge::NWAVES);
    // the chain wave's resize only exists where it pays: at 320 px it costs the launch 3.5 us and saves k_flow_up's 37; the small
    // levels are latency-bound on exactly the chain wave that would do it (160 px: 45 -> 84 us per launch against 12 saved)
    if constexpr (Ge::W == 320) {
        if (mode == 1) { hipLaunchKernelGGL((k_fb_fast<Ge, true>), g, t, 0, stream, R, fin, fout, ftmp, mag, flags, pairdiff, np, nstrips, ow, 0, dbg); return; }
    }
    if constexpr (Ge::W == 160 || Ge::W == 80) {
        if (mode == 2) { hipLaunchKernelGGL((k_fb_fast<Ge, false, 1, true>), g, t, 0, stream, R, fin, fout, ftmp, mag, flags, pairdiff, np, nstrips, ow, 0, dbg); return; }
    }
    if constexpr (Ge::W == 80) {
        if (mode == 4) { hipLaunchKernelGGL((k_fb_fast<Ge, false, 3, true>), g, t, 0, stream, R, fin, fout, ftmp, mag, flags, pairdiff, np, nstrips, ow, 0, dbg); return; }
    }
    if constexpr (Ge::W == 80 || Ge::W == 40) {
        if (mode == 3) { hipLaunchKernelGGL((k_fb_fast<Ge, false, 3, false>), g, t, 0, stream, R, fin, fout, ftmp, mag, flags, pairdiff, np, nstrips, ow, zero_first, dbg); return; }
    }
    hipLaunchKernelGGL((k_fb_fast<Ge, false>), g, t, 0, stream, R, fin, fout, ftmp, mag, flags, pairdiff, np, nstrips, ow, zero_first, dbg);
}

}  // namespace

// Blur iterations of one pyramid level for `np` pairs: R = polynomial expansions of np + 1 frames ([frame][y][x][5]), flows planar
// [pair][2][y][x].  mode (see launch_fast): 0 one iteration flow_in -> flow_out (different buffers); 1 / 2 the same with flow_in = the
// previous (coarser) level's final flow [pair][2][w/2][w/2], resized on the fly (1: 320 px, by the chain wave; 2: 160 / 80 px, in a
// prologue, through flow_tmp); 3 / 4 all three iterations in one launch (80 / 40 px: a pair is one workgroup), result in flow_out,
// flow_tmp as the second buffer (4: behind the prologue).
// mag_out (320-px level, last iteration; else null): float[pair][320][320] receives |flow|
// flags (may be null): int[np]; bit k of flags[p] is set when level k (0 = 320 px) of pair p met the solver's ill-posedness criterion, bit 4 + k when it met
// the border-sign criterion; a pair whose word is already non-zero is skipped.  pairdiff (may be null): int[np][kPairDiffTiles], non-zero where frame p
// differs from frame p + 1 (k_pyramid_all): bit-identical pairs are exempt from the border-sign criterion
int launch_fb_fast(avd_ctx* ctx, hipStream_t stream, int w, const float* R, const float* flow_in, float* flow_out, float* flow_tmp, float* mag_out,
                   int* flags, const int* pairdiff, int np, int zero_first, int mode)
{
    if (np <= 0) return 0;
    const bool ok = mode == 0 || (mode == 1 && w == 320) || (mode == 2 && (w == 160 || w == 80)) || (mode == 3 && (w == 80 || w == 40)) || (mode == 4 && w == 80);
    if (!ok) { ctx->err = "launch_fb_fast: this mode does not exist at this level size"; return AVD_ERR_ARG; }
    if (flow_in == flow_out || (mode >= 2 && (!flow_tmp || flow_tmp == flow_out))) { ctx->err = "launch_fb_fast: the flow is not updated in place"; return AVD_ERR_ARG; }
    switch (w) {
    case 320:
        // (neighbour-shared gathers -- each lane loads its left pixel, the right one comes from lane + 1 by a DPP wave shift -- were built,
        // bit-identical and slower, 164 us per launch against 142: profiles/r04_experiments.md section 2; the code is in the history, commit 3e8b43a)
        launch_fast<FGeo<320, 3, 2, 2, 1>>(stream, R, flow_in, flow_out, flow_tmp, mag_out, flags, pairdiff, np, 2, 160, zero_first, mode);
        break;
    case 160:
        // two shapes (ctx->fb_wide160; default 2 = chosen per call, see below): a pair as ONE strip of three blocks with the 320-px level's wave mix (119 workgroups of 12 waves,
        // 58 us per launch) or as two 80-column strips of 14 waves (238 workgroups, 48 us per launch).  The narrow shape finishes a launch sooner
        // (one clip alone: -25 us); the wide one costs fewer CU-microseconds (119 x 58 against 238 x 48: lanes 91 % instead of 73 % on image
        // columns, 14 halo columns per pair instead of 28) and that is what counts with clips in flight: +2.4 % frames/s.  The two differ in the
        // grouping of the solver's window sums (four columns per lane against two): bit-identical on well-posed content, like the 320-px level.
        // 2 = choose per call: this call is being enqueued and not yet counted, so > 0 means SOMEBODY ELSE's kernels will share the chip with it
        ctx->fb_wide160_used = ctx->fb_wide160 == 1 || (ctx->fb_wide160 == 2 && avd_calls_in_flight() - ctx->counted_in_flight > 0);
        if (ctx->fb_wide160_used) launch_fast<FGeo<160, 3, 2, 2, 1>>(stream, R, flow_in, flow_out, flow_tmp, nullptr, flags, pairdiff, np, 1, 160, zero_first, mode);
        else launch_fast<FGeo<160, 2, 1, 4, 2>>(stream, R, flow_in, flow_out, flow_tmp, nullptr, flags, pairdiff, np, 2, 80, zero_first, mode);
        break;
    case 80: launch_fast<FGeo<80, 2, 1, 4, 2>>(stream, R, flow_in, flow_out, flow_tmp, nullptr, flags, pairdiff, np, 1, 80, zero_first, mode); break;
    case 40: launch_fast<FGeo<40, 1, 1, 4, 2>>(stream, R, flow_in, flow_out, flow_tmp, nullptr, flags, pairdiff, np, 1, 40, zero_first, mode); break;
    default: ctx->err = "launch_fb_fast: unsupported level size"; return AVD_ERR_ARG;
    }
    HIP_TRY(ctx, hipGetLastError());
#ifdef AVD_FBF_DEBUG
    static const int stamps_w = [] { const char* e = std::getenv("AVD_FBF_STAMPS"); return e ? std::atoi(e) : 0; }();
    static int printed = 0;
    if (stamps_w == w && printed++ == 10) {
        long long h[16][3];
        (void)hipStreamSynchronize(stream);
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fbf_stamps), sizeof(h)) == hipSuccess)
            for (int i = 0; i < 16; i++)
                if (h[i][0])
                    fprintf(stderr, "fbfast stamps w=%d wave %2d role %lld: total %lld cycles, at barriers %lld (%.1f %%)\n", w, i, h[i][2], h[i][0],
                            h[i][1], 100. * h[i][1] / h[i][0]);
    }
#endif
    return 0;
}
