// avd_fbfused.hip -- FarnebackUpdateFlow_Blur (winsize 15) with ALL iterations of a pyramid level in one launch and
// nothing but the flow leaving the chip (gfx950).
//
// Replaces, per level, the pair k_uv / k_uvp + k_hscan of avd_farneback.hip (reference site:
// cv2.calcOpticalFlowFarneback(prev, cur, None, 0.5, 3, 15, 3, 5, 1.2, 0), app/analyzers/video.py:45).  Those two
// kernels exchange the exact double intermediate D = vsum(x+7) - vsum(x-8) through HBM (40 B per pixel written and
// read again: 488 MB per iteration at 320x320 x 119 pairs, more than half of the stage's traffic), because the
// vertical running sums want lanes along x and the horizontal ones lanes along y.  Here ONE workgroup owns ONE pair
// and the transpose happens in LDS, G image rows (one "group") per workgroup barrier:
//
//   V  waves 0..NVW-1 (lanes along x, 64 columns each, sequential in y): normal equations of the entering row
//      (software-pipelined loads: flow/R0 three rows ahead, bilinear gather of R1 one row ahead), the 15-row history
//      of the box filter in REGISTERS (a 16-slot ring, statically indexed: the row loop is unrolled 16 times), the
//      five running double sums; writes the vsum row into the group buffer being filled.
//   S  one wave, sequential in x: the horizontal running sums, IN PLACE (vsum -> g).  A lane owns a (segment, row,
//      channel): the line is cut into K column segments that are scanned at the same time, segment k working on the
//      group that segment k-1 scanned one barrier earlier (a systolic chain: the running sum and the 16-deep delay
//      line of a line are handed to the next lane group through LDS).  LDS holds only ~12 image rows, so this is what
//      gives the dependent chain enough lanes: K x G x 5.  Each vsum value is read from LDS exactly once (16 bytes =
//      two columns per read) and kept in a register delay line (entering column x+7, leaving column x-8); the
//      subtractions of the NEXT 16 columns are interleaved with the dependent adds of the current ones.
//   X  two waves, lanes along x again: read g, 2x2 solve in double, coalesced flow stores; between barriers they also
//      touch the rows the vertical waves will need ~20 rows later (plain loads whose values are discarded), so that
//      those find their operands in L2: a vertical wave has no registers left for a deeper pipeline of its own.
//
// K + 2 group buffers rotate (filled, scanned by segment 0 .. K-1, solved); one workgroup barrier per group, no
// polling, no cross-workgroup dependency.  LDS at W = 320: 6 buffers x 2 rows x 5 channels x 322 doubles + hand-over
// = 160,320 B: one workgroup per CU.  The iterations of a level run inside the launch (a workgroup owns its pair's
// flow; its stores are drained before the barrier that precedes the next iteration's loads, and the CU's L1 is
// invalidated).  Every double operation happens in cv2's order (the chains are literal), so results are bit-identical
// to the two-kernel path and to oracle/avd_oracle.c.
//
// HBM traffic per pair and iteration: R0 + R1 + flow in + flow out = 56 B per pixel instead of 136 B.  The kernel is
// bound by VALU issue of the normal equations (~180 wave-instructions per row and wave), not by memory.
#include <cstdio>
#include <cstdlib>
#include "avd_internal.h"
#include "avd_fb_device.h"

#pragma clang fp contract(off)

namespace {

constexpr int kM = 7;                 // (winsize - 1) / 2

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef float flt4 __attribute__((ext_vector_type(4)));

// W: level size; K: column segments scanned concurrently; G: image rows per group (= per barrier)
template <int W, int K, int G>
struct Geo {
    static constexpr int H = W;
    static constexpr int NVW = (W + 63) / 64;          // vertical waves
    static constexpr int NWAVES = NVW + 3;             // + scanner + two solvers
    static constexpr int P = W + 2;                    // doubles per (row, channel) line: 2P mod 64 = 4 -> the scanner's
                                                       // 16-byte accesses (lane stride = one line) spread over the banks
    static constexpr int GROUP = G * 5 * P;            // doubles per group buffer
    static constexpr int NBUF = K + 2;                 // being filled, scanned by segment 0 .. K-1, solved
    static constexpr int NT = H / G;                   // groups per iteration
    static constexpr int NBAR = NT + K + 1;            // workgroup barriers per iteration (every wave executes all)
    static constexpr int NE = H + kM;                  // entries of the vertical pass: image row min(e, H-1)
    static constexpr int SEG = W / K;                  // columns per scanner segment
    static constexpr int NBODY = (SEG + 15) / 16;      // 16-column bodies per segment (the last may have 8 columns)
    static constexpr int LINES = G * 5;                // (row, channel) lines per group
    static constexpr int SLANES = K * LINES;           // active scanner lanes
    static constexpr int HAND = 18;                    // doubles handed from a segment to the next: 16 delay line + sum (+ pad)
    static constexpr int LDS_DOUBLES = NBUF * GROUP + SLANES * HAND;
    static_assert(H % G == 0 && 16 % G == 0, "rows per group: 1, 2, 4, 8");
    static_assert(W % K == 0 && (K == 1 || SEG % 16 == 0), "segments are whole 16-column bodies");
    static_assert(SLANES <= 64 && LDS_DOUBLES * 8 <= 163840, "one wave scans; 160 KiB of LDS");
};

// workgroup barrier; debug builds (AVD_FB_DEBUG) account the cycles a wave spends waiting at it
#ifdef AVD_FB_DEBUG
__device__ long long g_fb_stamps[8][3];                  // [wave][total cycles, cycles at barriers, barriers] of workgroup 0
#define FB_BARRIER()                                                      \
    do {                                                                  \
        const long long t0__ = __builtin_amdgcn_s_memtime();              \
        __syncthreads();                                                  \
        fb_wait += __builtin_amdgcn_s_memtime() - t0__;                   \
        fb_nbar++;                                                        \
    } while (0)
#else
#define FB_BARRIER() __syncthreads()
#endif

__device__ __forceinline__ unsigned next_buf(unsigned off, unsigned group, unsigned nbuf)
{
    return off + group == nbuf * group ? 0u : off + group;
}

// ------------------------------------------------------------------------------------------------------------------
// V: one wave, columns 64*vw .. 64*vw+63 of pair p.
// ------------------------------------------------------------------------------------------------------------------
template <int W, int K, int G>
__device__ __forceinline__ void role_vertical(const float* __restrict__ R, const float* __restrict__ flow,
                                              double* __restrict__ buf, int p, int vw, int lane, int dbg, long long& fb_wait, int& fb_nbar,
                                              bool zf)
{
    using Ge = Geo<W, K, G>;
    constexpr int H = W, plane = W * H, NE = Ge::NE;
    constexpr int NB = NE / 16, TAIL = NE - NB * 16;
    const int xl = vw * 64 + lane;
    const bool act = xl < W;
    const int x = act ? xl : W - 1;                    // idle lanes of the last wave run a duplicate chain, never write
    const unsigned r0base = (unsigned)p * 5u * plane, r1base = r0base + 5u * plane, flbase = (unsigned)p * 2u * plane;
    auto row_of = [](int e) { return e < H - 1 ? e : H - 1; };

    NeIn in[4];
    NeG2 g[2];
    const float sx = border_factor(x, W);                 // x part of the border attenuation: a per-lane constant
    float ring[16][5];                                  // ring[e & 15] = normal-equation row of entry e
    double vs[5] = {0., 0., 0., 0., 0.};
    unsigned goff = 0;                                  // group buffer being filled (offset in doubles)
#pragma unroll
    for (int k = 0; k < 3; k++) ne_load(R, flow, r0base, flbase, x, row_of(k), W, plane, in[k]);
    ne_gather2(R, r1base, in[0], x, row_of(0), W, H, g[0], zf);

    // one entry: evaluate, refill the prefetch slots, update the running sums, publish the vsum row
    auto step = [&](int e, int kk, bool first, bool refill_g, bool refill_in) __attribute__((always_inline)) {
        // Refills FIRST: the gather of the next entry goes into the slot the previous entry released (this entry's is
        // the other one) and the inputs three entries ahead into the slot of entry e - 1, so nothing this entry still
        // needs is overwritten -- and the gather gets a whole row of lead instead of the tail of one (issued after the
        // arithmetic it had ~300 cycles before its use at the top of the next step: less than an L2 round trip).
        if (refill_g && !(dbg & 4)) ne_gather2(R, r1base, in[(kk + 1) & 3], x, row_of(e + 1), W, H, g[(kk + 1) & 1], zf);
        if (refill_in && !(dbg & 4)) ne_load(R, flow, r0base, flbase, x, row_of(e + 3), W, plane, in[(kk + 3) & 3]);
        __builtin_amdgcn_sched_barrier(0);
        float a[5];
        ne_finish2(in[kk & 3], g[kk & 1], x, row_of(e), W, H, sx, border_factor(row_of(e), H), a, zf);
        if (first && kk == 0) {
#pragma unroll
            for (int c = 0; c < 5; c++) vs[c] = (double)(a[c] * (float)(kM + 2));
        } else if (first && kk < kM) {
#pragma unroll
            for (int c = 0; c < 5; c++) vs[c] += (double)a[c];
        } else {
            // leaving row y - 8 = entry e - 15 (row 0 while the window still touches the top edge)
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const float b = first ? ring[0][c] : ring[(kk + 1) & 15][c];
                vs[c] += (double)(a[c] - b);
            }
        }
#pragma unroll
        for (int c = 0; c < 5; c++) ring[kk & 15][c] = a[c];
        if (!first || kk >= kM) {
            const int yr = (kk + 16 - kM) % G;          // (e - 7) % G: 16 is a multiple of G
            if (act) {
#pragma unroll
                for (int c = 0; c < 5; c++) buf[goff + (unsigned)((yr * 5 + c) * Ge::P) + (unsigned)xl] = vs[c];
            }
            if (yr == G - 1) {
                FB_BARRIER();
                goff = next_buf(goff, Ge::GROUP, Ge::NBUF);
            }
        }
    };

    for (int eb = 0; eb < NB * 16; eb += 16) {
        const bool first = eb == 0;                     // uniform: selects between VALU-only variants of the sums
#pragma unroll
        for (int kk = 0; kk < 16; kk++) step(eb + kk, kk, first, true, true);
    }
#pragma unroll
    for (int kk = 0; kk < TAIL; kk++)
        step(NB * 16 + kk, kk, NB == 0, NB * 16 + kk + 1 < NE, NB * 16 + kk + 3 < NE);
#pragma unroll
    for (int i = 0; i < K + 1; i++) FB_BARRIER();
}

// ------------------------------------------------------------------------------------------------------------------
// S: horizontal running sums, in place.  Lane = seg * LINES + row * 5 + channel.
//   g(-1) = 9 * vs[0] + vs[1] + ... + vs[6];   g(x) = g(x-1) + (vs[min(x+7, W-1)] - vs[max(x-8, 0)])
// with e(j) = vs[clamp(j + 7)] the entering value of column j: g(x) = g(x-1) + (e(x) - e(x-15)); dl[j & 15] = e(j).
// ------------------------------------------------------------------------------------------------------------------
template <int W, int K, int G>
__device__ __forceinline__ void scan_segment(double* __restrict__ q, double* __restrict__ hand_in,
                                             double* __restrict__ hand_out, int seg)
{
    using Ge = Geo<W, K, G>;
    constexpr int HP = W / 2;                             // aligned pairs per line
    constexpr int SEG = Ge::SEG, NBODY = Ge::NBODY;
    constexpr int NLAST = SEG - 16 * (NBODY - 1);         // columns of a segment's last body: 16, or 8 at W = 40
    const int x0 = seg * SEG;
    dbl2* Q = reinterpret_cast<dbl2*>(q) + x0 / 2;        // pair 0 = columns x0, x0+1
    const bool lastseg = seg == K - 1;
    double gs, dl[16];
    dbl2 carry;                                           // pair 3: its .y is vs[x0 + 7] = e(x0)
    if (seg == 0) {
        const dbl2 a0 = Q[0], a1 = Q[1], a2 = Q[2], a3 = Q[3];
        gs = a0.x * (double)(kM + 2);
        gs += a0.y; gs += a1.x; gs += a1.y; gs += a2.x; gs += a2.y; gs += a3.x;
#pragma unroll
        for (int s = 1; s <= 8; s++) dl[s] = a0.x;        // e(-15 .. -8) = vs[0]
        dl[9] = a0.x; dl[10] = a0.y; dl[11] = a1.x; dl[12] = a1.y; dl[13] = a2.x; dl[14] = a2.y; dl[15] = a3.x;
        dl[0] = 0.;                                       // e(-16): never read
        carry = a3;
    } else {
        // state of this line at the end of the previous segment (written one barrier ago by the lane LINES below)
        const dbl2* hin = reinterpret_cast<const dbl2*>(hand_in);
#pragma unroll
        for (int t = 0; t < 8; t++) { const dbl2 v = hin[t]; dl[2 * t] = v.x; dl[2 * t + 1] = v.y; }
        gs = hand_in[16];
        carry = Q[3];                                     // columns x0+6, x0+7: not yet overwritten
    }
    // pairs 4 + 8b .. 11 + 8b of body b (index clamped to the last pair of the line)
    auto fetch = [&](dbl2 (&dst)[8], int b) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int k = 4 + 8 * b + t;                  // relative to x0 / 2
            const int lim = HP - 1 - x0 / 2;
            dst[t] = Q[k < lim ? k : lim];
        }
    };
    // entering values e(x0 + 16b + kk), kk = 0..15, of body b from its eight pairs and the carried one.  Pairs at or
    // beyond the end of the line repeat vs[W-1]: their fetch index was clamped to the last pair (vs[W-2], vs[W-1]).
    // Only the last body of the last segment can reach the end of the line.
    auto entering = [&](const dbl2 (&src)[8], double (&e)[16], int b) __attribute__((always_inline)) {
        e[0] = carry.y;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            dbl2 v = src[t];
            if (4 + 8 * b + t >= SEG / 2 && lastseg) v = dbl2{v.y, v.y};
            e[2 * t + 1] = v.x;
            if (t < 7) e[2 * t + 2] = v.y;
        }
        carry = src[7];
    };
    dbl2 pr[2][8];
    double d[16];                                         // differences: consumed by the chain and refilled in place
    fetch(pr[0], 0);
    if (NBODY > 1) fetch(pr[1], 1);
    {
        double e[16];
        entering(pr[0], e, 0);
#pragma unroll
        for (int kk = 0; kk < (NBODY == 1 ? NLAST : 16); kk++) {
            const double l = dl[(kk + 1) & 15];
            dl[kk] = e[kk];
            d[kk] = e[kk] - l;
        }
    }
#pragma unroll
    for (int b = 0; b < NBODY; b++) {
        const int nsteps = b == NBODY - 1 ? NLAST : 16;
        const int nnext = b + 1 < NBODY ? (b + 1 == NBODY - 1 ? NLAST : 16) : 0;
        double e[16], o[16];
        if (b + 1 < NBODY) entering(pr[(b + 1) & 1], e, b + 1);
        if (b + 2 < NBODY) fetch(pr[b & 1], b + 2);       // its previous contents went into d: free
        // the chain of this body, ONE dependent add per column, with the independent subtractions of the next body
        // in its shadow (as "sub, add, sub, add" of the same column every instruction would wait for the one before)
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            if (kk < nsteps) { gs += d[kk]; o[kk] = gs; }
            if (kk < nnext) {
                const double l = dl[(kk + 1) & 15];
                dl[kk] = e[kk];
                d[kk] = e[kk] - l;
            }
            if ((kk & 1) && kk < nsteps) Q[8 * b + (kk >> 1)] = dbl2{o[kk - 1], o[kk]};
        }
    }
    if (K > 1 && !lastseg) {
        dbl2* hout = reinterpret_cast<dbl2*>(hand_out);
#pragma unroll
        for (int t = 0; t < 8; t++) hout[t] = dbl2{dl[2 * t], dl[2 * t + 1]};
        hand_out[16] = gs;
    }
}

template <int W, int K, int G>
__device__ __forceinline__ void role_scan(double* __restrict__ buf, double* __restrict__ hand, int lane, int dbg, long long& fb_wait, int& fb_nbar)
{
    using Ge = Geo<W, K, G>;
    const bool on = lane < Ge::SLANES && !(dbg & 1);
    const int seg = on ? lane / Ge::LINES : 0, line = on ? lane - seg * Ge::LINES : 0;
    // after barrier j segment k works on group j - k, which lives in buffer (j - k) mod NBUF
    int grp = -seg;
    unsigned goff = seg == 0 ? 0u : (unsigned)((Ge::NBUF - seg) % Ge::NBUF) * Ge::GROUP;     // buffer of group -seg (mod NBUF)
    for (int j = 0; j < Ge::NT + K - 1; j++) {
        FB_BARRIER();
        if (on && grp >= 0 && grp < Ge::NT)
            scan_segment<W, K, G>(buf + goff + (unsigned)line * Ge::P, hand + (seg ? lane - Ge::LINES : 0) * Ge::HAND,
                                  hand + lane * Ge::HAND, seg);
        grp++;
        goff = next_buf(goff, Ge::GROUP, Ge::NBUF);
    }
    FB_BARRIER();
    FB_BARRIER();
}

// ------------------------------------------------------------------------------------------------------------------
// X: 2x2 solve per pixel (double, cv2's operation order), flow stores.  Solver xi takes every second (row, block).
// Between barriers it also touches, with plain 16-byte loads whose values are discarded, the rows of R0 / R1 / flow
// that the vertical waves will load ~20 rows later.
// ------------------------------------------------------------------------------------------------------------------
template <int W, int K, int G>
__device__ __forceinline__ void role_solve(const float* __restrict__ R, const double* __restrict__ buf,
                                           float* __restrict__ flow, int p, int xi, int lane, int dbg, long long& fb_wait, int& fb_nbar)
{
    using Ge = Geo<W, K, G>;
    constexpr int H = W;
    constexpr unsigned plane = W * H;
    constexpr int PF_ROWS = 18;                           // distance of the touch loads ahead of the group being filled
    const double scale = 1. / (15 * 15);
    float* fl = flow + (size_t)p * 2 * plane;
    // solver 0 touches R0 and the x flow plane, solver 1 R1 and the y plane
    const char* rsrc = reinterpret_cast<const char*>(R + ((size_t)p + xi) * 5 * plane);
    const char* fsrc = reinterpret_cast<const char*>(fl + (size_t)xi * plane);
    // Touch loads: ONE dword per 128-byte line (a wave instruction covers 64 lines = 8 KiB of the buffer).  The point of a
    // touch is the HBM / Infinity Cache -> L2 fetch of the line; what comes back to the CU is thrown away, and a CU ingests
    // only ~40-60 GB/s (tools/ldsdma_bench.hip) -- of which this kernel's vertical waves need ~40: whole-row touches
    // (16 bytes per lane) took a third of that path for bytes nobody uses.
    constexpr int RL = (G * W * 20 + 8191) / 8192, FL = (G * W * 4 + 8191) / 8192;
    // touch loads in flight: issued in one barrier interval, retired (values discarded) TWO intervals later -- an
    // interval (~1.5 us at W = 320) is shorter than a loaded HBM round trip, and the solver must never wait for them
    float tv[2][RL + FL];
#pragma unroll
    for (int i = 0; i < 2 * (RL + FL); i++) tv[i / (RL + FL)][i % (RL + FL)] = 0.f;
    auto touch_retire = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RL + FL; i++) asm volatile("" ::"v"(tv[par][i]));
    };
    auto touch_issue = [&](int j, int par) __attribute__((always_inline)) {
        // V is filling group j + 1 now; rows [y0, y0 + G) enter its box filter PF_ROWS rows later.  Plain loads on purpose:
        // as non-temporal loads (no L1 allocation, streaming hint) the level kernel took 0.95 instead of 0.87 ms
        const int y0 = G * (j + 1) + kM + PF_ROWS;
        if (y0 >= H || (dbg & 8)) return;
        const int rows = y0 + G <= H ? G : H - y0;
        const unsigned rbytes = (unsigned)rows * W * 20u, fbytes = (unsigned)rows * W * 4u;
        const unsigned rbeg = (unsigned)y0 * W * 20u, fbeg = (unsigned)y0 * W * 4u;
#pragma unroll
        for (int i = 0; i < RL; i++) {
            const unsigned o = (unsigned)i * 8192u + (unsigned)lane * 128u;
            tv[par][i] = *reinterpret_cast<const float*>(rsrc + rbeg + (o < rbytes ? o : rbytes - 4u));
        }
#pragma unroll
        for (int i = 0; i < FL; i++) {
            const unsigned o = (unsigned)i * 8192u + (unsigned)lane * 128u;
            tv[par][RL + i] = *reinterpret_cast<const float*>(fsrc + fbeg + (o < fbytes ? o : fbytes - 4u));
        }
    };
    unsigned goff = 0;
    // interval j (after barrier j): solve group j - K (every segment of it has been scanned), then touch
    auto interval = [&](int j, int par) __attribute__((always_inline)) {
        FB_BARRIER();
        touch_retire(par);
        const int g = j - K;
        if (g >= 0) {
#pragma unroll
            for (int item = 0; item < G * Ge::NVW; item++) {
                if ((item & 1) != xi || (dbg & 2)) continue;
                const int r = item / Ge::NVW, b = item % Ge::NVW;
                const int x = b * 64 + lane;
                if (x < W) {
                    const double* q = buf + goff + (unsigned)(r * 5) * Ge::P + (unsigned)x;
                    const double g11 = q[0] * scale, g12 = q[Ge::P] * scale, g22 = q[2 * Ge::P] * scale;
                    const double h1 = q[3 * Ge::P] * scale, h2 = q[4 * Ge::P] * scale;
                    const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                    const unsigned o = (unsigned)((g * G + r) * W + x);
                    fl[o] = (float)((g11 * h2 - g12 * h1) * idet);
                    fl[o + plane] = (float)((g22 * h1 - g12 * h2) * idet);
                }
            }
            goff = next_buf(goff, Ge::GROUP, Ge::NBUF);
            // the stores of this group are complete (acknowledged by L2) before the next barrier: the next iteration's
            // loads of the vertical waves, which come after a later barrier, are ordered behind them.  The touch
            // loads are issued AFTER this wait and stay in flight across barriers.
            __builtin_amdgcn_s_waitcnt(0);
        }
        touch_issue(j, par);
    };
    constexpr int NI = Ge::NT + K;                         // intervals with work; then the last barrier
    int j = 0;
    for (; j + 2 <= NI; j += 2) {
        interval(j, 0);
        interval(j + 1, 1);
    }
    if (NI & 1) interval(j, 0);
    FB_BARRIER();
    touch_retire(0);
    touch_retire(1);
}

// All iterations of one pyramid level for pair p, executed by the first Geo::NWAVES waves of the workgroup; further waves (the
// re-run kernel below is launched with the widest level's wave count) only keep the barrier count.
template <int W, int K, int G>
__device__ __forceinline__ void level_body(const float* __restrict__ R, float* __restrict__ flow, double* __restrict__ lds, int p, int iterations,
                                           int dbg, int zero_first, int lane, int wave)
{
    using Ge = Geo<W, K, G>;
    double* buf = lds;
    double* hand = lds + Ge::NBUF * Ge::GROUP;
    long long fb_wait = 0;                                // debug builds: cycles at barriers (unused otherwise)
    int fb_nbar = 0;
#ifdef AVD_FB_DEBUG
    const long long fb_t0 = __builtin_amdgcn_s_memtime();
#endif
    // every role runs its own loop over the iterations (one common loop around the role dispatch lets the compiler
    // hoist the invariants of ALL roles in front of it, and the sum of their registers spills)
    // Waves w and w + 4 of a workgroup share a SIMD.  The scanner (a chain of dependent double adds, ~3 instructions
    // per column) must not share its SIMD with a vertical wave, which always has an instruction ready and takes
    // every other issue slot: wave 3 scans, the vertical waves are the first NVW of the others (0, 1, 2, 4, 5 at
    // W = 320: two SIMDs get a pair of them), the solvers the rest (6 and 7: one beside a vertical wave, one
    // beside the scanner).
    const int smap = Ge::NWAVES > 3 ? 3 : Ge::NVW;          // scanner's wave
    const int ridx = wave < smap ? wave : wave - 1;        // index among the non-scanner waves
    if (wave >= Ge::NWAVES) {
        for (int i = 0; i < iterations * Ge::NBAR; i++) FB_BARRIER();
    } else if (wave != smap && ridx < Ge::NVW) {
        // two SIMDs carry two V waves each (waves w and w + 4); issue arbitration favours the OLDER wave, so the younger
        // one of a pair would be the pole of every barrier interval (barrier-wait stamps: 7 % vs 20-27 % of its partner):
        // a static priority for the younger ones evens the pair out
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
        for (int it = 0; it < iterations; it++) {
            // flow rows cached in this CU's L1 during the previous iteration are stale now
            if (it > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            role_vertical<W, K, G>(R, flow, buf, p, ridx, lane, dbg, fb_wait, fb_nbar, zero_first && it == 0);
        }
        if (wave >= 4) __builtin_amdgcn_s_setprio(0);
    } else if (wave == smap) {
        __builtin_amdgcn_s_setprio(3);                    // its operands are rarely ready: take the slot when they are
        for (int it = 0; it < iterations; it++) role_scan<W, K, G>(buf, hand, lane, dbg, fb_wait, fb_nbar);
        __builtin_amdgcn_s_setprio(0);
    } else {
        for (int it = 0; it < iterations; it++) role_solve<W, K, G>(R, buf, flow, p, ridx - Ge::NVW, lane, dbg, fb_wait, fb_nbar);
    }
#ifdef AVD_FB_DEBUG
    if (W == 320 && p == 0 && lane == 0) {
        g_fb_stamps[wave][0] = __builtin_amdgcn_s_memtime() - fb_t0;
        g_fb_stamps[wave][1] = fb_wait;
        g_fb_stamps[wave][2] = fb_nbar;
    }
#endif
}

template <int W, int K, int G>
__global__ __launch_bounds__((64 * Geo<W, K, G>::NWAVES)) void k_fb_level(const float* __restrict__ R, float* __restrict__ flow,
                                                                        int npairs, int iterations, int dbg_arg, int zero_first, const int* __restrict__ plist)
{
#ifdef AVD_FB_DEBUG
    const int dbg = dbg_arg;                              // timing experiments (AVD_FB_DBG), see launch_fb_level
#else
    constexpr int dbg = 0;
#endif
    using Ge = Geo<W, K, G>;
    __shared__ __align__(16) double lds[Ge::LDS_DOUBLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // consecutive pairs share a frame (pair p gathers as R1 what pair p+1 reads as R0): an XCD (one L2) gets a
    // contiguous run of pairs; workgroups are dealt round-robin to the 8 XCDs
    const int ppx = (npairs + 7) >> 3;
    const int pi = (blockIdx.x & 7) * ppx + (blockIdx.x >> 3);
    if (pi >= npairs) return;                             // whole workgroup
    const int p = plist ? plist[pi] : pi;                 // (exact re-run of the flagged pairs of the fast mode: a compacted list)
    level_body<W, K, G>(R, flow, lds, p, iterations, dbg, zero_first, lane, wave);
}

template <int W, int K, int G>
void launch_one(hipStream_t stream, int grid, const float* R, float* flow, int np, int iterations, int dbg, int zero_first, const int* plist)
{
    hipLaunchKernelGGL((k_fb_level<W, K, G>), dim3(grid), dim3(64 * Geo<W, K, G>::NWAVES), 0, stream, R, flow, np, iterations, dbg, zero_first, plist);
}

}  // namespace

// one launch = all iterations of one pyramid level for `np` pairs; R = polynomial expansions of np + 1 frames
// ([frame][y][x][5]), flow planar [pair][2][y][x] (initial flow in, final flow out); plist (may be null): the pairs are plist[0 .. np)
int launch_fb_level(avd_ctx* ctx, hipStream_t stream, int w, const float* R, float* flow, int np, int iterations, int zero_first, const int* plist)
{
    if (np <= 0) return 0;
    const int grid = 8 * ((np + 7) / 8);
    // timing experiments only (results are wrong when set): 1 no scan, 2 no solve / stores, 4 no refill loads, 8 no touch loads
    static const int dbg = [] { const char* e = std::getenv("AVD_FB_DBG"); return e ? std::atoi(e) : 0; }();
    static const int var = [] { const char* e = std::getenv("AVD_FB_VARIANT"); return e ? std::atoi(e) : 0; }();
    switch (w) {
    case 320:
        if (var == 1) launch_one<320, 2, 2>(stream, grid, R, flow, np, iterations, dbg, zero_first, plist);      // A/B: two segments (slower)
        else launch_one<320, 4, 2>(stream, grid, R, flow, np, iterations, dbg, zero_first, plist);
        break;
    case 160: launch_one<160, 2, 4>(stream, grid, R, flow, np, iterations, dbg, zero_first, plist); break;
    case 80: launch_one<80, 1, 4>(stream, grid, R, flow, np, iterations, dbg, zero_first, plist); break;
    case 40: launch_one<40, 1, 4>(stream, grid, R, flow, np, iterations, dbg, zero_first, plist); break;
    default: ctx->err = "launch_fb_level: unsupported level size"; return AVD_ERR_ARG;
    }
    HIP_TRY(ctx, hipGetLastError());
#ifdef AVD_FB_DEBUG
    static const bool stamps = std::getenv("AVD_FB_STAMPS") != nullptr;
    static int printed = 0;
    if (stamps && w == 320 && printed++ == 6) {
        long long h[8][3];
        (void)hipStreamSynchronize(stream);
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fb_stamps), sizeof(h)) == hipSuccess)
            for (int i = 0; i < 8; i++)
                fprintf(stderr, "fb stamps wave %d: total %lld cycles, at barriers %lld (%.1f %%), %lld barriers\n", i, h[i][0], h[i][1],
                        h[i][0] ? 100. * h[i][1] / h[i][0] : 0., h[i][2]);
    }
#endif
    return 0;
}
