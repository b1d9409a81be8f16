// avd_cnn.hip -- ResNet-50-style CNN forward on the matrix cores (gfx950): SURVEY.md section 8 row A9.
//
// BUILD-DEFINED EXTENSION: the reference has no learned model (its per-frame "model" is the
// closed form of app/analyzers/video.py:54-56).  BASELINE.json's north_star names a "CNN (ResNet-50-style) forward" on
// MFMA; this file is that stage with caller-supplied (seeded random) weights, gated off from ai_score: nothing in the
// parity path calls it.  Its oracle is a float32 restatement with the same bf16 roundings (tests/test_cnn.py).
//
// Topology: 7x7/2 stem (64) + ReLU, 3x3/2 max pool, bottleneck stages [3, 4, 6, 3] of widths 64/128/256/512 (x4 out,
// stride on the 3x3), global average pool, 2048 -> 1000 linear: 53 convolutions, 4.1 GMAC per 224 x 224 frame.  Batch
// norm is taken as folded into the weights and a per-channel bias.
//
// DATA LAYOUT IN HBM: an activation is the matrix [pixels = n * H * W][channels], bf16, in the blocked + swizzled operand
// layout of avd_mfma_device.h (1-KiB blocks of 16 pixels x 32 channels), preceded by one KiB of zeros.  Every
// convolution is ONE implicit GEMM  out[pixel][cout] = sum_{tap, c} in[pixel shifted by tap][c] * W[cout][tap][c] :
//  * the weight operand [cout][K = taps * cin] is re-tiled once on upload: an LDS-DMA instruction copies one KiB block;
//  * the activation operand is GATHERED by the LDS-DMA itself: lane -> (tile row, 16-byte chunk), so a lane's source
//    address is "the 8 channels I need of the input pixel this output pixel sees through tap (dy, dx)"; taps that fall
//    outside the image read the zero page.  No im2col matrix is ever written -- the 7x7 stem gathers pixel pairs from a
//    zero-bordered copy of the input image instead (MODE 1 below);
//  * the epilogue adds bias (+ the residual, read in the same layout), applies ReLU, rounds to bf16 and stores 16 bytes
//    per lane straight into the blocked layout of the NEXT layer's operand.
// Tile: 256 pixels x (64 | 128 | 256) output channels per 512-thread workgroup, K in half stages of 32, ring of four half
// stages with counted vmcnt + raw s_barrier (the scheme of avd_vit.hip; K is a run-time value here).
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "avd_internal.h"
#include "avd_mfma_device.h"

namespace {

using namespace avd_mfma;

constexpr int kSide = 224;
constexpr int kZeroPage = 512;                  // elements (1 KiB) of zeros in front of every activation
constexpr int kImgSide = 232;                   // the stem reads a 224 x 224 image with a zero border: pixel (y, x) at (y + 3, x + 3)

struct ConvGeom {
    int hin, win, cin, hout, wout, cout, ksize, stride, pad;
    int m_out;                                  // n * hout * wout
    int nh;                                     // half stages: ksize * ksize * cin / 32
    int cpb;                                    // cin / 32
};

#define AVD_WAIT_VMC(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// MODE 0: the activation operand is a blocked activation, gathered per tap.  MODE 1 (the stem): the operand is the
// zero-bordered 224 x 224 input image [frame][232][232][4 channels, the fourth zero]; half stage ky holds the 8 x 4 = 32
// (kx, c) values of kernel row ky (kx = 7 and c = 3 carry zero weights), i.e. chunk j of a row = the two pixels
// (2 ox + 2 j, 2 ox + 2 j + 1) of image row 2 oy + ky in border coordinates: 16 contiguous, 16-byte aligned bytes.
template <int BM, int WAVES_M, int TI, int MODE, int KS>
__global__ __launch_bounds__(512) void k_conv_bf16(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                  const float* __restrict__ bias, const uint16_t* __restrict__ R,
                                                  uint16_t* __restrict__ Y, ConvGeom g, int relu)
{
    constexpr int WAVES_N = 8 / WAVES_M, TJ = 4, BN = WAVES_N * 64;
    static_assert(WAVES_M * TI * 16 == BM && (BM == 256 || BM == 128), "a workgroup covers 256 or 128 output pixels");
    constexpr int RA = BM / 8, QA = RA / 16;               // activation rows / LDS-DMA instructions per wave and half stage
    constexpr int RB = BN / 8;                              // weight rows a wave stages per half stage: 32, 16 or 8
    constexpr int QB = (RB + 15) / 16;
    constexpr int P = QA + QB;                              // LDS-DMA instructions per wave and half stage
    constexpr int HALF_A = BM * 64, HALF_B = BN * 64, STAGE = HALF_A + HALF_B;
    extern __shared__ __align__(16) char lds[];            // ring of four half stages [A | B]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = g.cout / BN, tiles_m = (g.m_out + BM - 1) / BM, total = tiles_m * tiles_n;
    // the column tiles of one pixel block run together on one XCD (the gathered activation rows are fetched once)
    const int per = (gridDim.x + 7) >> 3;
    const int lid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (lid >= total) return;
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;

    // ---- the activation rows this lane gathers per half stage (instruction q: tile row wave * RA + q * 16 + lane / 4)
    // MODE 0 keeps per row: the pixel index of tap (0, 0) (rin0), a 9-bit mask of the taps that fall inside the image, and
    // the chunk that belongs in this lane's LDS slot; a tap then costs one scalar offset and a dozen vector operations
    int rin0[QA], cw[QA];
    unsigned tapmask[QA];
    const int hw = g.hout * g.wout;
#pragma unroll
    for (int q = 0; q < QA; q++) {
        const int r = wave * RA + q * 16 + (lane >> 2), m = m0 + r;
        const bool valid = m < g.m_out;
        const int img = m / hw, rem = m - img * hw, oy = rem / g.wout, ox = rem - oy * g.wout;
        cw[q] = (lane & 3) ^ swz((r >> 2) & 3);             // the chunk that belongs in this lane's LDS slot
        if (MODE == 1) {                                     // byte offset of pixel (2 oy, 2 ox + 2 chunk) of the bordered image
            rin0[q] = valid ? ((img * kImgSide + 2 * oy) * kImgSide + 2 * ox + 2 * cw[q]) * 8 : 0;
            tapmask[q] = 0;
        } else {
            const int y0 = oy * g.stride - g.pad, x0 = ox * g.stride - g.pad;
            rin0[q] = (img * g.hin + y0) * g.win + x0;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < KS * KS; t++) {
                const int yi = y0 + t / KS, xi = x0 + t % KS;
                if (valid && (unsigned)yi < (unsigned)g.hin && (unsigned)xi < (unsigned)g.win) mk |= 1u << t;
            }
            tapmask[q] = mk;
        }
    }
    unsigned vob[QB];
#pragma unroll
    for (int q = 0; q < QB; q++) {
        const int r = wave * RB + q * 16 + (lane >> 2);
        vob[q] = (unsigned)((r >> 4) * g.nh * 1024 + (r & 15) * 64 + (lane & 3) * 16);
    }
    const char* xb = reinterpret_cast<const char*>(X);
    int is_cb = 0, is_dy = 0, is_dx = 0;                    // channel block and tap of the next half stage to be issued
    auto issue = [&](int hs) __attribute__((always_inline)) {
        char* st = lds + (hs & 3) * STAGE;
#pragma unroll
        for (int q = 0; q < QA; q++) {
            unsigned off;
            if (MODE == 1) {
                off = (unsigned)(rin0[q] + hs * (kImgSide * 8));  // kernel row ky = hs: one image row down
            } else {
                const int rin = rin0[q] + is_dy * g.win + is_dx;  // (the tap offset is uniform: scalar arithmetic)
                const bool ok = (tapmask[q] >> (is_dy * KS + is_dx)) & 1u;
                off = (unsigned)(kZeroPage * 2) + (unsigned)((rin >> 4) * g.cpb + is_cb) * 1024u + (unsigned)((rin & 15) * 64) +
                      (unsigned)((cw[q] ^ swz((rin >> 2) & 3)) << 4);
                off = ok ? off : (unsigned)(lane * 16);       // outside the image (or past the last pixel): the zero page
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + off),
                                             (__attribute__((address_space(3))) void*)(st + (wave * RA + q * 16) * 64), 16, 0, 0);
        }
        const char* wb = reinterpret_cast<const char*>(Wt + ((int64_t)(n0 >> 4) * g.nh + hs) * 512);
#pragma unroll
        for (int q = 0; q < QB; q++) {
            char* dst = st + HALF_A + (wave * RB + q * 16) * 64;
            if (q * 16 + 16 <= RB) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + vob[q]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else if (lane < 32) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + vob[q]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
        if (++is_cb == g.cpb) {
            is_cb = 0;
            if (++is_dx == KS) { is_dx = 0; ++is_dy; }
        }
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weight row -> MFMA row permutation of avd_vit.hip: a lane ends up with eight consecutive output channels
    auto b_row = [&](int j, int rho) __attribute__((always_inline)) {
        return wn * 64 + (j >> 1) * 32 + (rho >> 2) * 8 + (j & 1) * 4 + (rho & 3);
    };

    const int nh = g.nh;
    for (int hs = 0; hs < 3 && hs < nh; hs++) issue(hs);
    for (int hs = 0; hs < nh; hs++) {
        // half stage hs must have landed; younger LDS-DMAs of this wave: the half stages issued after it
        __builtin_amdgcn_sched_barrier(0);
        const int younger = (hs + 2 < nh - 1 ? hs + 2 : nh - 1) - hs;
        if (younger >= 2) AVD_WAIT_VMC(2 * P);
        else if (younger == 1) AVD_WAIT_VMC(P);
        else AVD_WAIT_VMC(0);
        __builtin_amdgcn_s_barrier();                        // ... for every wave, and everyone is done with half stage hs - 1
        __builtin_amdgcn_sched_barrier(0);
        if (hs + 3 < nh) issue(hs + 3);
        const char* cur = lds + (hs & 3) * STAGE;
        const int chunk = lane >> 4, r16 = lane & 15;
        bf16x8 a[TI], b[TJ];
#pragma unroll
        for (int j = 0; j < TJ; j++) b[j] = frag(cur + HALF_A, b_row(j, r16), chunk);
#pragma unroll
        for (int i = 0; i < TI; i++) a[i] = frag(cur, (wm * TI + i) * 16 + r16, chunk);
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue: bias (+ residual), ReLU, bf16, 16 bytes per lane into the blocked layout of the next operand
    const unsigned loff = (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ swz((lane >> 2) & 3)) << 4));
    const int cblocks = g.cout >> 5;
#pragma unroll
    for (int i = 0; i < TI; i++) {
        const int rblk = (m0 >> 4) + wm * TI + i;
#pragma unroll
        for (int jp = 0; jp < TJ / 2; jp++) {
            const int col0 = n0 + wn * 64 + jp * 32;
            const float* pb = bias + col0 + (lane >> 4) * 8;
            f32x4 lo = acc[i][2 * jp] + *reinterpret_cast<const f32x4*>(pb);
            f32x4 hi = acc[i][2 * jp + 1] + *reinterpret_cast<const f32x4*>(pb + 4);
            const int64_t boff = (int64_t)kZeroPage * 2 + ((int64_t)rblk * cblocks + (col0 >> 5)) * 1024;
            if (R) {
                const uint4 rv = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(R) + boff + loff);
                lo[0] += bf16_to_f32(rv.x & 0xFFFF); lo[1] += bf16_to_f32(rv.x >> 16);
                lo[2] += bf16_to_f32(rv.y & 0xFFFF); lo[3] += bf16_to_f32(rv.y >> 16);
                hi[0] += bf16_to_f32(rv.z & 0xFFFF); hi[1] += bf16_to_f32(rv.z >> 16);
                hi[2] += bf16_to_f32(rv.w & 0xFFFF); hi[3] += bf16_to_f32(rv.w >> 16);
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; e++) { lo[e] = fmaxf(lo[e], 0.f); hi[e] = fmaxf(hi[e], 0.f); }
            }
            uint4 pk;
            pk.x = pack_bf16x2(lo[0], lo[1]);
            pk.y = pack_bf16x2(lo[2], lo[3]);
            pk.z = pack_bf16x2(hi[0], hi[1]);
            pk.w = pack_bf16x2(hi[2], hi[3]);
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Y) + boff + loff) = pk;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// A bottleneck block's 3x3 convolution AND its expanding 1x1 in one kernel ("conv2 + conv3"): the 3x3's output tile
// (BM pixels x all `mid` channels) never leaves the CU.  The expanding layers are the slowest of the unfused network
// (K = mid is short: they move a residual and an output four times the size of their input and multiply little:
// 140-260 TFLOP/s); fused, their operand comes from LDS, their residual reads and output stores overlap the NEXT
// workgroup's 3x3 on the same CU, and the mid activation's write + read (2 x 48 MB per block at 56 x 56) disappears.
//   1. the 3x3 exactly as k_conv_bf16 (same ring, same K order: same bits), BN = mid, one column tile;
//   2. its epilogue (bias, ReLU, bf16) goes to LDS in the operand layout (KS2 half tiles of BM rows x 64 B) and every
//      wave takes the fragments of ITS pixel rows into registers (KS2 x TI fragments) -- the ring is free again;
//   3. the expanding 1x1 in NCH = 4 chunks of BN output channels: W3's chunk (BN x mid, one contiguous piece of the
//      blocked weight) is copied into one of two LDS buffers by LDS-DMA while the previous chunk multiplies; epilogue
//      = k_conv_bf16's (bias + residual + ReLU -> 16-byte stores into the blocked layout), the residual loads of a chunk
//      are issued before its MFMAs.  Same k order as the unfused layer: the result is bit-identical to conv2 -> conv3.
template <int BM, int WAVES_M, int TI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_conv3_expand(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W2,
                                                     const float* __restrict__ bias2, const uint16_t* __restrict__ W3,
                                                     const float* __restrict__ bias3, const uint16_t* __restrict__ R,
                                                     uint16_t* __restrict__ Y, ConvGeom g)
{
    constexpr int KS = 3, WAVES_N = 8 / WAVES_M, TJ = 4, BN = WAVES_N * 64;
    static_assert(WAVES_M * TI * 16 == BM, "tile rows");
    constexpr int RA = BM / 8, QA = RA / 16, RB = BN / 8, QB = (RB + 15) / 16, P = QA + QB;
    constexpr int HALF_A = BM * 64, HALF_B = BN * 64, STAGE = HALF_A + HALF_B, RING = 4 * STAGE;
    constexpr int KS2 = BN / 32;                            // k steps of the expanding layer (K = mid = BN)
    constexpr int A2 = BM * BN * 2, B2 = BN * BN * 2;       // bytes: the mid tile; one chunk of W3
    constexpr int NBLK = (BN / 16) * KS2;                   // KiB blocks of one W3 chunk
    static_assert(A2 + B2 <= RING && 2 * B2 <= RING && NBLK % 8 == 0, "the expand's buffers live in the 3x3's ring");
    constexpr int NCH = 4;
    extern __shared__ __align__(16) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int total = (g.m_out + BM - 1) / BM;
    const int per = (gridDim.x + 7) >> 3;
    const int lid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (lid >= total) return;
    const int m0 = lid * BM;
    // The expanding layer's bias is fetched NOW, before the first LDS-DMA enters the in-order vector-memory queue (one element per
    // thread, parked in LDS after the 3x3): a load issued inside the chunk loop would make the compiler wait for everything in
    // flight around it.
    static_assert(NCH * BN <= 512, "one bias3 element per thread");
    float b3v = threadIdx.x < NCH * BN ? bias3[threadIdx.x] : 0.f;

    int rin0[QA], cw[QA];
    unsigned tapmask[QA];
    const int hw = g.hout * g.wout;
#pragma unroll
    for (int q = 0; q < QA; q++) {
        const int r = wave * RA + q * 16 + (lane >> 2), m = m0 + r;
        const bool valid = m < g.m_out;
        const int img = m / hw, rem = m - img * hw, oy = rem / g.wout, ox = rem - oy * g.wout;
        cw[q] = (lane & 3) ^ swz((r >> 2) & 3);
        const int y0 = oy * g.stride - g.pad, x0 = ox * g.stride - g.pad;
        rin0[q] = (img * g.hin + y0) * g.win + x0;
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < KS * KS; t++) {
            const int yi = y0 + t / KS, xi = x0 + t % KS;
            if (valid && (unsigned)yi < (unsigned)g.hin && (unsigned)xi < (unsigned)g.win) mk |= 1u << t;
        }
        tapmask[q] = mk;
    }
    unsigned vob[QB];
#pragma unroll
    for (int q = 0; q < QB; q++) {
        const int r = wave * RB + q * 16 + (lane >> 2);
        vob[q] = (unsigned)((r >> 4) * g.nh * 1024 + (r & 15) * 64 + (lane & 3) * 16);
    }
    const char* xb = reinterpret_cast<const char*>(X);
    int is_cb = 0, is_dy = 0, is_dx = 0;
    auto issue = [&](int hs) __attribute__((always_inline)) {
        char* st = lds + (hs & 3) * STAGE;
#pragma unroll
        for (int q = 0; q < QA; q++) {
            const int rin = rin0[q] + is_dy * g.win + is_dx;
            const bool ok = (tapmask[q] >> (is_dy * KS + is_dx)) & 1u;
            unsigned off = (unsigned)(kZeroPage * 2) + (unsigned)((rin >> 4) * g.cpb + is_cb) * 1024u + (unsigned)((rin & 15) * 64) +
                           (unsigned)((cw[q] ^ swz((rin >> 2) & 3)) << 4);
            off = ok ? off : (unsigned)(lane * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + off),
                                             (__attribute__((address_space(3))) void*)(st + (wave * RA + q * 16) * 64), 16, 0, 0);
        }
        const char* wb = reinterpret_cast<const char*>(W2 + (int64_t)hs * 512);
#pragma unroll
        for (int q = 0; q < QB; q++) {
            char* dst = st + HALF_A + (wave * RB + q * 16) * 64;
            if (q * 16 + 16 <= RB) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + vob[q]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else if (lane < 32) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + vob[q]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
        if (++is_cb == g.cpb) {
            is_cb = 0;
            if (++is_dx == KS) { is_dx = 0; ++is_dy; }
        }
    };

    f32x4 acc[TI][TJ];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    auto b_row = [&](int j, int rho) __attribute__((always_inline)) {
        return wn * 64 + (j >> 1) * 32 + (rho >> 2) * 8 + (j & 1) * 4 + (rho & 3);
    };
    const int chunk = lane >> 4, r16 = lane & 15;

    asm volatile("" : "+v"(b3v));                            // it has arrived (as far as the compiler is concerned too) before the queue fills

    // LDS after the 3x3.  W3 is RESIDENT when all of it fits beside the mid tile (mid = 64: 32 KiB in one go, no barrier in the
    // chunk loop); otherwise its chunks alternate between two buffers, the second of which takes over the mid tile's LDS once
    // the fragments are in registers.  W3 (its first chunk) and the first residuals are requested during the 3x3's LAST step,
    // into ring slots that step no longer reads: the last step reads slot (nh - 1) & 3 only (checked by the launcher).
    //   resident (nh & 3 == 2, slot 1 live):  mid tile [0, A2) | W3 [2 STAGE, + NCH B2) | bias3 behind it
    //   otherwise (nh & 3 == 0, slot 3 live): W3 chunk c at (c & 1) B2 | mid tile [B2, B2 + A2) | bias3 behind the ring
    constexpr bool RESIDENT = NCH * B2 + NCH * BN * 4 <= 2 * STAGE && A2 <= 2 * STAGE;
    static_assert(RESIDENT ? (2 * STAGE + NCH * B2 + NCH * BN * 4 <= RING) : (B2 + A2 <= RING && 2 * B2 <= RING && B2 <= 2 * STAGE), "LDS plan");
    constexpr int W3BASE = RESIDENT ? 2 * STAGE : 0, A2BASE = RESIDENT ? 0 : B2;
    constexpr int NRES = TI * TJ / 2;                       // residual loads (= output stores) per wave and chunk
    char* const a2 = lds + A2BASE;
    float* const lbias3 = reinterpret_cast<float*>(lds + (RESIDENT ? W3BASE + NCH * B2 : RING));
    auto issue_w3 = [&](int c) __attribute__((always_inline)) {
        char* buf = lds + W3BASE + (RESIDENT ? c : (c & 1)) * B2;
        const char* src = reinterpret_cast<const char*>(W3) + (size_t)c * NBLK * 1024 + lane * 16;
#pragma unroll
        for (int t0 = 0; t0 < NBLK; t0 += 8) {
            const int t = t0 + wave, rbl = t / KS2, kb = t % KS2;     // block t of the chunk: 16 weight rows x 32 k
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + t * 1024),
                                             (__attribute__((address_space(3))) void*)(buf + kb * (BN * 64) + rbl * 1024), 16, 0, 0);
        }
    };
    const unsigned loff = (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ swz((lane >> 2) & 3)) << 4));
    const int cblocks = (NCH * BN) >> 5;
    uint4 rv[2][TI][TJ / 2];                                 // residuals: the chunk being finished and the next one (in flight)
    // byte offset of this lane's 16 bytes of output piece (chunk c, row tile i, column pair jp): 32-bit (an activation is < 4 GiB)
    const unsigned obase = (unsigned)(kZeroPage * 2) + (unsigned)(((m0 >> 4) + wm * TI) * cblocks + wn * 2) * 1024u + loff;
    auto out_off = [&](int c, int i, int jp) __attribute__((always_inline)) {
        return obase + (unsigned)((i * cblocks + c * (BN >> 5) + jp) * 1024);
    };
    // PF: a chunk's residuals are requested one chunk ahead (two register sets); without it (the 128-channel shape, whose
    // fragments of the mid tile take 32 registers) at the top of their own chunk, in front of its MFMAs
    constexpr bool PF = RESIDENT;
    auto load_res = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int jp = 0; jp < TJ / 2; jp++)
                rv[PF ? (c & 1) : 0][i][jp] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(R) + out_off(c, i, jp));
    };

    // ---- 1. the 3x3 (the loop of k_conv_bf16)
    const int nh = g.nh;
    auto step = [&](int hs, auto last_c) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        const int younger = (hs + 2 < nh - 1 ? hs + 2 : nh - 1) - hs;
        if (LAST) AVD_WAIT_VMC(0);
        else if (younger >= 2) AVD_WAIT_VMC(2 * P);
        else if (younger == 1) AVD_WAIT_VMC(P);
        else AVD_WAIT_VMC(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (LAST) {                                          // everyone is past step nh - 2: three ring slots are free
            if (RESIDENT) {
#pragma unroll
                for (int c = 0; c < NCH; c++) issue_w3(c);
            } else {
                issue_w3(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PF) load_res(0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (hs + 3 < nh) {
            issue(hs + 3);
        }
        const char* cur = lds + (hs & 3) * STAGE;
        bf16x8 a[TI], b[TJ];
#pragma unroll
        for (int j = 0; j < TJ; j++) b[j] = frag(cur + HALF_A, b_row(j, r16), chunk);
#pragma unroll
        for (int i = 0; i < TI; i++) a[i] = frag(cur, (wm * TI + i) * 16 + r16, chunk);
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    };
    for (int hs = 0; hs < 3; hs++) issue(hs);                // nh >= 18
    for (int hs = 0; hs < nh - 1; hs++) step(hs, std::false_type{});
    step(nh - 1, std::true_type{});

    // ---- 2. the mid tile -> LDS (operand layout) -> this wave's fragments
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // every wave has read its last fragments: the whole ring is free
    __builtin_amdgcn_sched_barrier(0);
    // the 3x3's bias: this lane's 16 values.  (The compiler puts a vmcnt(0) in front of the LDS stores below anyway, because
    // LDS-DMAs are in flight -- W3 and the first chunk of residuals have had the last step to arrive -- so this load costs no
    // extra wait.)
    f32x4 b2lo[TJ / 2], b2hi[TJ / 2];
#pragma unroll
    for (int jp = 0; jp < TJ / 2; jp++) {
        const float* pb = bias2 + wn * 64 + jp * 32 + (lane >> 4) * 8;
        b2lo[jp] = *reinterpret_cast<const f32x4*>(pb);
        b2hi[jp] = *reinterpret_cast<const f32x4*>(pb + 4);
    }
    if (threadIdx.x < NCH * BN) lbias3[threadIdx.x] = b3v;
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int jp = 0; jp < TJ / 2; jp++) {
            f32x4 lo = acc[i][2 * jp] + b2lo[jp];
            f32x4 hi = acc[i][2 * jp + 1] + b2hi[jp];
#pragma unroll
            for (int e = 0; e < 4; e++) { lo[e] = fmaxf(lo[e], 0.f); hi[e] = fmaxf(hi[e], 0.f); }
            uint4 pk;
            pk.x = pack_bf16x2(lo[0], lo[1]);
            pk.y = pack_bf16x2(lo[2], lo[3]);
            pk.z = pack_bf16x2(hi[0], hi[1]);
            pk.w = pack_bf16x2(hi[2], hi[3]);
            *reinterpret_cast<uint4*>(a2 + (wn * 2 + jp) * (BM * 64) + (wm * TI + i) * 1024 + loff) = pk;
        }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's part of W3 has landed, its part of the mid tile is written
    __builtin_amdgcn_s_barrier();                            // ... and everybody else's
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 am[KS2][TI];
#pragma unroll
    for (int ks = 0; ks < KS2; ks++)
#pragma unroll
        for (int i = 0; i < TI; i++) am[ks][i] = frag(a2 + ks * (BM * 64), (wm * TI + i) * 16 + r16, chunk);
    if (!RESIDENT) {                                         // the second W3 buffer takes over the mid tile's LDS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- 3. the expanding 1x1, BN output channels at a time.  Vector-memory queue of a wave (in order), resident W3: res(1) st(0) |
    // res(2) st(1) | res(3) st(2) | st(3); otherwise res(c) DMA(c + 1) st(c) per chunk: the wait for DMA(c + 1) at the end of
    // iteration c leaves the stores of chunk c in flight.
    static_assert(NCH % 2 == 0, "the chunk loop is unrolled by two (the residual buffers alternate)");
#pragma unroll 1
    for (int c2 = 0; c2 < NCH; c2 += 2)
#pragma unroll
    for (int cc = 0; cc < 2; cc++) {
        const int c = c2 + cc;
        if (!PF) load_res(c);
        if (!RESIDENT && c + 1 < NCH) issue_w3(c + 1);
        if (PF && c + 1 < NCH) load_res(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        zero_acc();
        const char* buf = lds + W3BASE + (RESIDENT ? c : (c & 1)) * B2;
#pragma unroll
        for (int ks = 0; ks < KS2; ks++) {
            bf16x8 b[TJ];
#pragma unroll
            for (int j = 0; j < TJ; j++) b[j] = frag(buf + ks * (BN * 64), b_row(j, r16), chunk);
#pragma unroll
            for (int i = 0; i < TI; i++)
#pragma unroll
                for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], am[ks][i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int jp = 0; jp < TJ / 2; jp++) {
                const float* pb = lbias3 + c * BN + wn * 64 + jp * 32 + (lane >> 4) * 8;
                f32x4 lo = acc[i][2 * jp] + *reinterpret_cast<const f32x4*>(pb);
                f32x4 hi = acc[i][2 * jp + 1] + *reinterpret_cast<const f32x4*>(pb + 4);
                const uint4 r = rv[PF ? (c & 1) : 0][i][jp];
                lo[0] += bf16_to_f32(r.x & 0xFFFF); lo[1] += bf16_to_f32(r.x >> 16);
                lo[2] += bf16_to_f32(r.y & 0xFFFF); lo[3] += bf16_to_f32(r.y >> 16);
                hi[0] += bf16_to_f32(r.z & 0xFFFF); hi[1] += bf16_to_f32(r.z >> 16);
                hi[2] += bf16_to_f32(r.w & 0xFFFF); hi[3] += bf16_to_f32(r.w >> 16);
#pragma unroll
                for (int e = 0; e < 4; e++) { lo[e] = fmaxf(lo[e], 0.f); hi[e] = fmaxf(hi[e], 0.f); }
                uint4 pk;
                pk.x = pack_bf16x2(lo[0], lo[1]);
                pk.y = pack_bf16x2(lo[2], lo[3]);
                pk.z = pack_bf16x2(hi[0], hi[1]);
                pk.w = pack_bf16x2(hi[2], hi[3]);
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Y) + out_off(c, i, jp)) = pk;
            }
        __builtin_amdgcn_sched_barrier(0);
        if (!RESIDENT && c + 1 < NCH) {
            AVD_WAIT_VMC((PF ? 2 : 1) * NRES);               // younger than DMA(c + 1): [res(c + 1),] st(c)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // the next chunk has landed for every wave; everyone is done reading this one
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same fused block for mid = 64, stride 1 (the 56 x 56 stage) with the 3x3's input as ONE slab in LDS.  k_conv3_expand
// gathers the 256 output pixels' inputs from L2 once per tap: 9 x 32 KB of its 520 KB of ingest per workgroup, on a path that
// delivers ~37-40 GB/s per CU whatever the instruction (profiles/r04_experiments.md section 5).  Output pixels m0 .. m0 + 255 of
// the linear pixel index see, through the nine taps, input pixels m0 - 57 .. m0 + 312: the 384 pixels m0 - 64 .. m0 + 319 are
// 24 pixel blocks x 2 channel blocks = 48 CONTIGUOUS KiB of the blocked activation, copied once (48 LDS-DMA instructions).  A tap
// is then a fragment read at a shifted slab row -- lane (pixel r, chunk) reads row 64 + tile row + r + dy W + dx, which carries its
// own swizzle key -- with the fragment zeroed where the tap leaves the image (a per-lane 9-bit mask).  Only the weights stream:
// one tap (64 x 64 bf16 = 8 KB, one LDS-DMA instruction per wave) per step, ring of four, 16 MFMAs per wave between barriers.
// Same K order (tap, then channel block) as the gathering kernels: bit-identical.  The expanding layer follows as in
// k_conv3_expand (resident W3; it is requested after the last tap, when the slab is dead).
// MID = 64 (56 x 56: 256-pixel tiles, a whole tap of W2 per step, W3 resident) or 128 (28 x 28: 128-pixel tiles, slab = 192 pixels x 4 channel
// blocks -- again 48 contiguous KiB --, W2 streams in half stages of 32 input channels = 8 KiB like the gathering kernel, W3 in two alternating
// chunk buffers).
template <int MID>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_slab3_expand(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W2,
                                                     const float* __restrict__ bias2, const uint16_t* __restrict__ W3,
                                                     const float* __restrict__ bias3, const uint16_t* __restrict__ R,
                                                     uint16_t* __restrict__ Y, ConvGeom g, int in_blocks)
{
    static_assert(MID == 64 || MID == 128, "the two early stages");
    constexpr int BN = MID, WAVES_N = BN / 64, WAVES_M = 8 / WAVES_N, TI = 2, TJ = 4, BM = WAVES_M * TI * 16;   // 256 x 64 or 128 x 128
    constexpr int CPB = MID / 32, KS2 = CPB, NCH = 4;       // channel blocks of the input = k steps of the expanding layer
    constexpr int HALO = MID == 64 ? 64 : 32;               // pixels in front of m0 (>= W + 1, whole pixel blocks)
    constexpr int NSLAB = BM + 2 * HALO, SLAB = NSLAB * CPB * 64;                 // 384 x 128 B = 192 x 256 B = 48 KiB
    constexpr int KPS = MID == 64 ? 2 : 1;                  // k blocks of W2 per step (a whole tap at 64 channels, a quarter tap at 128)
    constexpr int STEPB = BN * 64 * KPS, NSTEP = 9 * CPB / KPS, LDS = SLAB + 4 * STEPB;   // 8 KiB per step, ring of four: 80 KiB
    constexpr int A2 = BM * BN * 2, B2 = BN * BN * 2;
    constexpr bool RESIDENT = NCH * B2 + A2 + NCH * BN * 4 <= LDS;               // W3 | mid tile | bias3 (mid 64); else two chunk buffers
    constexpr int A2BASE = RESIDENT ? NCH * B2 : B2, BIAS3 = RESIDENT ? A2BASE + A2 : 2 * B2;
    static_assert(SLAB == 49152 && STEPB == 8192 && BIAS3 + NCH * BN * 4 <= LDS && A2BASE + A2 <= LDS, "LDS plan");
    constexpr int NBLK = (BN / 16) * KS2;                   // KiB blocks of one W3 chunk
    constexpr int NRES = TI * TJ / 2;
    extern __shared__ __align__(16) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int total = (g.m_out + BM - 1) / BM;
    const int per = (gridDim.x + 7) >> 3;
    const int lid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (lid >= total) return;
    const int m0 = lid * BM;
    static_assert(NCH * BN <= 512, "one bias3 element per thread");
    float b3v = threadIdx.x < NCH * BN ? bias3[threadIdx.x] : 0.f;
    asm volatile("" : "+v"(b3v));                            // arrived before the queue fills (k_conv3_expand)
    const int chunk = lane >> 4, r16 = lane & 15;
    const char* xb = reinterpret_cast<const char*>(X);

    // ---- the slab: NSLAB / 16 pixel blocks from (m0 - HALO) >> 4 on, CPB channel blocks each = 48 KiB blocks, six per wave; blocks outside
    // the activation come from the zero page (their pixels are masked in every tap: they only must not fault)
    const int blk0 = (m0 - HALO) >> 4;
#pragma unroll
    for (int j0 = 0; j0 < 48; j0 += 8) {
        const int j = j0 + wave, pb = blk0 + j / CPB;
        const bool ok = pb >= 0 && pb < in_blocks;          // wave-uniform
        const char* src = ok ? xb + kZeroPage * 2 + ((size_t)pb * CPB + j % CPB) * 1024 + lane * 16 : xb + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
    }
    // one step of W2 = KPS k blocks of all BN weight rows: eight KiB blocks, one per wave (k block of step s: s * KPS ..; K = 9 MID)
    auto issue_w2 = [&](int st) __attribute__((always_inline)) {
        const int rb = MID == 64 ? wave >> 1 : wave, kbl = MID == 64 ? wave & 1 : 0;
        const char* src = reinterpret_cast<const char*>(W2) + ((size_t)(rb * (9 * CPB) + st * KPS + kbl)) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + SLAB + (st & 3) * STEPB + kbl * (BN * 64) + rb * 1024), 16, 0, 0);
    };
    issue_w2(0); issue_w2(1); issue_w2(2);

    // per-lane tap masks of this wave's two row tiles, and the slab row of shift 0 of each
    unsigned tapmask[TI];
    int qbase[TI];
    const int hw = g.hout * g.wout;
#pragma unroll
    for (int i = 0; i < TI; i++) {
        const int pl = (wm * TI + i) * 16 + r16, m = m0 + pl;
        const bool valid = m < g.m_out;
        const int img = m / hw, rem = m - img * hw, oy = rem / g.wout, ox = rem - oy * g.wout;
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int yi = oy - 1 + t / 3, xi = ox - 1 + t % 3;
            if (valid && (unsigned)yi < (unsigned)g.hin && (unsigned)xi < (unsigned)g.win) mk |= 1u << t;
        }
        tapmask[i] = mk;
        qbase[i] = HALO + pl;
    }
    f32x4 acc[TI][TJ];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    auto b_row = [&](int j, int rho) __attribute__((always_inline)) { return wn * 64 + (j >> 1) * 32 + (rho >> 2) * 8 + (j & 1) * 4 + (rho & 3); };
    const unsigned loff = (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ swz((lane >> 2) & 3)) << 4));
    constexpr int cblocks = (NCH * BN) >> 5;
    constexpr bool PF = RESIDENT;                            // residuals one chunk ahead (two register sets) only where the registers allow
    uint4 rv[PF ? 2 : 1][TI][TJ / 2];
    const unsigned obase = (unsigned)(kZeroPage * 2) + (unsigned)(((m0 >> 4) + wm * TI) * cblocks + wn * 2) * 1024u + loff;
    auto out_off = [&](int c, int i, int jp) __attribute__((always_inline)) { return obase + (unsigned)((i * cblocks + c * (BN >> 5) + jp) * 1024); };
    auto load_res = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int jp = 0; jp < TJ / 2; jp++)
                rv[PF ? (c & 1) : 0][i][jp] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(R) + out_off(c, i, jp));
    };

    // ---- 1. the nine taps (K order: tap, then channel block -- the gathering kernels' order)
#pragma unroll
    for (int st = 0; st < NSTEP; st++) {
        __builtin_amdgcn_sched_barrier(0);
        if (st < NSTEP - 2) AVD_WAIT_VMC(2);                 // steps st + 1, st + 2 may still be in flight (the slab is older than step 0)
        else if (st == NSTEP - 2) AVD_WAIT_VMC(1);
        else AVD_WAIT_VMC(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (st + 3 < NSTEP) issue_w2(st + 3);
        else if (PF && st == NSTEP - 1) load_res(0);         // the first residuals, a whole step + the mid tile's epilogue ahead
        __builtin_amdgcn_sched_barrier(0);
        const int t = st * KPS / CPB, cb0 = st * KPS % CPB;  // tap and first channel block of the step
        const int shift = (t / 3 - 1) * g.win + (t % 3 - 1);
        const char* sb = lds + SLAB + (st & 3) * STEPB;
        bf16x8 a[TI][KPS], b[TJ][KPS];
#pragma unroll
        for (int kb = 0; kb < KPS; kb++) {
#pragma unroll
            for (int j = 0; j < TJ; j++) b[j][kb] = frag(sb + kb * (BN * 64), b_row(j, r16), chunk);
#pragma unroll
            for (int i = 0; i < TI; i++) {
                const int q = qbase[i] + shift;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(lds + (q >> 4) * (CPB * 1024) + (cb0 + kb) * 1024 + (q & 15) * 64 +
                                                                  ((chunk ^ swz(((q & 15) >> 2) & 3)) << 4));
                a[i][kb] = ((tapmask[i] >> t) & 1u) ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int kb = 0; kb < KPS; kb++)
#pragma unroll
            for (int i = 0; i < TI; i++)
#pragma unroll
                for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][kb], a[i][kb], acc[i][j], 0, 0, 0);
    }

    // ---- 2. W3 (its first chunk) into the dead slab, the mid tile -> LDS -> this wave's fragments
    char* const a2 = lds + A2BASE;
    float* const lbias3 = reinterpret_cast<float*>(lds + BIAS3);
    auto issue_w3 = [&](int c) __attribute__((always_inline)) {
        char* buf = lds + (RESIDENT ? c : (c & 1)) * B2;
        const char* src = reinterpret_cast<const char*>(W3) + (size_t)c * NBLK * 1024 + lane * 16;
#pragma unroll
        for (int t0 = 0; t0 < NBLK; t0 += 8) {
            const int t = t0 + wave, rbl = t / KS2, kb = t % KS2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + t * 1024),
                                             (__attribute__((address_space(3))) void*)(buf + kb * (BN * 64) + rbl * 1024), 16, 0, 0);
        }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // every wave has read its last fragments
    __builtin_amdgcn_sched_barrier(0);
    f32x4 b2lo[TJ / 2], b2hi[TJ / 2];
#pragma unroll
    for (int jp = 0; jp < TJ / 2; jp++) {
        const float* pb = bias2 + wn * 64 + jp * 32 + (lane >> 4) * 8;
        b2lo[jp] = *reinterpret_cast<const f32x4*>(pb);
        b2hi[jp] = *reinterpret_cast<const f32x4*>(pb + 4);
    }
    if (RESIDENT) {
#pragma unroll
        for (int c = 0; c < NCH; c++) issue_w3(c);
    } else {
        issue_w3(0);
    }
    if (threadIdx.x < NCH * BN) lbias3[threadIdx.x] = b3v;
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int jp = 0; jp < TJ / 2; jp++) {
            f32x4 lo = acc[i][2 * jp] + b2lo[jp];
            f32x4 hi = acc[i][2 * jp + 1] + b2hi[jp];
#pragma unroll
            for (int e = 0; e < 4; e++) { lo[e] = fmaxf(lo[e], 0.f); hi[e] = fmaxf(hi[e], 0.f); }
            uint4 pk;
            pk.x = pack_bf16x2(lo[0], lo[1]); pk.y = pack_bf16x2(lo[2], lo[3]);
            pk.z = pack_bf16x2(hi[0], hi[1]); pk.w = pack_bf16x2(hi[2], hi[3]);
            *reinterpret_cast<uint4*>(a2 + (wn * 2 + jp) * (BM * 64) + (wm * TI + i) * 1024 + loff) = pk;
        }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // W3 (its first chunk) and the mid tile are complete for everyone
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 am[KS2][TI];
#pragma unroll
    for (int ks = 0; ks < KS2; ks++)
#pragma unroll
        for (int i = 0; i < TI; i++) am[ks][i] = frag(a2 + ks * (BM * 64), (wm * TI + i) * 16 + r16, chunk);
    if (!RESIDENT) {                                         // the second W3 buffer takes over the mid tile's LDS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- 3. the expanding 1x1 (k_conv3_expand's chunk loop)
    static_assert(NCH % 2 == 0, "unrolled by two");
#pragma unroll 1
    for (int c2 = 0; c2 < NCH; c2 += 2)
#pragma unroll
    for (int cc = 0; cc < 2; cc++) {
        const int c = c2 + cc;
        if (!PF) load_res(c);
        if (!RESIDENT && c + 1 < NCH) issue_w3(c + 1);
        if (PF && c + 1 < NCH) load_res(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        zero_acc();
        const char* buf = lds + (RESIDENT ? c : (c & 1)) * B2;
#pragma unroll
        for (int ks = 0; ks < KS2; ks++) {
            bf16x8 b[TJ];
#pragma unroll
            for (int j = 0; j < TJ; j++) b[j] = frag(buf + ks * (BN * 64), b_row(j, r16), chunk);
#pragma unroll
            for (int i = 0; i < TI; i++)
#pragma unroll
                for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], am[ks][i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int jp = 0; jp < TJ / 2; jp++) {
                const float* pb = lbias3 + c * BN + wn * 64 + jp * 32 + (lane >> 4) * 8;
                f32x4 lo = acc[i][2 * jp] + *reinterpret_cast<const f32x4*>(pb);
                f32x4 hi = acc[i][2 * jp + 1] + *reinterpret_cast<const f32x4*>(pb + 4);
                const uint4 r = rv[PF ? (c & 1) : 0][i][jp];
                lo[0] += bf16_to_f32(r.x & 0xFFFF); lo[1] += bf16_to_f32(r.x >> 16);
                lo[2] += bf16_to_f32(r.y & 0xFFFF); lo[3] += bf16_to_f32(r.y >> 16);
                hi[0] += bf16_to_f32(r.z & 0xFFFF); hi[1] += bf16_to_f32(r.z >> 16);
                hi[2] += bf16_to_f32(r.w & 0xFFFF); hi[3] += bf16_to_f32(r.w >> 16);
#pragma unroll
                for (int e = 0; e < 4; e++) { lo[e] = fmaxf(lo[e], 0.f); hi[e] = fmaxf(hi[e], 0.f); }
                uint4 pk;
                pk.x = pack_bf16x2(lo[0], lo[1]); pk.y = pack_bf16x2(lo[2], lo[3]);
                pk.z = pack_bf16x2(hi[0], hi[1]); pk.w = pack_bf16x2(hi[2], hi[3]);
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Y) + out_off(c, i, jp)) = pk;
            }
        __builtin_amdgcn_sched_barrier(0);
        if (!RESIDENT && c + 1 < NCH) {
            AVD_WAIT_VMC(NRES);                              // younger than DMA(c + 1): the stores of chunk c
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // the next chunk has landed for every wave; everyone is done reading this one
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// BGR uint8 frame -> 224 x 224 (float bilinear taps, cv2's INTER_LINEAR centre mapping), RGB, (x / 255 - mean) / std,
// bf16: img[frame][y + 3][x + 3][4] of the zero-bordered 232 x 232 image (fourth channel zero; the border is cleared once,
// when the buffer is allocated).  Same arithmetic as k_vit_patchify.
__global__ __launch_bounds__(256) void k_cnn_input(const uint8_t* __restrict__ bgr, int n, int h, int w, int64_t row_stride,
                                                  int64_t frame_stride, uint16_t* __restrict__ img)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n * kSide * kSide) return;
    const int x = gid % kSide, y = (gid / kSide) % kSide, f = gid / (kSide * kSide);
    const float sx = (float)w / kSide, sy = (float)h / kSide;
    float fx = (x + 0.5f) * sx - 0.5f, fy = (y + 0.5f) * sy - 0.5f;
    int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
    fx -= x0; fy -= y0;
    if (x0 < 0) { x0 = 0; fx = 0.f; }
    if (x0 >= w - 1) { x0 = w - 1; fx = 0.f; }
    if (y0 < 0) { y0 = 0; fy = 0.f; }
    if (y0 >= h - 1) { y0 = h - 1; fy = 0.f; }
    const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    const uint8_t* fr = bgr + (int64_t)f * frame_stride;
    const uint8_t *p00 = fr + (int64_t)y0 * row_stride + x0 * 3, *p01 = fr + (int64_t)y0 * row_stride + x1 * 3;
    const uint8_t *p10 = fr + (int64_t)y1 * row_stride + x0 * 3, *p11 = fr + (int64_t)y1 * row_stride + x1 * 3;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, istd[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};   // RGB (ImageNet)
    uint16_t v4[4] = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int s = 2 - c;
        const float top = p00[s] + (p01[s] - (float)p00[s]) * fx, bot = p10[s] + (p11[s] - (float)p10[s]) * fx;
        const float v = top + (bot - top) * fy;
        v4[c] = f32_to_bf16((v * (1.f / 255.f) - mean[c]) * istd[c]);
    }
    uint2 pk;
    pk.x = v4[0] | ((unsigned)v4[1] << 16);
    pk.y = v4[2];
    reinterpret_cast<uint2*>(img)[((int64_t)f * kImgSide + y + 3) * kImgSide + x + 3] = pk;
}

__device__ __forceinline__ unsigned max_bf16x2(unsigned a, unsigned b)     // inputs are >= 0 (after a ReLU): integer order = value order
{
    const unsigned lo = max(a & 0xFFFFu, b & 0xFFFFu), hi = max(a >> 16, b >> 16);
    return lo | (hi << 16);
}

// 3x3 / 2 max pool (pad 1) over a blocked [n * hin * win][c] activation of non-negative values
__global__ __launch_bounds__(256) void k_maxpool3(const uint16_t* __restrict__ X, int n, int hin, int win, int c, uint16_t* __restrict__ Y)
{
    const int hout = hin / 2, wout = win / 2, c8n = c / 8;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)n * hout * wout * c8n) return;
    const int c8 = (int)(gid % c8n);
    const int m = (int)(gid / c8n);
    const int f = m / (hout * wout), rem = m - f * hout * wout, oy = rem / wout, ox = rem - oy * wout;
    uint4 best = {0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const int yi = oy * 2 - 1 + dy, xi = ox * 2 - 1 + dx;
            if ((unsigned)yi < (unsigned)hin && (unsigned)xi < (unsigned)win) {
                const uint4 v = *reinterpret_cast<const uint4*>(X + kZeroPage + blocked_index((f * hin + yi) * win + xi, c8 * 8, c));
                best.x = max_bf16x2(best.x, v.x); best.y = max_bf16x2(best.y, v.y);
                best.z = max_bf16x2(best.z, v.z); best.w = max_bf16x2(best.w, v.w);
            }
        }
    *reinterpret_cast<uint4*>(Y + kZeroPage + blocked_index(m, c8 * 8, c)) = best;
}

// global average pool: blocked [n * hw][c] -> f32 [n][c] (sum in f32 in pixel order, then / hw)
__global__ __launch_bounds__(256) void k_avgpool(const uint16_t* __restrict__ X, int n, int hw, int c, float* __restrict__ out)
{
    const int c8n = c / 8;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n * c8n) return;
    const int c8 = gid % c8n, f = gid / c8n;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < hw; p++) {
        const uint4 v = *reinterpret_cast<const uint4*>(X + kZeroPage + blocked_index(f * hw + p, c8 * 8, c));
        s[0] += bf16_to_f32(v.x & 0xFFFF); s[1] += bf16_to_f32(v.x >> 16); s[2] += bf16_to_f32(v.y & 0xFFFF); s[3] += bf16_to_f32(v.y >> 16);
        s[4] += bf16_to_f32(v.z & 0xFFFF); s[5] += bf16_to_f32(v.z >> 16); s[6] += bf16_to_f32(v.w & 0xFFFF); s[7] += bf16_to_f32(v.w >> 16);
    }
    const float inv = 1.f / hw;
#pragma unroll
    for (int e = 0; e < 8; e++) out[(int64_t)f * c + c8 * 8 + e] = s[e] * inv;
}

// logits[f][o] = pooled[f][:] . W[o][:] + b[o]; one wave per (output, group of 8 frames): the weight row (k = 2048: 32
// values per lane) stays in registers, the pooled features come from L2
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ x, const uint16_t* __restrict__ W, const float* __restrict__ b,
                                               int n, int nout, float* __restrict__ y)
{
    constexpr int K = 2048, FG = 8;
    const int groups = (n + FG - 1) / FG;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wid >= groups * nout) return;
    const int o = wid % nout, f0 = (wid / nout) * FG;
    float wr[K / 64];
#pragma unroll
    for (int i = 0; i < K / 64; i++) wr[i] = bf16_to_f32(W[(int64_t)o * K + i * 64 + lane]);
    const float bo = b[o];
    for (int f = f0; f < f0 + FG && f < n; f++) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < K / 64; i++) s = __builtin_fmaf(x[(int64_t)f * K + i * 64 + lane], wr[i], s);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        if (lane == 0) y[(int64_t)f * nout + o] = s + bo;
    }
}

// ---- network description -----------------------------------------------------------------------------------------
struct Layer { int cin, cout, ksize, stride; size_t w_off, b_off; };        // offsets into the flat parameter arrays (elements)

struct Net {
    std::vector<Layer> convs;          // stem, then per block: conv1, conv2, conv3 [, downsample]
    size_t n_w = 0, n_b = 0;           // including the final linear layer
    size_t fc_w = 0, fc_b = 0;
    Net()
    {
        auto add = [&](int cin, int cout, int k, int s) {
            convs.push_back({cin, cout, k, s, n_w, n_b});
            n_w += (size_t)cout * k * k * cin;
            n_b += cout;
        };
        add(3, 64, 7, 2);
        const int depth[4] = {3, 4, 6, 3};
        int cin = 64;
        for (int st = 0; st < 4; st++) {
            const int mid = 64 << st, out = mid * 4;
            for (int b = 0; b < depth[st]; b++) {
                const int s = (b == 0 && st > 0) ? 2 : 1;
                add(cin, mid, 1, 1);
                add(mid, mid, 3, s);
                add(mid, out, 1, 1);
                if (b == 0) add(cin, out, 1, s);
                cin = out;
            }
        }
        fc_w = n_w; fc_b = n_b;
        n_w += (size_t)1000 * 2048;
        n_b += 1000;
    }
};
const Net& net() { static const Net n; return n; }

int k_padded(const Layer& l) { return l.ksize == 7 ? 7 * 8 * 4 : l.ksize * l.ksize * l.cin; }   // the stem: [7 ky][8 kx][4 c], kx = 7 and c = 3 zero

// stem == true: x is the bordered input image, (hin, win, cin, ksize, stride) describe the 7x7/2 convolution
int launch_conv(avd_ctx* ctx, const uint16_t* x, const uint16_t* w, const float* bias, const uint16_t* res, uint16_t* y, int n, int hin,
                int win, int cin, int cout, int ksize, int stride, int relu, bool stem = false)
{
    ConvGeom g;
    g.hin = hin; g.win = win; g.cin = cin; g.cout = cout; g.ksize = ksize; g.stride = stride; g.pad = ksize / 2;
    g.hout = (hin + 2 * g.pad - ksize) / stride + 1;
    g.wout = (win + 2 * g.pad - ksize) / stride + 1;
    g.m_out = n * g.hout * g.wout;
    g.cpb = stem ? 1 : cin / 32;
    g.nh = stem ? 7 : ksize * ksize * g.cpb;
    if (!stem && (cin % 32 || cout % 64 || (ksize != 1 && ksize != 3))) { ctx->err = "conv: cin % 32, cout % 64, ksize 1 or 3"; return AVD_ERR_ARG; }
    auto go = [&](auto kern, int bm, int bn) -> int {
        const int total = ((g.m_out + bm - 1) / bm) * (cout / bn), grid = (total + 7) / 8 * 8;
        const size_t lds = (size_t)(g.nh < 4 ? g.nh : 4) * (size_t)(bm * 64 + bn * 64);   // short K: fewer ring slots, more workgroups per CU
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, ctx->stream, x, w, bias, res, y, g, relu);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    };
    if (stem) return go(k_conv_bf16<256, 8, 2, 1, 1>, 256, 64);
    // 256-pixel tiles for the long-K layers that fill the chip.  128 x 128 tiles (74 registers, 16 KiB per ring slot: several
    // workgroups per CU) where 256-pixel tiles would leave most of the chip idle (the 14 x 14 and 7 x 7 stages), and for the
    // short-K 1x1 layers, which move bytes rather than multiply: there the time goes to load / store latency, and
    // co-resident workgroups are what hides it.
    static const int env_force = [] { const char* e = std::getenv("AVD_CNN_TILES"); return e ? std::atoi(e) : 0; }();
    const int force = ctx->cnn_tiles ? ctx->cnn_tiles : env_force;   // avd_set_option "cnn_tiles" / AVD_CNN_TILES: 1 = always 256-pixel tiles, 2 = 128 x 128 wherever possible
    const int bn_big = cout % 256 == 0 ? 256 : cout % 128 == 0 ? 128 : 64;
    const int wgs_big = ((g.m_out + 255) / 256) * (cout / bn_big);
    const bool small_ok = cout % 128 == 0;
    // tuning knobs (A/B only): AVD_CNN_FILL = workgroups of the 256-pixel tiling, in percent of the CU count, below which the
    // 128 x 128 tiling is taken; AVD_CNN_SHORTK = largest number of half stages that counts as "short K"
    static const int fill_pct = [] { const char* e = std::getenv("AVD_CNN_FILL"); return e ? std::atoi(e) : 150; }();
    static const int short_k = [] { const char* e = std::getenv("AVD_CNN_SHORTK"); return e ? std::atoi(e) : 8; }();
    const bool want_small = wgs_big * 100 < ctx->num_cus * fill_pct || g.nh <= short_k;
    if (ksize == 3) {
        if (small_ok && force != 1 && (want_small || force == 2)) return go(k_conv_bf16<128, 4, 2, 0, 3>, 128, 128);
        if (bn_big == 256) return go(k_conv_bf16<256, 2, 8, 0, 3>, 256, 256);
        if (bn_big == 128) return go(k_conv_bf16<256, 4, 4, 0, 3>, 256, 128);
        return go(k_conv_bf16<256, 8, 2, 0, 3>, 256, 64);
    }
    if (small_ok && force != 1 && (want_small || force == 2)) return go(k_conv_bf16<128, 4, 2, 0, 1>, 128, 128);
    if (bn_big == 256) return go(k_conv_bf16<256, 2, 8, 0, 1>, 256, 256);
    if (bn_big == 128) return go(k_conv_bf16<256, 4, 4, 0, 1>, 256, 128);
    return go(k_conv_bf16<256, 8, 2, 0, 1>, 256, 64);
}

// conv2 (3x3, mid -> mid, stride s, ReLU) + conv3 (1x1, mid -> 4 mid, + residual, ReLU) of a bottleneck block in one launch
// (k_conv3_expand); mid = 64 (256-pixel tiles) or 128 (128-pixel tiles).  y must not be the 3x3's input.
bool can_fuse_expand(int mid) { return mid == 64 || mid == 128; }
int launch_conv3_expand(avd_ctx* ctx, const uint16_t* x, const uint16_t* w2, const float* b2, const uint16_t* w3, const float* b3,
                        const uint16_t* res, uint16_t* y, int n, int hin, int win, int mid, int stride)
{
    ConvGeom g;
    g.hin = hin; g.win = win; g.cin = mid; g.cout = mid; g.ksize = 3; g.stride = stride; g.pad = 1;
    g.hout = (hin + 2 - 3) / stride + 1;
    g.wout = (win + 2 - 3) / stride + 1;
    g.m_out = n * g.hout * g.wout;
    g.cpb = mid / 32;
    g.nh = 9 * g.cpb;
    if (!can_fuse_expand(mid) || !res || y == x) { ctx->err = "conv3_expand: mid 64 or 128, a residual, output apart from the input"; return AVD_ERR_ARG; }
    auto go = [&](auto kern, int bm, int bn) -> int {
        const int total = (g.m_out + bm - 1) / bm, grid = (total + 7) / 8 * 8;
        // the ring; W3 resident (mid = 64): the expanding layer's bias fits inside it, otherwise behind it.  The kernel's LDS plan
        // assumes which ring slot the 3x3's last step reads: nh & 3 == 2 (mid = 64: 18 half stages) or 0 (mid = 128: 36)
        if ((g.nh & 3) != (bn == 64 ? 2 : 0)) { ctx->err = "conv3_expand: ring phase"; return AVD_ERR_ARG; }
        const size_t lds = (size_t)4 * (size_t)(bm * 64 + bn * 64) + (bn == 64 ? 0 : (size_t)bn * 4 * sizeof(float));
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, ctx->stream, x, w2, b2, w3, b3, res, y, g);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    };
    if (stride == 1 && ctx->cnn_fuse == 2) {
        // the 3x3's input as one slab in LDS (k_slab3_expand); the blocks of the input activation that exist: its rows are padded to 256
        const int in_blocks = (int)(((size_t)n * hin * win + 255) / 256 * 256 / 16);
        const int bm = mid == 64 ? 256 : 128;
        if (win + 1 > (mid == 64 ? 64 : 32)) { ctx->err = "conv3_expand: the slab's halo is sized for 56 x 56 (mid 64) and 28 x 28 (mid 128)"; return AVD_ERR_ARG; }
        const int total = (g.m_out + bm - 1) / bm, grid = (total + 7) / 8 * 8;
        const size_t lds = 49152 + 4 * 8192;
        auto slab = [&](auto kern) -> int {
            HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, ctx->stream, x, w2, b2, w3, b3, res, y, g, in_blocks);
            HIP_TRY(ctx, hipGetLastError());
            return 0;
        };
        return mid == 64 ? slab(k_slab3_expand<64>) : slab(k_slab3_expand<128>);
    }
    if (mid == 64) return go(k_conv3_expand<256, 8, 2>, 256, 64);
    return go(k_conv3_expand<128, 4, 2>, 128, 128);
}

size_t act_elems(size_t rows, int c) { return kZeroPage + (rows + 255) / 256 * 256 * (size_t)c; }

}  // namespace

void cnn_param_counts(size_t* n_w, size_t* n_b) { *n_w = net().n_w; *n_b = net().n_b; }

// flat parameters (conv weights [cout][kh][kw][cin] bf16 in network order, then the linear layer [1000][2048]; biases
// f32 in the same order) -> device: every convolution's weight re-tiled into the blocked operand layout
int cnn_set_weights(avd_ctx* ctx, const uint16_t* w, const float* b)
{
    const Net& nt = net();
    Workspace& ws = ctx->ws;
    size_t blocked_total = 0;
    for (const Layer& l : nt.convs) blocked_total += (size_t)l.cout * k_padded(l);
    std::vector<uint16_t> host(blocked_total + (size_t)1000 * 2048), tmp;
    size_t off = 0;
    ws.cnn_w_off.clear();
    for (const Layer& l : nt.convs) {
        const int K = k_padded(l), kin = l.ksize * l.ksize * l.cin;
        const uint16_t* src = w + l.w_off;
        if (K != kin) {                                      // the stem: [cout][7][7][3] -> [cout][7 ky][8 kx][4 c], the extra taps zero
            tmp.assign((size_t)l.cout * K, 0);
            for (int o = 0; o < l.cout; o++)
                for (int ky = 0; ky < 7; ky++)
                    for (int kx = 0; kx < 7; kx++)
                        for (int c = 0; c < 3; c++) tmp[(size_t)o * K + (ky * 8 + kx) * 4 + c] = src[(size_t)o * kin + (ky * 7 + kx) * 3 + c];
            src = tmp.data();
        }
        gemm_block_operand(src, host.data() + off, l.cout, K);
        ws.cnn_w_off.push_back(off);
        off += (size_t)l.cout * K;
    }
    ws.cnn_fc_off = off;
    for (size_t i = 0; i < (size_t)1000 * 2048; i++) host[off + i] = w[nt.fc_w + i];
    if (!ws.d_cnn_w) if (int e = dev_alloc(ctx, ws.d_cnn_w, host.size())) return e;
    if (!ws.d_cnn_b) if (int e = dev_alloc(ctx, ws.d_cnn_b, nt.n_b)) return e;
    HIP_TRY(ctx, hipMemcpyAsync(ws.d_cnn_w, host.data(), host.size() * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ws.d_cnn_b, b, nt.n_b * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AVD_OK;
}

// activation scratch for n frames: four rotating activations, the zero-bordered 224 x 224 input, pooled features, logits
int cnn_reserve(avd_ctx* ctx, int n)
{
    Workspace& ws = ctx->ws;
    if (n <= ws.cnn_frames) return AVD_OK;
    ws.cnn_frames = 0;                                                   // not valid again until every buffer below exists
    const size_t act = act_elems((size_t)n * 112 * 112, 64);             // the largest activation (= n * 56 * 56 x 256)
    for (int i = 0; i < 4; i++) {
        if (int e = dev_alloc(ctx, ws.d_cnn_act[i], act)) return e;
        HIP_TRY(ctx, hipMemsetAsync(ws.d_cnn_act[i], 0, kZeroPage * sizeof(uint16_t), ctx->stream));
    }
    if (int e = dev_alloc(ctx, ws.d_cnn_img, (size_t)n * kImgSide * kImgSide * 4)) return e;
    HIP_TRY(ctx, hipMemsetAsync(ws.d_cnn_img, 0, (size_t)n * kImgSide * kImgSide * 4 * sizeof(uint16_t), ctx->stream));   // the zero border
    if (int e = dev_alloc(ctx, ws.d_cnn_pool, (size_t)n * 2048)) return e;
    if (int e = dev_alloc(ctx, ws.d_cnn_logits, (size_t)n * 1000)) return e;
    ws.cnn_frames = n;
    return AVD_OK;
}

// frames (device BGR) -> logits (device f32 [n][1000]); everything on ctx->stream
int launch_cnn_forward(avd_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, int64_t row_stride, int64_t frame_stride)
{
    const Net& nt = net();
    Workspace& ws = ctx->ws;
    const int64_t px = (int64_t)n * kSide * kSide;
    hipLaunchKernelGGL(k_cnn_input, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, ctx->stream, d_bgr, n, h, w, row_stride, frame_stride, ws.d_cnn_img);
    size_t li = 0;
    auto wptr = [&](size_t i) { return ws.d_cnn_w + ws.cnn_w_off[i]; };
    auto bptr = [&](size_t i) { return ws.d_cnn_b + nt.convs[i].b_off; };
    // the stem gathers straight from the bordered image (one half stage per kernel row)
    if (int e = launch_conv(ctx, ws.d_cnn_img, wptr(0), bptr(0), nullptr, ws.d_cnn_act[0], n, kSide, kSide, 3, 64, 7, 2, 1, true)) return e;
    li = 1;
    const int64_t pooled = (int64_t)n * 56 * 56 * 8;
    hipLaunchKernelGGL(k_maxpool3, dim3((unsigned)((pooled + 255) / 256)), dim3(256), 0, ctx->stream, ws.d_cnn_act[0], n, 112, 112, 64, ws.d_cnn_act[1]);
    int cur = 1, hgt = 56;                                   // index of the block input among the four rotating buffers
    const int depth[4] = {3, 4, 6, 3};
    int cin = 64;
    for (int st = 0; st < 4; st++) {
        const int mid = 64 << st, out = mid * 4;
        for (int b = 0; b < depth[st]; b++) {
            const int s = (b == 0 && st > 0) ? 2 : 1;
            const int t1 = (cur + 1) & 3, t2 = (cur + 2) & 3, sc = (cur + 3) & 3;
            uint16_t *x = ws.d_cnn_act[cur], *a1 = ws.d_cnn_act[t1], *a2 = ws.d_cnn_act[t2], *a3 = ws.d_cnn_act[sc];
            if (int e = launch_conv(ctx, x, wptr(li), bptr(li), nullptr, a1, n, hgt, hgt, cin, mid, 1, 1, 1)) return e;
            const int ho = hgt / s;
            const uint16_t* res = x;
            if (b == 0) {                                    // projection shortcut into a3
                if (int e = launch_conv(ctx, x, wptr(li + 3), bptr(li + 3), nullptr, a3, n, hgt, hgt, cin, out, 1, s, 0)) return e;
                res = a3;
            }
            if (ctx->cnn_fuse && can_fuse_expand(mid)) {     // conv2 + conv3 in one launch: a1 -> a2 (other workgroups still read a1's halo)
                if (int e = launch_conv3_expand(ctx, a1, wptr(li + 1), bptr(li + 1), wptr(li + 2), bptr(li + 2), res, a2, n, hgt, hgt, mid, s)) return e;
                cur = t2;
            } else {                                         // conv3 writes over a1
                if (int e = launch_conv(ctx, a1, wptr(li + 1), bptr(li + 1), nullptr, a2, n, hgt, hgt, mid, mid, 3, s, 1)) return e;
                if (int e = launch_conv(ctx, a2, wptr(li + 2), bptr(li + 2), res, a1, n, ho, ho, mid, out, 1, 1, 1)) return e;
                cur = t1;
            }
            li += b == 0 ? 4 : 3;
            hgt = ho; cin = out;
        }
    }
    hipLaunchKernelGGL(k_avgpool, dim3((unsigned)((n * 256 + 255) / 256)), dim3(256), 0, ctx->stream, ws.d_cnn_act[cur], n, 49, 2048, ws.d_cnn_pool);
    hipLaunchKernelGGL(k_linear, dim3((unsigned)(((n + 7) / 8 * 1000 + 3) / 4)), dim3(256), 0, ctx->stream, ws.d_cnn_pool, ws.d_cnn_w + ws.cnn_fc_off,
                       ws.d_cnn_b + nt.fc_b, n, 1000, ws.d_cnn_logits);
    HIP_TRY(ctx, hipGetLastError());
    return AVD_OK;
}

// ONE convolution on host tensors in NHWC order (test entry: the blocked layout stays an internal matter)
int cnn_conv_host(avd_ctx* ctx, const uint16_t* x, int n, int hin, int win, int cin, const uint16_t* w, const float* bias, int cout, int ksize,
                  int stride, int relu, const uint16_t* residual, uint16_t* y)
{
    if (cin % 32 || cout % 64 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || n <= 0) {
        ctx->err = "cnn_conv: cin % 32 == 0, cout % 64 == 0, ksize 1 or 3, stride 1 or 2";
        return AVD_ERR_ARG;
    }
    const int pad = ksize / 2, hout = (hin + 2 * pad - ksize) / stride + 1, wout = (win + 2 * pad - ksize) / stride + 1;
    const size_t min_ = (size_t)n * hin * win, mout = (size_t)n * hout * wout;
    const int K = ksize * ksize * cin;
    std::vector<uint16_t> hx(act_elems(min_, cin), 0), hw((size_t)cout * K), hr, hy(act_elems(mout, cout), 0);
    for (size_t m = 0; m < min_; m++)
        for (int c = 0; c < cin; c++) hx[kZeroPage + blocked_index((int)m, c, cin)] = x[m * cin + c];
    gemm_block_operand(w, hw.data(), cout, K);
    if (residual) {
        hr.assign(act_elems(mout, cout), 0);
        for (size_t m = 0; m < mout; m++)
            for (int c = 0; c < cout; c++) hr[kZeroPage + blocked_index((int)m, c, cout)] = residual[m * cout + c];
    }
    uint16_t *dx = nullptr, *dw = nullptr, *dr = nullptr, *dy = nullptr;
    float* db = nullptr;
    int rc = AVD_OK;
    auto cleanup = [&]() { (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dr); (void)hipFree(dy); (void)hipFree(db); };
#define CNN_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ctx->err = hipGetErrorString(e_); cleanup(); return AVD_ERR_DEVICE; } } while (0)
    CNN_TRY(hipMalloc(&dx, hx.size() * 2)); CNN_TRY(hipMalloc(&dw, hw.size() * 2)); CNN_TRY(hipMalloc(&dy, hy.size() * 2));
    CNN_TRY(hipMalloc(&db, cout * sizeof(float)));
    if (residual) CNN_TRY(hipMalloc(&dr, hr.size() * 2));
    CNN_TRY(hipMemcpyAsync(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    CNN_TRY(hipMemcpyAsync(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    CNN_TRY(hipMemcpyAsync(db, bias, cout * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    CNN_TRY(hipMemsetAsync(dy, 0, hy.size() * 2, ctx->stream));
    if (residual) CNN_TRY(hipMemcpyAsync(dr, hr.data(), hr.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_conv(ctx, dx, dw, db, dr, dy, n, hin, win, cin, cout, ksize, stride, relu);
    if (rc == AVD_OK) {
        CNN_TRY(hipMemcpyAsync(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
        CNN_TRY(hipStreamSynchronize(ctx->stream));
        for (size_t m = 0; m < mout; m++)
            for (int c = 0; c < cout; c++) y[m * cout + c] = hy[kZeroPage + blocked_index((int)m, c, cout)];
    }
#undef CNN_TRY
    cleanup();
    return rc;
}
