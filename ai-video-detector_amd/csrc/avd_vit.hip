// avd_vit.hip -- ViT-B/16 patch embedding on the matrix cores (gfx950): SURVEY.md section 8 row A10.
//
// BUILD-DEFINED EXTENSION: the reference has no learned model (its only per-frame "model" is the
// closed form of app/analyzers/video.py:54-56).  BASELINE.json's north_star / configs[3] name a "ViT-B/16 patch-embed
// MFMA path"; this file is that stage, with seeded random weights supplied by the caller, gated off from ai_score:
// nothing in the parity path calls it.  Its oracle is a float64 numpy restatement (tests/test_vit.py).
//
//   k_vit_patchify   BGR uint8 frame -> 224x224 bilinear (float taps, cv2's INTER_LINEAR centre mapping) -> RGB,
//                    (x/255 - mean)/std -> bf16 (round to nearest even), written as the im2col matrix of a 16x16/16
//                    convolution: A[frame*196 + patch][c*256 + py*16 + px]  (the order of a [768][3][16][16] conv weight)
//   k_gemm_bf16_nt   tokens[M][768] = A[M][768] x Wt[768][768]^T + bias, bf16 in, f32 accumulate / out.
//
// The GEMM: 256x256 output tile per 512-thread workgroup (8 waves as 2 x 4, a wave owns 128x64 = 8x4 MFMA tiles of
// v_mfma_f32_16x16x32_bf16: 128 accumulator registers).  Both operands are K-contiguous, so one staging routine serves
// both: global_load_lds_dwordx4 (16 bytes per lane straight into LDS, no VGPR round trip).  The staging unit is a HALF
// stage, 32 deep in K (256 rows x 64 B per operand, 32 KiB for both): a ring of four of them in 128 KiB of LDS keeps
// THREE in flight beside the one being read (a 64-deep double buffer would keep one, less than a loaded HBM round trip),
// with counted vmcnt + a raw s_barrier per half stage (a __syncthreads would drain the LDS-DMAs).  The LDS image is
// XOR-swizzled (16-byte chunk c of row r sits in slot c ^ ((r >> 2) & 3) of its 64-byte row), applied on the SOURCE
// address since the LDS side of an LDS-DMA is lane-linear; a ds_read_b128 of a fragment (16 rows x one chunk) then
// touches 16 distinct 16-byte bank groups.  The epilogue goes through LDS so that a store instruction writes whole
// 256-byte rows.  XCD-aware tile order: the three 256-column tiles of one 256-row block run on the same XCD, so a block
// of A is fetched from HBM once.
#include <cstdlib>
#include "avd_internal.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPatch = 16, kSide = 224, kGrid = kSide / kPatch, kTokens = kGrid * kGrid;   // 14 x 14 = 196 patches
constexpr int kDim = 3 * kPatch * kPatch;                                                  // 768

__device__ __forceinline__ uint16_t f32_to_bf16(float v)
{
    const unsigned u = __float_as_uint(v);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);       // round to nearest even (inputs are finite)
}

// one thread = one output pixel (x, y) of one frame, three channels
__global__ __launch_bounds__(256) void k_vit_patchify(const uint8_t* __restrict__ bgr, int n, int h, int w, int64_t row_stride,
                                                     int64_t frame_stride, uint16_t* __restrict__ A)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n * kSide * kSide) return;
    const int x = gid % kSide, y = (gid / kSide) % kSide, f = gid / (kSide * kSide);
    // cv2.resize INTER_LINEAR centre mapping with float weights (same formulas as the 32F path of resize.cpp)
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
    const int patch = (y / kPatch) * kGrid + x / kPatch;
    uint16_t* out = A + ((int64_t)f * kTokens + patch) * kDim + (y % kPatch) * kPatch + (x % kPatch);
#pragma unroll
    for (int c = 0; c < 3; c++) {                       // c = RGB channel index; the frame is BGR
        const int s = 2 - c;
        const float top = p00[s] + (p01[s] - (float)p00[s]) * fx, bot = p10[s] + (p11[s] - (float)p10[s]) * fx;
        const float v = top + (bot - top) * fy;
        out[c * kPatch * kPatch] = f32_to_bf16((v * (1.f / 255.f) - mean[c]) * istd[c]);
    }
}

constexpr int BM = 256, BN = 256, BKH = 32;               // a staging unit ("half stage") is 32 deep in K
constexpr int kHalfTile = BM * BKH * 2;                     // 16 KiB: one operand, 256 rows x 64 B
constexpr int kStage = 2 * kHalfTile;                       // A half tile | B half tile
constexpr int kStages = 4;                                  // ring: kStages - 1 half stages in flight beside the one being read
                                                            // (5 = all 160 KiB of LDS measured no faster than 4)
constexpr int kGemmLds = kStages * kStage;                  // 128 KiB

// Bank swizzle of a half tile (64-byte rows, four 16-byte chunks per row, four rows per 256-byte bank row): chunk c of
// row r sits in slot c ^ swz((r >> 2) & 3).  A ds_read_b128 is served in groups of 16 lanes that are NOT contiguous
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...): with lane = (chunk << 4) | row a group reads rows {0-3, 12-15} of
// one chunk and rows {4-11} of the next, and swz = {0, 2, 3, 1} is what makes those sixteen accesses hit sixteen
// different 16-byte bank groups (the plain XOR with (r >> 2) & 3 is two-way conflicted for these groups).
__device__ __forceinline__ int swz(int k) { return k ? (k % 3) + 1 : 0; }

// stage one operand half tile (256 rows x 32 k = 64 B per row) into LDS: wave `wave` issues instructions 2*wave and
// 2*wave+1, each 16 rows.
__device__ __forceinline__ void stage_half(const uint16_t* __restrict__ src, int row0, int rows_total, int K, int k0,
                                           char* lds_tile, int wave, int lane)
{
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int inst = wave * 2 + i;
        const int r = inst * 16 + (lane >> 2);                         // tile row this lane fills
        const int chunk = (lane & 3) ^ swz((r >> 2) & 3);              // which 16-byte chunk of the row lands in slot lane & 3
        int gr = row0 + r;
        gr = gr < rows_total ? gr : rows_total - 1;                    // rows past the end repeat the last one (never stored)
        const uint16_t* g = src + (int64_t)gr * K + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lds_tile + inst * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 frag(const char* lds_tile, int row, int chunk)
{
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 64 + ((chunk ^ swz((row >> 2) & 3)) << 4));
}

template <int K, int DBG>
__global__ __launch_bounds__(512) void k_gemm_bf16_nt(const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bt,
                                                     const float* __restrict__ bias, float* __restrict__ C, int M, int N)
{
    constexpr int dbg = DBG;                               // timing experiments (AVD_GEMM_DBG), 0 in production
    extern __shared__ __align__(16) char lds[];            // ring of 4 half stages [A | B]; reused by the epilogue
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM, total = tiles_m * tiles_n;
    // consecutive logical tiles (the N tiles of one M block) on one XCD
    const int per = (gridDim.x + 7) >> 3;
    const int lid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (lid >= total) return;
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    constexpr int NH = K / BKH;                            // half stages of this tile (the loop below is fully unrolled)

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int hs) __attribute__((always_inline)) {
        char* st = lds + (hs % kStages) * kStage;
        if (dbg & 1) return;                               // timing experiment: no global -> LDS traffic
        stage_half(A, m0, M, K, hs * BKH, st, wave, lane);
        stage_half(Bt, n0, N, K, hs * BKH, st + kHalfTile, wave, lane);
    };
    // Software pipeline over half stages: while the MFMAs of half stage hs run from REGISTERS, the fragments of hs + 1
    // are read from LDS (the compiler interleaves the two: they are independent) and hs + 2 .. hs + 4 are in flight as
    // LDS-DMAs.  A half stage's LDS slot is free as soon as every wave holds its fragments, i.e. at the next barrier.
    // A wave issues 4 LDS-DMA instructions per half stage: vmcnt(4 * y) = "all but the y youngest half stages landed".
    auto wait_landed = [&](int hs, int issued_last) __attribute__((always_inline)) {      // hs must be readable afterwards
        __builtin_amdgcn_sched_barrier(0);
        const int younger = issued_last - hs;                // half stages issued after hs (a wave: 4 instructions each)
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);                   // nothing (LDS reads, MFMAs) may be scheduled across the wait + barrier
    };
    auto load_frags = [&](bf16x8 (&a)[8], bf16x8 (&b)[4], int hs) __attribute__((always_inline)) {
        const char* cur = lds + (hs % kStages) * kStage;
        const int chunk = lane >> 4;
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = frag(cur, wm * 128 + i * 16 + (lane & 15), chunk);
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = frag(cur + kHalfTile, wn * 64 + j * 16 + (lane & 15), chunk);
    };
    auto mfma_all = [&](const bf16x8 (&a)[8], const bf16x8 (&b)[4]) __attribute__((always_inline)) {
        if (dbg & 2) {                                      // timing experiment: no MFMA (keep the fragments alive)
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" ::"v"(a[i]));
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" ::"v"(b[j]));
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    bf16x8 a0[8], b0[4], a1[8], b1[4];
#pragma unroll
    for (int i = 0; i < kStages - 1; i++)
        if (i < NH) issue(i);
    wait_landed(0, NH - 1 < kStages - 2 ? NH - 1 : kStages - 2);
    if (kStages - 1 < NH) issue(kStages - 1);
    load_frags(a0, b0, 0);
    // one step: fragments of hs are in (ac, bc); fetch hs + 1 into (an, bn) while multiplying
    auto step = [&](const bf16x8 (&ac)[8], const bf16x8 (&bc)[4], bf16x8 (&an)[8], bf16x8 (&bn)[4], int hs) __attribute__((always_inline)) {
        if (hs + 1 < NH) {
            // stage hs + 1 readable; the barrier also says every wave has the fragments of hs in registers, so the
            // slot of hs is refilled with hs + kStages
            wait_landed(hs + 1, hs + kStages - 1 < NH - 1 ? hs + kStages - 1 : NH - 1);
            if (hs + kStages < NH) issue(hs + kStages);
            load_frags(an, bn, hs + 1);
        }
        mfma_all(ac, bc);
    };
#pragma unroll
    for (int hs = 0; hs < NH; hs += 2) {
        step(a0, b0, a1, b1, hs);
        if (hs + 1 < NH) step(a1, b1, a0, b0, hs + 1);
    }
    // ---- epilogue: through LDS, so that a store instruction writes four whole 256-byte rows of the wave's 128 x 64
    // block (the accumulator layout -- column = lane & 15, rows (lane >> 4) * 4 + r -- would give 64-byte pieces and four
    // times as many store instructions)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // every wave is done with the ring
    constexpr int PITCH = 68;                              // floats per staged row (64 + pad: 16-byte aligned, 2-way banks)
    float* reg = reinterpret_cast<float*>(lds) + wave * (32 * PITCH);
    const int ccol = n0 + wn * 64 + (lane & 15) * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ccol);
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {                 // 32 rows of the wave's block per pass (2 MFMA row tiles)
#pragma unroll
        for (int ii = 0; ii < 2; ii++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    reg[(ii * 16 + (lane >> 4) * 4 + r) * PITCH + j * 16 + (lane & 15)] = acc[pass * 2 + ii][j][r];
        // a wave reads back only what it wrote itself: its own LDS operations are ordered
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int rl = q * 4 + (lane >> 4);
            const int row = m0 + wm * 128 + pass * 32 + rl;
            f32x4 v = *reinterpret_cast<const f32x4*>(reg + rl * PITCH + (lane & 15) * 4);
            v += bv;
            if (row < M && !(dbg & 4)) *reinterpret_cast<f32x4*>(C + (int64_t)row * N + ccol) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form of the same GEMM: one workgroup per CU walks a list of tiles and NEVER drains its pipeline.
//  * the half-stage stream is continuous across tiles: during the last four steps of a tile the LDS-DMAs of the next
//    tile's first half stages are already issued -- no prologue latency, no drained ring behind the epilogue;
//  * the product is formed TRANSPOSED (mfma(b, a): the accumulator tile has n on its rows and m on the lane), so a lane
//    holds four consecutive COLUMNS of one output row: the epilogue is 32 plain 16-byte (f32) / 8-byte (bf16) stores
//    per wave straight from the accumulators, no LDS staging, and the ring stays free for the next tile;
//  * loads and stores of a wave share one in-order vmcnt: the waits of the three steps after an epilogue allow the 32
//    stores to be outstanding (vmcnt(8 + 32)) -- the half stages they wait for were issued BEFORE the stores; from the
//    fourth step on the stores are older than what is waited for and have long completed;
//  * the bias sits in LDS (3 KiB) so that the epilogue issues no loads.
// Tile order: logical workgroup id = (blockIdx % 8) * (grid / 8) + blockIdx / 8 and tile = id + i * grid: the three
// 256-column tiles of a 256-row block of A run at the same time on one XCD.
// ---------------------------------------------------------------------------------------------------------------
template <int K, int OUT_BF16>
__global__ __launch_bounds__(512) void k_gemm_bf16_nt_persistent(const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bt,
                                                                const float* __restrict__ bias, void* __restrict__ Cv, int M, int N)
{
    extern __shared__ __align__(16) char lds[];            // ring of kStages half stages [A | B], then the bias (N floats)
    constexpr int NH = K / BKH;
    static_assert(NH >= kStages + 1 && kStages == 4, "the vmcnt immediates below assume a ring of four");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM, total = tiles_m * tiles_n;
    const int lw = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (lw >= total) return;
    float* lbias = reinterpret_cast<float*>(lds + kGemmLds);
    for (int i = threadIdx.x; i < N; i += 512) lbias[i] = bias ? bias[i] : 0.f;

    f32x4 acc[8][4];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // LDS-DMA addressing: a uniform tile pointer (scalar registers) + a per-lane 32-bit byte offset that is the same
    // for every tile, half stage and operand: row (2 * wave + i) * 16 + lane / 4 of the tile, swizzled chunk.  A is padded
    // to whole tiles by the caller, so no row needs clamping.
    unsigned voff[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = (wave * 2 + i) * 16 + (lane >> 2);
        voff[i] = (unsigned)(r * K + (((lane & 3) ^ swz((r >> 2) & 3)) << 3)) * 2u;
    }
    auto issue = [&](int m0, int n0, int hs, int slot) __attribute__((always_inline)) {
        char* st = lds + slot * kStage;
        const char* pa = reinterpret_cast<const char*>(A + (int64_t)m0 * K + hs * BKH);
        const char* pb = reinterpret_cast<const char*>(Bt + (int64_t)n0 * K + hs * BKH);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + voff[i]),
                                             (__attribute__((address_space(3))) void*)(st + (wave * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + voff[i]),
                                             (__attribute__((address_space(3))) void*)(st + kHalfTile + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };
    // the product is formed transposed: rows of a 16x16 result = n (B fragment as the first operand), lane column = m
    auto store_tile = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;          // four consecutive columns of this lane
            const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias + col);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int row = m0 + wm * 128 + i * 16 + (lane & 15);
                const f32x4 v = acc[i][j] + bv;
                if (row < M) {
                    if (OUT_BF16) {
                        uint2 pk;
                        pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                        pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(Cv) + (int64_t)row * N + col) = pk;
                    } else {
                        *reinterpret_cast<f32x4*>(static_cast<float*>(Cv) + (int64_t)row * N + col) = v;
                    }
                }
            }
        }
    };

    int tile = lw;
    int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    // prologue of the first tile only: three half stages in flight
#pragma unroll
    for (int hs = 0; hs < kStages - 1; hs++) issue(m0, n0, hs, hs);
    zero_acc();
    bool first = true;
    for (;;) {
        const int nxt = tile + gridDim.x;
        const bool more = nxt < total;
        const int m1 = more ? (nxt / tiles_n) * BM : m0, n1 = more ? (nxt % tiles_n) * BN : n0;
        // Step hs.  A half stage's ring slot is hs mod 4; NH is a multiple of 4, so the next tile's half stage 0 lands in
        // slot 0 again and the stream of half stages runs on across tiles.
#pragma unroll
        for (int hs = 0; hs < NH; hs++) {
            // half stage hs must have landed.  Younger operations of this wave: the two half stages behind it (8 LDS-DMAs),
            // plus the 32 stores of the previous tile's epilogue during the first three steps of a later tile (the half
            // stages waited for there were issued BEFORE those stores).  At the end of the stream there are fewer.
            __builtin_amdgcn_sched_barrier(0);
            if (hs < 3 && !first) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            else if (hs == NH - 2 && !more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (hs == NH - 1 && !more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            // ... for every wave; the same barrier says everyone has finished reading half stage hs - 1, whose slot is
            // refilled with stream position hs + 3
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);               // nothing (LDS reads, MFMAs) may be scheduled across the wait + barrier
        __builtin_amdgcn_sched_barrier(0);                   // nothing (LDS reads, MFMAs) may be scheduled across the wait + barrier
            if (hs + kStages - 1 < NH) issue(m0, n0, hs + kStages - 1, (hs + kStages - 1) % kStages);
            else if (more) issue(m1, n1, hs + kStages - 1 - NH, (hs + kStages - 1) % kStages);
            // fragments in two halves: the reads of A rows 64..127 are in flight while the first 16 MFMAs run (a single
            // "read all, wait, multiply" leaves the matrix pipe idle for an LDS round trip per step: both waves of a SIMD
            // come out of the barrier together)
            {
                const char* cur = lds + (hs % kStages) * kStage;
                const int chunk = lane >> 4, r16 = lane & 15;
                bf16x8 b[4], alo[4], ahi[4];
#pragma unroll
                for (int j = 0; j < 4; j++) b[j] = frag(cur + kHalfTile, wn * 64 + j * 16 + r16, chunk);
#pragma unroll
                for (int i = 0; i < 4; i++) alo[i] = frag(cur, wm * 128 + i * 16 + r16, chunk);
#pragma unroll
                for (int i = 0; i < 4; i++) ahi[i] = frag(cur, wm * 128 + (i + 4) * 16 + r16, chunk);
                __builtin_amdgcn_sched_barrier(0);          // all twelve reads are issued before the first MFMA
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], alo[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i + 4][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], ahi[i], acc[i + 4][j], 0, 0, 0);
            }
        }
        store_tile(m0, n0);
        if (!more) break;
        zero_acc();
        tile = nxt; m0 = m1; n0 = n1;
        first = false;
    }
}

}  // namespace

// tokens[M][768] (device, f32) = patchify(frames) x Wt^T + bias.  d_wt: bf16 [768 out][768 k]; d_bias f32[768] or null.
int launch_vit_patch_embed(avd_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, int64_t row_stride, int64_t frame_stride,
                           const uint16_t* d_wt, const float* d_bias, void* d_tokens, int tokens_bf16, uint16_t* d_patches)
{
    if (n <= 0) return 0;
    const int64_t px = (int64_t)n * kSide * kSide;
    hipLaunchKernelGGL(k_vit_patchify, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, ctx->stream, d_bgr, n, h, w, row_stride,
                       frame_stride, d_patches);
    return launch_gemm_bf16_nt(ctx, d_patches, d_wt, d_bias, d_tokens, tokens_bf16, n * kTokens, kDim, kDim);
}

int launch_gemm_bf16_nt(avd_ctx* ctx, const uint16_t* d_a, const uint16_t* d_bt, const float* d_bias, void* d_c, int out_bf16,
                        int M, int N, int K)
{
    if (M <= 0) return 0;
    if (N % BN || K != kDim || N > 4096) { ctx->err = "gemm_bf16_nt: N must be a multiple of 256 (<= 4096) and K = 768"; return AVD_ERR_ARG; }
    static const int dbg = [] { const char* e = std::getenv("AVD_GEMM_DBG"); return e ? std::atoi(e) : 0; }();   // timing experiments only
    static const int variant = [] { const char* e = std::getenv("AVD_GEMM_VARIANT"); return e ? std::atoi(e) : 0; }();   // f32 tokens: 0 = one tile per workgroup (measured faster), 1 = persistent
    const int total = ((M + BM - 1) / BM) * (N / BN);
    auto go = [&](auto kern, int grid, size_t lds, auto... args) -> int {
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, ctx->stream, args...);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    };
    if (variant == 1 || out_bf16) {
        // one workgroup per CU (LDS: ring + bias), grid a multiple of 8 so that the XCD-aware order is a bijection
        int grid = ctx->num_cus / 8 * 8;
        if (grid < 8) grid = 8;
        if (grid > (total + 7) / 8 * 8) grid = (total + 7) / 8 * 8;
        const size_t lds = kGemmLds + (size_t)N * sizeof(float);
        if (out_bf16) return go(k_gemm_bf16_nt_persistent<kDim, 1>, grid, lds, d_a, d_bt, d_bias, d_c, M, N);
        return go(k_gemm_bf16_nt_persistent<kDim, 0>, grid, lds, d_a, d_bt, d_bias, d_c, M, N);
    }
    const int grid = (total + 7) / 8 * 8;
    float* c32 = static_cast<float*>(d_c);
    switch (dbg) {
#ifdef AVD_GEMM_DEBUG
    case 1: return go(k_gemm_bf16_nt<kDim, 1>, grid, (size_t)kGemmLds, d_a, d_bt, d_bias, c32, M, N);
    case 2: return go(k_gemm_bf16_nt<kDim, 2>, grid, (size_t)kGemmLds, d_a, d_bt, d_bias, c32, M, N);
    case 4: return go(k_gemm_bf16_nt<kDim, 4>, grid, (size_t)kGemmLds, d_a, d_bt, d_bias, c32, M, N);
    case 5: return go(k_gemm_bf16_nt<kDim, 5>, grid, (size_t)kGemmLds, d_a, d_bt, d_bias, c32, M, N);
    case 6: return go(k_gemm_bf16_nt<kDim, 6>, grid, (size_t)kGemmLds, d_a, d_bt, d_bias, c32, M, N);
    case 7: return go(k_gemm_bf16_nt<kDim, 7>, grid, (size_t)kGemmLds, d_a, d_bt, d_bias, c32, M, N);
#endif
    default: return go(k_gemm_bf16_nt<kDim, 0>, grid, (size_t)kGemmLds, d_a, d_bt, d_bias, c32, M, N);
    }
}
