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
//   k_gemm_bf16_nt   tokens[M][768] = A[M][768] x Wt[768][768]^T + bias, bf16 in, f32 accumulate, f32 or bf16 out.
//
// OPERAND LAYOUT IN HBM ("blocked"): both GEMM operands are K-contiguous matrices stored as 1-KiB blocks of 16 rows x 32
// k, block (row / 16, k / 32) at ((row / 16) * (K / 32) + k / 32) * 1 KiB, and INSIDE a block exactly the bytes of the LDS
// image the MFMA fragments are read from: row r at r * 64, its four 16-byte chunks XOR-swizzled (chunk c in slot
// c ^ swz((r >> 2) & 3)).  The patchify kernel writes A that way and avd_vit_set_weights re-tiles the row-major weight
// once on upload, so that one global_load_lds_dwordx4 wave-instruction (16 bytes per lane straight into LDS, no VGPR
// round trip) copies ONE contiguous, 128-byte-aligned KiB: eight whole cache lines.  With row-major operands the same
// instruction touched sixteen half lines (64 bytes of each of 16 rows) and the L2 -> LDS path, not the matrix pipe, set
// the kernel's time (profiles/r02_experiments.md).
//
// The GEMM: a 512-thread workgroup per CU walks 256 x 256 output tiles; 8 waves, a wave owns 8 x 4
// MFMA tiles of v_mfma_f32_16x16x32_bf16.  The staging unit is a HALF stage, 32 deep in K (64 B per row, 32 KiB for both
// operands of a 256 x 256 tile): a ring of four of them in LDS keeps THREE in flight beside the one being read, with
// counted vmcnt + a raw s_barrier per half stage (a __syncthreads would drain the LDS-DMAs).  A ds_read_b128 of a
// fragment (16 rows x one chunk) touches 16 distinct 16-byte bank groups thanks to the swizzle.
#include <cstdlib>
#include <type_traits>
#include "avd_internal.h"
#include "avd_mfma_device.h"

namespace {

using namespace avd_mfma;

constexpr int kPatch = 16, kSide = 224, kGrid = kSide / kPatch, kTokens = kGrid * kGrid;   // 14 x 14 = 196 patches
constexpr int kDim = 3 * kPatch * kPatch;                                                  // 768

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
    const int row = f * kTokens + patch, k0 = (y % kPatch) * kPatch + (x % kPatch);
#pragma unroll
    for (int c = 0; c < 3; c++) {                       // c = RGB channel index; the frame is BGR
        const int s = 2 - c;
        const float top = p00[s] + (p01[s] - (float)p00[s]) * fx, bot = p10[s] + (p11[s] - (float)p10[s]) * fx;
        const float v = top + (bot - top) * fy;
        A[blocked_index(row, c * kPatch * kPatch + k0, kDim)] = f32_to_bf16((v * (1.f / 255.f) - mean[c]) * istd[c]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The GEMM kernel is persistent: one workgroup per CU walks a list of tiles and NEVER drains its pipeline.
//  * the tile SHAPE is a template parameter (TileShape); 256 x 256 is the one instantiated -- 384 x 192, one round of
//    tiles for a 120-frame clip instead of two, needs 144 accumulator registers per wave and spills: slower at every size;
//  * the half-stage stream is continuous across tiles: during the last steps of a tile the LDS-DMAs of the next tile's
//    first half stages are already issued -- no prologue latency, no drained ring behind the epilogue;
//  * the product is formed TRANSPOSED (mfma(b, a): the accumulator tile has n on its rows and m on the lane), so a lane
//    holds four consecutive COLUMNS of one output row: the epilogue is plain 16-byte (f32) / 8-byte (bf16) stores
//    straight from the accumulators, no LDS staging, and the ring stays free for the next tile;
//  * loads and stores of a wave share one in-order vmcnt: the waits of the three steps after an epilogue allow the
//    stores to be outstanding (vmcnt(2 P + stores)) -- the half stages they wait for were issued BEFORE the stores; from
//    the fourth step on the stores are older than what is waited for and have long completed;
//  * the bias sits in LDS (3 KiB) so that the epilogue issues no loads.
// Tile order: logical workgroup id = (blockIdx % 8) * (grid / 8) + blockIdx / 8 and tile = id + i * grid: the column
// tiles of one row block of A run at the same time on one XCD.
// ---------------------------------------------------------------------------------------------------------------
// NW waves per workgroup: 8 (round 2: a wave owns 8 x 4 MFMA tiles) or 16 (round 5: 4 x 4 tiles).  What a CU ingests through LDS-DMA is set by
// how many WAVES issue it, not by the bytes in flight (tools/ldsdma_occ_bench.hip: 31-35 / 60-72 / 92-99 GB/s per CU from L2 with 4 / 8 / 16
// waves, 2 to 18 KiB in flight per wave alike) -- so the operand stream of the same 256 x 256 tile runs half as long with sixteen waves.
template <int WAVES_M, int TI, int TJ, int NW = 8, bool SKEW_ = true>
struct TileShape {
    static constexpr int NWAVES = NW;
    static constexpr bool SKEW = SKEW_;                         // half of each SIMD's waves run half a step late (see the kernel)
    static constexpr int WAVES_N = NW / WAVES_M, TI_ = TI, TJ_ = TJ;
    static constexpr int BM = WAVES_M * TI * 16, BN = WAVES_N * TJ * 16;
    static constexpr int RA = BM / NW, RB = BN / NW;            // rows of each operand that one wave stages per half stage
    static_assert(RA % 8 == 0 && RB % 8 == 0 && TI % 2 == 0, "a wave stages whole or half LDS-DMA instructions (16 / 8 rows)");
    static constexpr int QA = (RA + 15) / 16, QB = (RB + 15) / 16;   // LDS-DMA instructions (the last one may be half masked)
    static constexpr int P = QA + QB;                           // ... per wave and half stage
    static constexpr int HALF_A = BM * BKH * 2, HALF_B = BN * BKH * 2, STAGE = HALF_A + HALF_B;
    static constexpr int LDS = kStages * STAGE;
    static_assert(TJ % 2 == 0 && 2 * P + TI * TJ <= 63, "column tiles pair up in the epilogue; vmcnt is six bits");
};

// R rows per wave of one operand half tile: instruction q covers rows wave * R + q * 16 + lane / 4; when R is an odd
// multiple of 8 the last instruction is issued for the lower 32 lanes only (8 rows).
template <int R>
__device__ __forceinline__ void stage_rows(const char* tile, const unsigned (&voff)[(R + 15) / 16], char* lds_half, int wave, int lane)
{
    constexpr int Q = (R + 15) / 16;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        char* dst = lds_half + (wave * R + q * 16) * 64;
        if (q * 16 + 16 <= R) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile + voff[q]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        } else if (lane < 32) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile + voff[q]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    }
}

// per-lane byte offset, relative to the tile's first block of a half stage, of the 16 bytes a lane copies with
// instruction q: tile row r = wave * R + q * 16 + lane / 4 lives in block r / 16 (blocks of one half stage are K / 32 KiB
// apart), at r % 16 * 64 + (lane % 4) * 16 inside it -- the swizzle is already in the data
template <int R>
__device__ __forceinline__ void stage_offsets(unsigned (&voff)[(R + 15) / 16], int K, int wave, int lane)
{
#pragma unroll
    for (int q = 0; q < (R + 15) / 16; q++) {
        const int r = wave * R + q * 16 + (lane >> 2);
        voff[q] = (unsigned)((r >> 4) * (K >> 5) * 1024 + (r & 15) * 64 + (lane & 3) * 16);
    }
}

#define AVD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int K, int OUT_BF16, class T, int DBG = 0>
__global__ __launch_bounds__(64 * T::NWAVES) void k_gemm_bf16_nt_persistent(const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bt,
                                                                const float* __restrict__ bias, void* __restrict__ Cv, int M, int N_)
{
    constexpr int N = K;                                   // square weight (the patch embedding): offsets fold into immediates
    extern __shared__ __align__(16) char lds[];            // ring of kStages half stages [A | B], then the bias (N floats)
    constexpr int NH = K / BKH, TI = T::TI_, TJ = T::TJ_;
    static_assert(NH >= kStages + 1 && NH % kStages == 0 && kStages == 4, "the waits below assume a ring of four");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // in scalar registers: everything derived from it is uniform
    const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
    const int tiles_n = N / T::BN, tiles_m = (M + T::BM - 1) / T::BM, total = tiles_m * tiles_n;
    const int lw = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (lw >= total) return;
    float* lbias = reinterpret_cast<float*>(lds + T::LDS);
    for (int i = threadIdx.x; i < N; i += 64 * T::NWAVES) lbias[i] = bias ? bias[i] : 0.f;

    f32x4 acc[TI][TJ];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // LDS-DMA addressing: a uniform pointer to the tile's first block of the half stage (scalar registers) + a per-lane
    // 32-bit byte offset that is the same for every tile and half stage.  A is padded to whole tiles by the caller, so no
    // row needs clamping.
    unsigned voa[T::QA], vob[T::QB];
    stage_offsets<T::RA>(voa, K, wave, lane);
    stage_offsets<T::RB>(vob, K, wave, lane);
    auto issue = [&](int m0, int n0, int hs, int slot) __attribute__((always_inline)) {
        if (DBG & 1) return;                                 // timing experiment: no global -> LDS traffic
        char* st = lds + slot * T::STAGE;
        stage_rows<T::RA>(reinterpret_cast<const char*>(A + ((int64_t)(m0 >> 4) * (K >> 5) + hs) * 512), voa, st, wave, lane);
        stage_rows<T::RB>(reinterpret_cast<const char*>(Bt + ((int64_t)(n0 >> 4) * (K >> 5) + hs) * 512), vob, st + T::HALF_A, wave, lane);
    };
    // the product is formed transposed: rows of a 16x16 result = n (B fragment as the first operand), lane column = m
    // Which weight row feeds which MFMA row is free to choose: MFMA row rho of column tile j takes the wave's local
    // column (j / 2) * 32 + (rho / 4) * 8 + (j % 2) * 4 + rho % 4, so that a lane (accumulator rows (lane / 16) * 4 + r of
    // tiles 2 jp and 2 jp + 1) owns EIGHT consecutive columns: one 16-byte store of bf16 tokens per lane, 64 contiguous
    // bytes per output row and instruction (the natural order gives 8-byte stores, 32 contiguous bytes).  The fragment
    // reads stay conflict-free: the four row groups of a read have swizzle keys (0, 2, 0, 2) + j % 2, and the lane groups
    // a ds_read_b128 is served in ({0-3, 12-15} of one chunk, {4-11} of the next) still land in four distinct slots.
    auto b_row = [&](int j, int rho) __attribute__((always_inline)) {
        return wn * (TJ * 16) + (j >> 1) * 32 + (rho >> 2) * 8 + (j & 1) * 4 + (rho & 3);
    };
    constexpr int ESZ = OUT_BF16 ? 2 : 4;
    const unsigned coff = (unsigned)((lane & 15) * N + (lane >> 4) * 8) * ESZ;     // this lane inside a 16-row x 32-column piece
    auto store_tile = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TI; i++) {
            const int row0 = m0 + (wm * TI + i) * 16;                             // uniform
#pragma unroll
            for (int jp = 0; jp < TJ / 2; jp++) {
                const int col0 = n0 + wn * (TJ * 16) + jp * 32;                   // uniform
                const float* lb = lbias + col0 + (lane >> 4) * 8;                 // eight consecutive columns of this lane
                const f32x4 lo = acc[i][2 * jp] + *reinterpret_cast<const f32x4*>(lb);
                const f32x4 hi = acc[i][2 * jp + 1] + *reinterpret_cast<const f32x4*>(lb + 4);
                char* piece = static_cast<char*>(Cv) + ((int64_t)row0 * N + col0) * ESZ;   // scalar base, 32-bit lane offset
                // DBG 16 (timing experiment): the stores stay in the program, so the MFMAs are not dead code, but never execute
                if (row0 + (lane & 15) < M && !(DBG & 4) && (!(DBG & 16) || lo[0] == 12345.678f)) {
                    if (OUT_BF16) {
                        uint4 pk;
                        pk.x = pack_bf16x2(lo[0], lo[1]);
                        pk.y = pack_bf16x2(lo[2], lo[3]);
                        pk.z = pack_bf16x2(hi[0], hi[1]);
                        pk.w = pack_bf16x2(hi[2], hi[3]);
                        *reinterpret_cast<uint4*>(piece + coff) = pk;
                    } else {
                        *reinterpret_cast<f32x4*>(piece + coff) = lo;
                        *reinterpret_cast<f32x4*>(piece + coff + 16) = hi;
                    }
                }
            }
        }
    };
    constexpr int STORES = OUT_BF16 ? TI * TJ / 2 : TI * TJ;   // epilogue store instructions per wave

    // Waves w and w + 4 share a SIMD.  If all eight waves read their fragments and then multiply, the matrix pipe idles
    // while the LDS serves 96 KiB of reads after every barrier, and the LDS idles while everybody multiplies.  So the two
    // halves of the workgroup run half a step apart: after the barrier of step hs the EARLY waves (0-3) read the fragments
    // of half stage hs and then multiply them; the LATE waves (4-7) first multiply the fragments they read during step
    // hs - 1 and then read those of hs.  One fragment set per wave; the ring slot of hs is still released by the barrier
    // of step hs + 1.  The late waves finish a tile one step later: their epilogue sits inside step 0 of the next tile.
    auto run = [&](auto late_c) __attribute__((always_inline)) {
        constexpr bool LATE = decltype(late_c)::value;
        bf16x8 a[TI], b[TJ];
        if (DBG & 8) {
#pragma unroll
            for (int i = 0; i < TI; i++) a[i] = bf16x8{(short)lane, 1, 2, 3, 4, 5, 6, 7};
#pragma unroll
            for (int j = 0; j < TJ; j++) b[j] = bf16x8{(short)wave, 1, 2, 3, 4, 5, 6, 7};
        }
        auto read_frags = [&](int hs) __attribute__((always_inline)) {
            if (DBG & 8) return;                             // timing experiment: no LDS fragment reads
            const char* cur = lds + (hs % kStages) * T::STAGE;
            const int chunk = lane >> 4, r16 = lane & 15;
#pragma unroll
            for (int j = 0; j < TJ; j++) b[j] = frag(cur + T::HALF_A, b_row(j, r16), chunk);
#pragma unroll
            for (int i = 0; i < TI; i++) a[i] = frag(cur, (wm * TI + i) * 16 + r16, chunk);
        };
        auto multiply = [&]() __attribute__((always_inline)) {
            if (DBG & 2) {                                   // timing experiment: no MFMA (keep the fragments alive)
#pragma unroll
                for (int i = 0; i < TI; i++) asm volatile("" ::"v"(a[i]));
#pragma unroll
                for (int j = 0; j < TJ; j++) asm volatile("" ::"v"(b[j]));
                return;
            }
#pragma unroll
            for (int i = 0; i < TI; i++)
#pragma unroll
                for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        };
        // sixteen-wave shape without the half-step skew: four waves per SIMD hide each other's LDS latency, so a wave reads its A fragments
        // one MFMA row ahead instead of all at once (128 registers per wave: the accumulators take 64 of them)
        auto read_multiply = [&](int hs) __attribute__((always_inline)) {
            if (DBG & 10) { read_frags(hs); multiply(); return; }      // timing experiments go through the separately switchable halves
            const char* cur = lds + (hs % kStages) * T::STAGE;
            const int chunk = lane >> 4, r16 = lane & 15;
#pragma unroll
            for (int j = 0; j < TJ; j++) b[j] = frag(cur + T::HALF_A, b_row(j, r16), chunk);
            bf16x8 an = frag(cur, (wm * TI) * 16 + r16, chunk);
#pragma unroll
            for (int i = 0; i < TI; i++) {
                const bf16x8 ac = an;
                if (i + 1 < TI) an = frag(cur, (wm * TI + i + 1) * 16 + r16, chunk);
#pragma unroll
                for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], ac, acc[i][j], 0, 0, 0);
            }
        };
        int tile = lw;
        int m0 = (tile / tiles_n) * T::BM, n0 = (tile % tiles_n) * T::BN;
        int pm0 = m0, pn0 = n0;                              // late waves: the tile whose last fragments are still to be multiplied
        // prologue of the first tile only: three half stages in flight
#pragma unroll
        for (int hs = 0; hs < kStages - 1; hs++) issue(m0, n0, hs, hs);
        zero_acc();
        bool first = true;
        for (;;) {
            const int nxt = tile + gridDim.x;
            const bool more = nxt < total;
            const int m1 = more ? (nxt / tiles_n) * T::BM : m0, n1 = more ? (nxt % tiles_n) * T::BN : n0;
            // Step hs.  A half stage's ring slot is hs mod 4; NH is a multiple of 4, so the next tile's half stage 0 lands in
            // slot 0 again and the stream of half stages runs on across tiles.
#pragma unroll
            for (int hs = 0; hs < NH; hs++) {
                // half stage hs must have landed.  Younger operations of this wave: the two half stages behind it (2 P
                // LDS-DMAs), plus the stores of the previous tile's epilogue while the half stages waited for were issued
                // BEFORE those stores: steps 0-2 for the early waves (epilogue at the end of step NH - 1), steps 1-2 for
                // the late ones (epilogue inside step 0, before that step's refill).  At the end of the stream there are fewer.
                __builtin_amdgcn_sched_barrier(0);
                if (!first && hs >= (LATE ? 1 : 0) && hs < 3) AVD_WAIT_VM(2 * T::P + STORES);
                else if (hs == NH - 2 && !more) AVD_WAIT_VM(T::P);
                else if (hs == NH - 1 && !more) AVD_WAIT_VM(0);
                else AVD_WAIT_VM(2 * T::P);
                // ... for every wave; the same barrier says everyone has finished reading half stage hs - 1, whose slot
                // is refilled with stream position hs + 3 (the late waves' reads of hs - 1 are their last instructions
                // before this point: they must have returned, not merely been issued)
                if (LATE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);           // nothing (LDS reads, MFMAs) may be scheduled across the wait + barrier
                // refill of the slot of half stage hs - 1 with stream position hs + 3.  A wave can sit at an LDS-DMA
                // instruction for hundreds of cycles when the L1's miss queue is full, and it issues in order: the early
                // waves issue theirs first thing (their partner on the SIMD is multiplying), the late waves after their
                // MFMAs (while the early waves multiply)
                auto refill = [&]() __attribute__((always_inline)) {
                    if (hs + kStages - 1 < NH) issue(m0, n0, hs + kStages - 1, (hs + kStages - 1) % kStages);
                    else if (more) issue(m1, n1, hs + kStages - 1 - NH, (hs + kStages - 1) % kStages);
                };
                if (!LATE && !T::SKEW) {
                    refill();
                    read_multiply(hs);
                } else if (!LATE) {
                    refill();
                    read_frags(hs);
                    __builtin_amdgcn_sched_barrier(0);      // all reads are issued before the first MFMA
                    multiply();
                } else {
                    if (hs > 0) {
                        multiply();                          // the fragments of hs - 1
                    } else if (!first) {
                        multiply();                          // the previous tile's last half stage, then its epilogue
                        store_tile(pm0, pn0);
                        zero_acc();
                    }
                    __builtin_amdgcn_sched_barrier(0);      // the MFMAs are issued before the reads that overwrite their operands
                    refill();
                    read_frags(hs);
                }
            }
            if (!LATE) {
                store_tile(m0, n0);
                zero_acc();
            }
            pm0 = m0; pn0 = n0;
            if (!more) break;
            tile = nxt; m0 = m1; n0 = n1;
            first = false;
        }
        if (LATE) {
            multiply();
            store_tile(pm0, pn0);
        }
    };
    if (T::SKEW && ((wave >> 2) & 1)) run(std::true_type{});              // waves w, w + 4 (, w + 8, w + 12) share a SIMD: half of each SIMD's waves run late
    else run(std::false_type{});
}

}  // namespace

// row-major [rows][K] bf16 -> the blocked operand layout (host side, once per weight upload); rows % 16 == 0, K % 32 == 0
void gemm_block_operand(const uint16_t* src, uint16_t* dst, int rows, int K)
{
    for (int r = 0; r < rows; r++)
        for (int k = 0; k < K; k++) dst[blocked_index(r, k, K)] = src[(size_t)r * K + k];
}

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
    if (N != kDim || K != kDim) { ctx->err = "gemm_bf16_nt: built for N = K = 768 (the ViT-B/16 patch embedding)"; return AVD_ERR_ARG; }
    auto go = [&](auto kern, int threads, int grid, size_t lds, auto... args) -> int {
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, ctx->stream, args...);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    };
    {
        // persistent kernel, one workgroup per CU (LDS: ring + bias), grid a multiple of 8 so that the XCD-aware order is
        // a bijection.  256 x 256 tiles: a 384 x 192 shape (one round of tiles for a 120-frame clip instead of two) was
        // measured slower at every size (144 accumulator registers + fragments spill): 0.0478 vs 0.0454 ms at 120 frames.
        using Sq = TileShape<2, 8, 4>;
        const int tiles = ((M + Sq::BM - 1) / Sq::BM) * (N / Sq::BN);
        int grid_sq = ctx->num_cus / 8 * 8;
        if (grid_sq < 8) grid_sq = 8;
        if (grid_sq > (tiles + 7) / 8 * 8) grid_sq = (tiles + 7) / 8 * 8;
        const size_t lds = (size_t)Sq::LDS + (size_t)N * sizeof(float);
#ifdef AVD_GEMM_DEBUG
        // timing experiments (tools/gemm_dbg.sh, tools/r05_gemm_dbg.sh): AVD_GEMM_DBG bits 1 = no LDS-DMA, 2 = no MFMA, 4 = no stores, 8 = no fragment reads
        static const int dbg = [] { const char* e = std::getenv("AVD_GEMM_DBG"); return e ? std::atoi(e) : 0; }();
        using Dbg16 = TileShape<4, 4, 4, 16, false>;
        switch (dbg) {
#define AVD_DBG_CASE(D) case D: if (ctx->gemm_waves == 16) return go(k_gemm_bf16_nt_persistent<kDim, 1, Dbg16, D>, 1024, grid_sq, lds, d_a, d_bt, d_bias, d_c, M, N); \
                                return go(k_gemm_bf16_nt_persistent<kDim, 1, Sq, D>, 512, grid_sq, lds, d_a, d_bt, d_bias, d_c, M, N);
            AVD_DBG_CASE(1) AVD_DBG_CASE(2) AVD_DBG_CASE(4) AVD_DBG_CASE(8) AVD_DBG_CASE(3) AVD_DBG_CASE(5) AVD_DBG_CASE(6) AVD_DBG_CASE(7)
            AVD_DBG_CASE(10) AVD_DBG_CASE(12) AVD_DBG_CASE(14) AVD_DBG_CASE(16) AVD_DBG_CASE(17) AVD_DBG_CASE(24)
#undef AVD_DBG_CASE
        default: break;
        }
#endif
        // ctx->gemm_waves (option "gemm_waves", AVD_GEMM_WAVES): 8 (default) = eight waves of 8 x 4 MFMA tiles, half of them half a step late;
        // 16 = the same tile on sixteen waves of 4 x 4 tiles, no skew (round 5: an operand stream that sixteen waves issue runs at 92-99 instead of
        // 60-72 GB/s per CU in isolation, tools/ldsdma_occ_bench.hip -- measured no faster in the kernel, 0.268-0.276 against 0.252-0.267 ms:
        // loads, MFMAs and stores of this kernel add up rather than overlap whichever waves issue them, profiles/r05_experiments.md section 3)
        using Sq16 = TileShape<4, 4, 4, 16, false>;
        static_assert(Sq16::BM == Sq::BM && Sq16::BN == Sq::BN && Sq16::LDS == Sq::LDS, "the same tile, ring and grid");
        if (ctx->gemm_waves == 16) {
            if (out_bf16) return go(k_gemm_bf16_nt_persistent<kDim, 1, Sq16>, 1024, grid_sq, lds, d_a, d_bt, d_bias, d_c, M, N);
            return go(k_gemm_bf16_nt_persistent<kDim, 0, Sq16>, 1024, grid_sq, lds, d_a, d_bt, d_bias, d_c, M, N);
        }
        if (out_bf16) return go(k_gemm_bf16_nt_persistent<kDim, 1, Sq>, 512, grid_sq, lds, d_a, d_bt, d_bias, d_c, M, N);
        return go(k_gemm_bf16_nt_persistent<kDim, 0, Sq>, 512, grid_sq, lds, d_a, d_bt, d_bias, d_c, M, N);
    }
}
