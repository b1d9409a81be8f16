// avd_mfma_device.h -- device helpers shared by the matrix-core kernels (avd_vit.hip, avd_cnn.hip): bf16 conversion, the
// blocked + swizzled operand layout and the LDS fragment read that goes with it.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace avd_mfma {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BKH = 32;                                     // a staging unit ("half stage") is 32 deep in K
constexpr int kStages = 4;                                  // ring: kStages - 1 half stages in flight beside the one being read
                                                            // (5 = all 160 KiB of LDS measured no faster than 4)

__device__ __forceinline__ uint16_t f32_to_bf16(float v)
{
    const unsigned u = __float_as_uint(v);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);       // round to nearest even (inputs are finite)
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
// two floats -> two bf16 in one dword (lo in bits 0-15): gfx950's v_cvt_pk_bf16_f32, round to nearest even -- the same bits as
// f32_to_bf16 for finite inputs, one instruction instead of eight
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi)
{
    const bf16x2_t r = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return __builtin_bit_cast(unsigned, r);
}

// Bank swizzle of a half tile (64-byte rows, four 16-byte chunks per row, four rows per 256-byte bank row): chunk c of
// row r sits in slot c ^ swz((r >> 2) & 3).  A ds_read_b128 is served in groups of 16 lanes that are NOT contiguous
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...): with lane = (chunk << 4) | row a group reads rows {0-3, 12-15} of
// one chunk and rows {4-11} of the next, and swz = {0, 2, 3, 1} is what makes those sixteen accesses hit sixteen
// different 16-byte bank groups (the plain XOR with (r >> 2) & 3 is two-way conflicted for these groups).
__host__ __device__ __forceinline__ int swz(int k) { return (0x78 >> (2 * k)) & 3; }      // {0, 2, 3, 1}

// OPERAND LAYOUT IN HBM ("blocked"): a K-contiguous matrix [rows][K] is stored as 1-KiB blocks of 16 rows x 32 k, block
// (row / 16, k / 32) at ((row / 16) * (K / 32) + k / 32) * 1 KiB, and INSIDE a block exactly the bytes of the LDS image
// the MFMA fragments are read from: row r at r * 64, its four 16-byte chunks swizzled as above.  One
// global_load_lds_dwordx4 wave-instruction then copies ONE contiguous KiB (eight whole cache lines) straight into LDS.
// Element index of (row, k):
__host__ __device__ __forceinline__ int64_t blocked_index(int row, int k, int K)
{
    const int r = row & 15, kk = k & 31;
    return ((int64_t)(row >> 4) * (K >> 5) + (k >> 5)) * 512 + r * 32 + (((kk >> 3) ^ swz((r >> 2) & 3)) << 3) + (kk & 7);
}

// MFMA fragment (16 rows x 8 k of one 16-byte chunk) of an LDS half tile in that layout
__device__ __forceinline__ bf16x8 frag(const char* lds_tile, int row, int chunk)
{
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 64 + ((chunk ^ swz((row >> 2) & 3)) << 4));
}

}  // namespace avd_mfma
