// avd_norm.hip -- LayerNorm and softmax (gfx950): the two non-GEMM elements of the classifier stack BASELINE.json's
// north_star names ("conv / GEMM / LayerNorm / softmax").  BUILD-DEFINED EXTENSIONS like avd_vit.hip / avd_cnn.hip: the
// reference has no learned model (its per-frame "model" is the closed form of app/analyzers/video.py:54-56), so there is no
// reference site; the oracle is float32 torch on the CPU (tests/test_norm.py).  Never part of ai_score.
//
// Both are one pass over HBM with a wave-level reduction: a wave owns a row, keeps it in registers (LayerNorm: 768 values =
// 12 per lane; softmax: 1000 logits = 16 per lane), reduces with DPP / ds_swizzle shuffles (__shfl_xor over 64 lanes), and
// writes the result -- the row is read once and written once, nothing is staged in LDS.  Bound: HBM.
#include "avd_internal.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float bf16_to_f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short f_to_bf16(float f)       // round to nearest even; NaN stays NaN
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// y = (x - mean) / sqrt(var + eps) * gamma + beta over rows of COLS = 256 * V values (V 16-byte / 8-byte pieces per lane).
// One wave per row, four rows per workgroup.  Statistics as torch.nn.functional.layer_norm: biased variance about the mean,
// float32 (two passes over the registers: mean first, then the squared deviations).
template <int V, bool BF16>
__global__ __launch_bounds__(256) void k_layernorm(const void* __restrict__ xin, void* __restrict__ yout, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float eps, long long rows)
{
    constexpr int COLS = 256 * V;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                                // whole wave
    float v[V][4];
    if (BF16) {
        const uint2* src = reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(xin) + row * COLS);
#pragma unroll
        for (int i = 0; i < V; i++) {
            const uint2 w = src[i * 64 + lane];
            v[i][0] = bf16_to_f((unsigned short)(w.x & 0xffff)); v[i][1] = bf16_to_f((unsigned short)(w.x >> 16));
            v[i][2] = bf16_to_f((unsigned short)(w.y & 0xffff)); v[i][3] = bf16_to_f((unsigned short)(w.y >> 16));
        }
    } else {
        const float4* src = reinterpret_cast<const float4*>(static_cast<const float*>(xin) + row * COLS);
#pragma unroll
        for (int i = 0; i < V; i++) {
            const float4 w = src[i * 64 + lane];
            v[i][0] = w.x; v[i][1] = w.y; v[i][2] = w.z; v[i][3] = w.w;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; i++) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum_f(s) * (1.f / COLS);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { const float d = v[i][j] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum_f(q) * (1.f / COLS) + eps);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < V; i++) {
        const float4 g = g4[i * 64 + lane], b = b4[i * 64 + lane];
        const float o0 = (v[i][0] - mean) * rstd * g.x + b.x, o1 = (v[i][1] - mean) * rstd * g.y + b.y;
        const float o2 = (v[i][2] - mean) * rstd * g.z + b.z, o3 = (v[i][3] - mean) * rstd * g.w + b.w;
        if (BF16) {
            uint2 w;
            w.x = (unsigned)f_to_bf16(o0) | ((unsigned)f_to_bf16(o1) << 16);
            w.y = (unsigned)f_to_bf16(o2) | ((unsigned)f_to_bf16(o3) << 16);
            reinterpret_cast<uint2*>(static_cast<unsigned short*>(yout) + row * COLS)[i * 64 + lane] = w;
        } else {
            reinterpret_cast<float4*>(static_cast<float*>(yout) + row * COLS)[i * 64 + lane] = make_float4(o0, o1, o2, o3);
        }
    }
}

// softmax over rows of `cols` float32 logits (cols <= 64 * 4 * P, cols % 4 == 0): y = exp(x - max) / sum.  One wave per row.
template <int P>
__global__ __launch_bounds__(256) void k_softmax(const float* __restrict__ x, float* __restrict__ y, long long rows, int cols)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* src = reinterpret_cast<const float4*>(x + row * cols);
    const int n4 = cols >> 2;
    float v[P][4];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int k = i * 64 + lane;
        float4 w = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (k < n4) w = src[k];
        v[i][0] = w.x; v[i][1] = w.y; v[i][2] = w.z; v[i][3] = w.w;
        m = fmaxf(m, fmaxf(fmaxf(w.x, w.y), fmaxf(w.z, w.w)));
    }
    m = wave_max_f(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { v[i][j] = expf(v[i][j] - m); s += v[i][j]; }      // exp(-inf) = 0 for the padding
    const float inv = 1.f / wave_sum_f(s);
    float4* dst = reinterpret_cast<float4*>(y + row * cols);
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int k = i * 64 + lane;
        if (k < n4) dst[k] = make_float4(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv);
    }
}

// ANY row length (round 5: the fast kernels above cover the shapes of the classifier stack; every other shape used to be refused): one wave per
// row, lanes stride over the columns, the row is read from memory once per pass (sum; squared deviations; output -- L2 serves the re-reads).
// The same float32 statistics in the same order per lane would not be bit-identical to the fast kernels (another grouping of the sums); both are
// checked against float32 torch (tests/test_norm.py).
template <bool BF16>
__global__ __launch_bounds__(256) void k_layernorm_any(const void* __restrict__ xin, void* __restrict__ yout, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, long long rows, int cols)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    auto at = [&](int c) -> float {
        return BF16 ? bf16_to_f(static_cast<const unsigned short*>(xin)[row * cols + c]) : static_cast<const float*>(xin)[row * cols + c];
    };
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += at(c);
    const float mean = wave_sum_f(s) / (float)cols;
    float q = 0.f;
    for (int c = lane; c < cols; c += 64) { const float d = at(c) - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum_f(q) / (float)cols + eps);
    for (int c = lane; c < cols; c += 64) {
        const float o = (at(c) - mean) * rstd * gamma[c] + beta[c];
        if (BF16) static_cast<unsigned short*>(yout)[row * cols + c] = f_to_bf16(o);
        else static_cast<float*>(yout)[row * cols + c] = o;
    }
}

__global__ __launch_bounds__(256) void k_softmax_any(const float* __restrict__ x, float* __restrict__ y, long long rows, int cols)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* src = x + row * cols;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, src[c]);
    m = wave_max_f(m);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += expf(src[c] - m);
    const float inv = 1.f / wave_sum_f(s);
    for (int c = lane; c < cols; c += 64) y[row * cols + c] = expf(src[c] - m) * inv;
}

}  // namespace

// device pointers; x / y float32 or bf16 [rows][cols], gamma / beta float32 [cols]; cols = 256, 512, 768, 1024, 2048 run the register-resident
// kernel, any other cols >= 1 the general one
int launch_layernorm(avd_ctx* ctx, const void* d_x, void* d_y, int bf16, long long rows, int cols, const float* d_gamma, const float* d_beta, float eps)
{
    if (rows <= 0) return 0;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define AVD_LN_CASE(V)                                                                                                       \
    do {                                                                                                                     \
        if (bf16) hipLaunchKernelGGL((k_layernorm<V, true>), grid, block, 0, ctx->stream, d_x, d_y, d_gamma, d_beta, eps, rows); \
        else hipLaunchKernelGGL((k_layernorm<V, false>), grid, block, 0, ctx->stream, d_x, d_y, d_gamma, d_beta, eps, rows);    \
    } while (0)
    switch (cols) {
    case 256: AVD_LN_CASE(1); break;
    case 512: AVD_LN_CASE(2); break;
    case 768: AVD_LN_CASE(3); break;
    case 1024: AVD_LN_CASE(4); break;
    case 2048: AVD_LN_CASE(8); break;
    default:
        if (cols < 1) { ctx->err = "avd_layernorm: cols must be positive"; return AVD_ERR_ARG; }
        if (bf16) hipLaunchKernelGGL(k_layernorm_any<true>, grid, block, 0, ctx->stream, d_x, d_y, d_gamma, d_beta, eps, rows, cols);
        else hipLaunchKernelGGL(k_layernorm_any<false>, grid, block, 0, ctx->stream, d_x, d_y, d_gamma, d_beta, eps, rows, cols);
    }
#undef AVD_LN_CASE
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int launch_softmax(avd_ctx* ctx, const float* d_x, float* d_y, long long rows, int cols)
{
    if (rows <= 0) return 0;
    if (cols < 1) { ctx->err = "avd_softmax: cols must be positive"; return AVD_ERR_ARG; }
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int p = (cols / 4 + 63) / 64;
    // rows of whole float4s up to 4096 logits stay in registers; any other length goes through the general kernel
    if ((cols & 3) || cols > 4096) hipLaunchKernelGGL(k_softmax_any, grid, block, 0, ctx->stream, d_x, d_y, rows, cols);
    else if (p <= 4) hipLaunchKernelGGL(k_softmax<4>, grid, block, 0, ctx->stream, d_x, d_y, rows, cols);
    else if (p <= 8) hipLaunchKernelGGL(k_softmax<8>, grid, block, 0, ctx->stream, d_x, d_y, rows, cols);
    else hipLaunchKernelGGL(k_softmax<16>, grid, block, 0, ctx->stream, d_x, d_y, rows, cols);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
