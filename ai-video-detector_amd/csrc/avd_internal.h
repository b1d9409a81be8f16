// avd_internal.h -- shared declarations of libavd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/avd.h"

#define AVD_FB_LEVELS 4            // pyramid scales 1/8,1/4,1/2,1 of 320 (see farneback.hip)
#define AVD_NPIX (AVD_SMALL * AVD_SMALL)

#define HIP_TRY(ctx, expr)                                                         \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) {                                                   \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);       \
            return AVD_ERR_DEVICE;                                                 \
        }                                                                          \
    } while (0)

// ---- host-built resampling tables (avd_tables.cpp) ------------------------------
// INTER_LINEAR uint8 -> 320x320 (cv2.resize default, reference video.py:43)
struct LinearTab {
    std::vector<int> x0, x1, y0, y1;          // clipped source indices
    std::vector<short> a0, a1, b0, b1;        // 11-bit fixed-point weights
};
void build_linear_tab(int src_h, int src_w, int dst_h, int dst_w, LinearTab& t);

// INTER_AREA uint8 -> 32x32 (reference video.py:6): per destination index a run of
// consecutive source indices with (first, middle, last) float weights.
struct AreaAxis {
    std::vector<int> begin, count;
    std::vector<float> w_first, w_mid, w_last;
};
struct AreaTab {
    AreaAxis x, y;
    int fast;          // both scales integral: integer box sums (ResizeAreaFast)
    int iscale_x, iscale_y;
};
int build_area_tab(int src_h, int src_w, int dst_h, int dst_w, AreaTab& t);   // <0: unsupported

// Farneback constant tables (polynomial expansion kernels, pyramid Gaussian kernels)
struct FbConsts {
    float g[11], xg[11], xxg[11];      // centre at index 5 (poly_n = 5)
    double ig11, ig03, ig33, ig55;
    float gk[AVD_FB_LEVELS][19];       // per-level Gaussian taps, level index = pyramid k
    int gksize[AVD_FB_LEVELS];
};
void build_fb_consts(FbConsts& c);

// libswscale's C yuv420 -> BGR24 conversion in integer form (avd_tables.cpp, yuv2rgb.c semantics, BT.601 limited):
//   value = clip8((c0 + (Y + off) * cy) >> 16),  off_r = ((V*crv)>>16) - (crv>>9),  off_b likewise with U and cbu,
//   off_g = ((U*cgu)>>16) - (cgu>>9) + ((V*cgv)>>16) - (cgv>>9)   (arithmetic shifts: cgu, cgv are negative)
struct YuvConsts { int cy, crv, cbu, cgu, cgv, c0, kr, kb, kg; };   // kr = -(crv>>9), kb = -(cbu>>9), kg = -(cgu>>9) - (cgv>>9)
void build_yuv_consts(YuvConsts& c);

struct Nv12Params {
    const uint8_t* uv;                 // interleaved U,V plane of frame 0 (the Y plane is the kernel's frame pointer)
    int64_t uv_row_stride, uv_frame_stride;
    YuvConsts k;
};

// ---- device-side parameter blocks ------------------------------------------------
struct LinTap { short i0, i1, w0, w1; };   // two source indices + 11-bit weights of one output row/column

struct PreParams {
    // linear 320: packed per-column and per-row taps
    const LinTap *lxt, *lyt;           // [320] each
    const int* band_dy;                // [nbands+1] first dy owned by each band
    // area x axis (32 entries)
    const int *ax_begin, *ax_count;
    const float *ax_first, *ax_mid, *ax_last;
    int area_fast;
    int area_x_uniform4;               // every x cell: begin%4==0, count%4==0, single weight
    int h, w, rows_per_band, nbands, pitch;
    int64_t row_stride, frame_stride;
    int dbg_skip;                      // timing experiments only (AVD_DBG_SKIP): 1 lap, 2 area, 4 linear, 8 gray math, 16 loads
};

struct HashParams {
    const int *ay_begin, *ay_count;
    const float *ay_first, *ay_mid, *ay_last;
    int area_fast, fast_area, fast_simd_w;   // fast: integer sums; area = iscale_x*iscale_y
    int h;
};

// Tables and band layout of one frame geometry.  A context keeps the last kGeomCache geometries (a mixed-resolution
// stream alternates between a few sizes): switching between cached geometries allocates and frees nothing.
constexpr int kGeomCache = 4;
struct Geom {
    int h = 0, w = 0;
    void* d_tables = nullptr; size_t tables_bytes = 0;
    PreParams pre{};
    HashParams hsh{};
    unsigned long long stamp = 0;      // last use (LRU eviction)
};

struct Workspace {
    // geometry of the clip being enqueued (a copy of its cache entry) and the capacities of the per-frame buffers, which
    // only ever grow: cap_n frames, rowbuf_cap floats, lappart_cap slots
    int cap_n = 0, h = 0, w = 0;
    size_t rowbuf_cap = 0, lappart_cap = 0;
    Geom geoms[kGeomCache];
    unsigned long long geom_clock = 0;
    // position of the clip being enqueued inside the buffers of the call (a batch concatenates its clips)
    int f0 = 0; size_t rowbuf_off = 0, lappart_off = 0;
    int fb_cap = 0;                    // pairs the Farneback scratch holds
    int* d_clipstart = nullptr; int* h_clipstart = nullptr; int clipstart_cap = 0;   // [n] 1 = first frame of a clip
    // preprocess
    uint8_t* d_stage = nullptr; size_t stage_bytes = 0;   // staged host frames
    uint8_t* d_small = nullptr;       // [n][320*320]
    float* d_rowbuf = nullptr;        // [n][h][32]
    uint8_t* d_area = nullptr;        // [n][1024]
    uint8_t* d_hash = nullptr;        // [n][1024]
    int* d_ham = nullptr;             // [n]
    unsigned long long* d_lap = nullptr;   // [n][2]
    long long* d_lap_part = nullptr;       // [n][nbands][8 waves][2] per-wave partial moments
    int lap_waves = 4;
    PreParams pre{};
    HashParams hsh{};
    // farneback
    float* d_pyr[AVD_FB_LEVELS] = {};     // [n][hL*wL]
    float* d_poly[AVD_FB_LEVELS] = {};    // [n][hL*wL][5] interleaved polynomial coefficients
    float* d_flow[AVD_FB_LEVELS] = {};    // [n-1][2][hL*wL]  planar
    float* d_flow2[AVD_FB_LEVELS] = {};   // second flow buffer of a level: the fast level kernel (avd_fbfast.hip) ping-pongs
    const float* flow_res[AVD_FB_LEVELS] = {};   // where the last call left the final flow of each level (d_flow or d_flow2)
    float* d_mag = nullptr;               // [n-1][320*320] |flow| of the full-resolution level (written by the fast level kernel, or by k_mag in exact mode)
    int mag_valid = 0;                    // d_mag holds the magnitudes of the chunk being processed
    int* d_fbflags = nullptr;             // [n-1] ill-posedness flags of the fast level kernels (bit k: level k met the solver's criterion, bit 4 + k: the border-sign criterion); such pairs are re-run exactly
    int* d_pairdiff = nullptr;            // [n-1][20] "frame p differs from frame p + 1" per tile of the pyramid kernel's 160-px scale (all zero: bit-identical frames)
    int* d_rlist = nullptr; int rlist_cap = 0;                  // exact re-run: the flagged pairs of a chunk, compacted by the host
    int* h_rlist = nullptr;                                     // pinned staging of that list
    double* d_vs_rerun = nullptr; double* d_vs0_rerun = nullptr; // the two-kernel path's double intermediate for kRerunTwoKernelMax pairs (allocated by the first re-run)
    double* d_vs = nullptr;               // [n-1] x 64x16 tiles of D = vsum(x+7)-vsum(x-8), double
    double* d_vs0 = nullptr;              // [n-1][5][320][8]  vsum columns 0..6 (row init)
    float* d_flow_il = nullptr;           // [n-1][320*320][2] interleaved (cv2 layout)
    float* d_stats = nullptr;             // [n-1][2] mean, var
    avd_frame_record* d_rec = nullptr;    // [n]
    avd_frame_record* h_rec = nullptr;    // [n] pinned landing buffer of the asynchronous copy-out
    // ViT patch-embed extension (avd_vit.hip): weights [768][768] bf16 + bias, im2col patches, token staging
    uint16_t* d_vit_w = nullptr; float* d_vit_bias = nullptr; int vit_has_bias = 0;
    uint16_t* d_vit_patches = nullptr; size_t vit_patch_elems = 0;
    float* d_vit_tokens = nullptr; size_t vit_token_elems = 0;
    // audio analyzer (avd_audio.hip): tables (hanning, twiddles) for the current window lengths, scratch, records
    double* d_audio_tab = nullptr; size_t audio_tab_elems = 0; int audio_win = 0, audio_last = 0;
    double* d_audio_buf = nullptr; size_t audio_buf_elems = 0;
    avd_audio_window* d_audio_out = nullptr; size_t audio_out_elems = 0;
    // CNN extension (avd_cnn.hip): blocked conv weights + linear layer, biases, activation scratch for cnn_frames frames
    uint16_t* d_cnn_w = nullptr; float* d_cnn_b = nullptr;
    std::vector<size_t> cnn_w_off; size_t cnn_fc_off = 0;
    uint16_t* d_cnn_act[4] = {}; uint16_t* d_cnn_img = nullptr;
    float* d_cnn_pool = nullptr; float* d_cnn_logits = nullptr; int cnn_frames = 0;
};

struct avd_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_in = nullptr;                // avd_wait_stream: recorded on the caller's stream, waited for by ours
    avd_frame_record* pending_out = nullptr;   // caller buffer the pinned records are handed to in avd_synchronize
    int pending_n = 0;
    hipEvent_t stage_ev[5] = {};
    hipEvent_t kern_ev[12] = {};           // profiling: start/stop of the 3 k_uv<320> and 3 k_hscan<320> launches of a segment
    int kern_ev_used = 0;
    int profiling = 0;
    int stage_marks = 0;                       // stage events recorded by the call in flight (5 = all of them)
    // per-kernel profiling (avd_kernel_ms): an event in front of every kernel (group) of the path, labelled with the avd_kernel_id of the
    // region that starts there; elapsed times between consecutive events are summed per id when the call is drained
    hipEvent_t kmark_ev[96] = {};
    int kmark_id[96] = {};
    int kmark_used = 0;
    int kmark_incomplete = 0;                  // the same for the call whose times kernel_ms holds
    int kmark_overflow = 0;                    // a call recorded more regions than kmark_ev holds (reset when the marks are)
    float kernel_ms[AVD_K_COUNT] = {};
    float stage_ms[6] = {};
    std::string err;
    Workspace ws;
    FbConsts fbc;
    void* d_fbc = nullptr;          // FbConsts on device
    int last_n = 0;
    int rec_n = 0;                   // records the last avd_analyze_* call left in ws.d_rec (0 after any other entry point: avd_allgather_last_records checks it)
    void* comm = nullptr;            // RCCL communicator (avd_comm.cpp), bound at run time
    int comm_rank = 0, comm_world = 1;
    void* d_comm = nullptr; size_t comm_bytes = 0;     // device staging of the record exchange
    int cnn_tiles = 0;              // convolution tiling of the CNN extension: 0 = heuristic, 1 = 256-pixel tiles, 2 = 128 x 128 wherever possible
    int fb_fold_blur = 1;           // the 320-px scale's 3 x 3 pyramid blur formed inside the polynomial expansion (no effect on results); AVD_FB_FOLD_BLUR / avd_set_option
    int fb_wide160 = 2;             // fast mode: the 160-px level as one three-block strip per pair (1: fewer CU-microseconds, throughput) or as two strips (0: shorter launches, latency);
                                    // 2 (default) = by what is in flight when the call is enqueued: one strip if another context of the process holds an undrained call, two if this clip is alone; AVD_FB_WIDE160 / avd_set_option
    int counted_in_flight = 0;      // this context's enqueued call is counted in avd_calls_in_flight()
    int fb_wide160_used = 0;        // the shape the last call's 160-px launches took (read-only option "fb_wide160_used")
    int gemm_waves = 8;             // patch-embed GEMM: waves per workgroup (8: 8 x 4 MFMA tiles per wave, 16: 4 x 4; measured no faster), the same 256 x 256 tile; AVD_GEMM_WAVES / avd_set_option
    int cnn_chunk = 128;            // CNN extension: frames per forward pass (activation scratch = 4 x 1.6 MB per frame)
    int cnn_fuse = 2;               // CNN extension: a block's 3x3 and expanding 1x1 in one launch (stages 1, 2): 2 = with the 3x3's input slab in LDS in the stride-1 blocks (k_slab3_expand), 1 = gathering kernels only, 0 = layer by layer
    int fb_fused = 0xF;             // bit k: pyramid level k runs the fused kernel (avd_fbfused.hip) instead of k_uv/k_uvp + k_hscan
    int fb_fold_up = 5;             // fast mode, bit mask (no effect on results; AVD_FB_FOLD_UP / avd_set_option): 1 the 320-px level's first launch resizes the
                                    // 160-px flow itself (no k_flow_up<320>), 2 the 160- / 80-px levels do so in a prologue, 4 the 80- / 40-px levels run their three
                                    // iterations in one launch
    int fb_mode = 1;                // 1 = fast level kernel (avd_fbfast.hip: literal vertical chain, direct horizontal window sums; flow within
                                    // 1e-5 px of the oracle, in practice identical), 0 = exact (avd_fbfused.hip / two-kernel path: bit-identical)
    int fb_rerun = 1;               // fast mode: pairs the level kernels flag as ill-posed are re-run by the exact kernels (launch_farneback_rerun); 0 = A/B, tests
    int fb_rerun_fused = 0xC;       // exact re-run of FEW pairs (<= kRerunTwoKernelMax): level mask of the fused kernel (bit 3 = 40 px must be set), the other levels run the two-kernel path
    int last_rerun = 0;             // pairs re-run by the last drained call
    // the last Farneback chunk of an asynchronous call, whose flags the host has not seen yet (impl_synchronize re-runs its flagged pairs)
    struct { int active = 0, p0 = 0, np = 0, fa = 0, n = 0; const int* clipstart = nullptr; } tail;
    // A thread that waits in avd_synchronize settles the tails of OTHER contexts whose fast pass has finished meanwhile (avd_capi.hip, tail_help_others):
    // one host thread driving several contexts (avd_hip.ClipsInFlight, bench.py) would otherwise start each clip's re-run only when it reaches that clip.
    std::recursive_mutex api_mu;    // held by every entry point that takes this context; helpers only try_lock it
    hipEvent_t tail_ev = nullptr;   // the call's records (with the flag words) have reached the pinned buffer
    int tail_registered = 0;        // in the process-wide list of contexts with an unsettled tail
    int tail_rc = 0;                // status of a settlement another thread did for this context (reported by its avd_synchronize)
    int tail_help = 1;              // option "tail_help": 0 = never settle other contexts' tails, wait with hipStreamSynchronize (A/B, tests)
};

// profiling only (avd_set_profiling): the region that starts here on the context's stream is kernel `id`
inline void kmark(avd_ctx* ctx, int id)
{
    // the last slot is kept for the closing mark (AVD_K_COUNT): a call with more regions than slots loses its LATER regions' split (they are
    // accounted to the region of mark 94), never the end of the timeline; kmark_overflow says so (avd_kernel_ms fails then)
    if (!ctx->profiling) return;
    if (ctx->kmark_used >= 96 || (ctx->kmark_used == 95 && id != AVD_K_COUNT)) { ctx->kmark_overflow = 1; return; }
    hipEvent_t& e = ctx->kmark_ev[ctx->kmark_used];
    if (!e && hipEventCreate(&e) != hipSuccess) { e = nullptr; return; }
    if (hipEventRecord(e, ctx->stream) == hipSuccess) ctx->kmark_id[ctx->kmark_used++] = id;
}

template <typename T>
inline int dev_alloc(avd_ctx* ctx, T*& p, size_t count)
{
    if (p) { (void)hipFree(p); p = nullptr; }
    if (count == 0) return 0;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        p = nullptr;
        return AVD_ERR_NOMEM;
    }
    return 0;
}

// ---- stage launchers (each enqueues on ctx->stream) --------------------------------
int avd_ws_geometry(avd_ctx* ctx, int h, int w);                       // make (h, w) the current geometry (cached tables)
int avd_ws_reserve_frames(avd_ctx* ctx, int n, size_t rowbuf_elems, size_t lappart_elems);   // grow-only per-frame buffers
int avd_ws_reserve(avd_ctx* ctx, int n, int h, int w);                 // both, for one clip at offset 0
int avd_ws_reserve_fb(avd_ctx* ctx, int n);
int launch_preprocess(avd_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w,
                      int64_t row_stride, int64_t frame_stride);
// NV12 input: Y plane rows at d_y + f*frame_stride + y*row_stride, chroma rows at nv.uv + f*uv_frame_stride + (y/2)*uv_row_stride
int launch_preprocess_nv12(avd_ctx* ctx, const uint8_t* d_y, const Nv12Params& nv, int n, int h, int w,
                           int64_t row_stride, int64_t frame_stride);
int launch_hash(avd_ctx* ctx, int n, bool with_hamming = true);
int avd_calls_in_flight();      // avd_capi.hip: contexts of this process holding an enqueued, undrained avd_analyze_* call
int launch_farneback(avd_ctx* ctx, hipStream_t stream, const uint8_t* d_small, int n, int frame_off, int pair_off);
int launch_flow_stats(avd_ctx* ctx, hipStream_t stream, int n, int frame_off, int pair_off);
// avd_vit.hip (extension): patchify + bf16 MFMA GEMM; all pointers device
int launch_vit_patch_embed(avd_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, int64_t row_stride, int64_t frame_stride,
                           const uint16_t* d_wt, const float* d_bias, void* d_tokens, int tokens_bf16, uint16_t* d_patches);
void gemm_block_operand(const uint16_t* src_row_major, uint16_t* dst_blocked, int rows, int K);
constexpr int kGemmRowPad = 256;     // the GEMM's tile height: A is allocated in multiples of it
int launch_gemm_bf16_nt(avd_ctx* ctx, const uint16_t* d_a, const uint16_t* d_bt, const float* d_bias, void* d_c, int out_bf16,
                        int M, int N, int K);
// avd_cnn.hip (extension): ResNet-50-style forward as implicit GEMMs on the matrix cores
void cnn_param_counts(size_t* n_weights, size_t* n_biases);
int cnn_set_weights(avd_ctx* ctx, const uint16_t* weights, const float* biases);
int cnn_reserve(avd_ctx* ctx, int n);
int launch_cnn_forward(avd_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, int64_t row_stride, int64_t frame_stride);
int cnn_conv_host(avd_ctx* ctx, const uint16_t* x, int n, int hin, int win, int cin, const uint16_t* w, const float* bias, int cout, int ksize,
                  int stride, int relu, const uint16_t* residual, uint16_t* y);
// avd_comm.cpp: RCCL all-gather of the per-frame records (dlopen, no link-time dependency)
int comm_unique_id(std::string& err, void* id128);
int comm_init(avd_ctx* ctx, int rank, int world, const void* id128);
void comm_destroy(avd_ctx* ctx);
int comm_allgather_records(avd_ctx* ctx, const avd_frame_record* local, int count, avd_frame_record* all);
int comm_allgather_last_records(avd_ctx* ctx, int count, avd_frame_record* all);
// avd_audio.hip: per-window features of a mono float32 waveform (device pointers)
int launch_audio_features(avd_ctx* ctx, const float* d_wav, int64_t n, int win, avd_audio_window* d_out, int nwin);
// avd_fbfused.hip: all blur iterations of one pyramid level (w = 40 / 80 / 160 / 320) in one launch, one workgroup per pair
// zero_first: the initial flow is zero whatever the buffer holds (the coarsest level: no clearing launch)
// plist (may be null): the launch works on pairs plist[0 .. np)
int launch_fb_level(avd_ctx* ctx, hipStream_t stream, int w, const float* R, float* flow, int np, int iterations, int zero_first, const int* plist = nullptr);
// avd_fbfast.hip: blur iterations of one pyramid level, a pair spread over several workgroups (column strips), the horizontal window sums
// formed directly in double (the vertical chain stays literal).  mode 0: one iteration flow_in -> flow_out; 1 / 2: the same with flow_in =
// the coarser level's flow, resized on the fly (320 px: by the chain wave; 160 / 80 px: in a prologue through flow_tmp); 3 / 4: all three
// iterations in one launch (80 / 40 px), result in flow_out, flow_tmp the second buffer (4: behind the prologue)
int launch_fb_fast(avd_ctx* ctx, hipStream_t stream, int w, const float* R, const float* flow_in, float* flow_out, float* flow_tmp, float* mag_out,
                   int* flags, const int* pairdiff, int np, int zero_first, int mode);
// avd_farneback.hip: exact re-run of the m flagged pairs h_list[0 .. m) (pair indices inside the chunk the workspace holds; h_list pinned) -- all four
// levels with the exact kernels' launches from a compacted list, |flow| and the statistics of those pairs; np_chunk = pairs of the chunk
constexpr int kRerunTwoKernelMax = 32;
int launch_farneback_rerun(avd_ctx* ctx, hipStream_t stream, const int* h_list, int m, int pair_off, int np_chunk);
// avd_norm.hip (extensions): LayerNorm over rows of 256..2048 values, softmax over rows of logits; device pointers
int launch_layernorm(avd_ctx* ctx, const void* d_x, void* d_y, int bf16, long long rows, int cols, const float* d_gamma, const float* d_beta, float eps);
int launch_softmax(avd_ctx* ctx, const float* d_x, float* d_y, long long rows, int cols);
