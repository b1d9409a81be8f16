/*
 * avd.h -- C-ABI of libavd_hip.so, the MI355X (gfx950) implementation of the
 * per-frame video-analysis hot path of backtato/ai-video-detector.
 *
 * The reference has no native boundary: its hot path is 82 lines of Python
 * (reference app/analyzers/video.py:10-83) that call OpenCV per sampled frame.
 * Each entry point below replaces the OpenCV/numpy calls named beside it; the
 * Python shim that binds them (ctypes) is ai-video-detector_amd/avd_hip/_lib.py
 * and the drop-in module is ai-video-detector_amd/app/analyzers/video.py
 * (same module path / signature as the reference).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / HIP types in signatures.
 *   - every function returns 0 on success, a negative avd_status otherwise and
 *     never throws; avd_last_error(ctx) gives a message owned by ctx.
 *   - the caller owns every buffer; the library never frees or retains them.
 *   - `mem` says where an INPUT buffer lives: AVD_MEM_HOST (staged over PCIe) or
 *     AVD_MEM_DEVICE (already resident in HBM of ctx's device, e.g. a
 *     torch-ROCm tensor's data_ptr or a hardware decoder surface).
 *     OUTPUT buffers are always host memory (they are tiny).
 *   - one avd_ctx = one device, one HIP stream, one workspace.  A ctx is not
 *     re-entrant: use one ctx per thread (calls on different ctxs run concurrently;
 *     two calls on the SAME ctx are serialised by a lock, they do not overlap).
 *     A thread that waits in avd_synchronize may enqueue the exact re-run of another
 *     ctx's flagged pairs meanwhile (option "tail_help"); that ctx's results and
 *     errors still come out of its own avd_synchronize.
 *   - if no HIP device is usable avd_create fails (AVD_ERR_DEVICE): there is no
 *     CPU fallback in this library.
 */
#ifndef AVD_H
#define AVD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct avd_ctx avd_ctx;

enum avd_status {
    AVD_OK = 0,
    AVD_ERR_ARG = -1,       /* bad argument (null pointer, size out of range) */
    AVD_ERR_DEVICE = -2,    /* no usable HIP device / HIP runtime error */
    AVD_ERR_NOMEM = -3,     /* workspace allocation failed */
    AVD_ERR_UNSUPPORTED = -4 /* frame smaller than 32x32 (INTER_AREA upscaling) etc. */
};

enum avd_mem { AVD_MEM_HOST = 0, AVD_MEM_DEVICE = 1 };

#define AVD_SMALL 320          /* video.py:43 resize target */
#define AVD_HASH 32            /* video.py:36 aHash size */
#define AVD_ABI_VERSION 3      /* 3 (round 5): avd_frame_record.reserved is a criterion / level mask, option "fb_rerun_fused", "fb_fold_up" is a 3-bit mask */

/* One record per sampled frame: everything video.py:36-57 derives from pixels.
 * The scalar tail (tex variance, ai_susp, summary, timeline; video.py:54-83) is
 * O(N) float64 host work done by the caller from these records. */
typedef struct avd_frame_record {
    int64_t lap_sum;     /* sum  of cv2.Laplacian(gray, CV_64F)        (video.py:52) */
    int64_t lap_sumsq;   /* sum of squares of the same (exact integers)            */
    float   flow_mean;   /* np.mean(|flow|) vs the previous sampled frame (video.py:47); 0 for frame 0 */
    float   flow_var;    /* np.var(|flow|)                                 (video.py:48); 0 for frame 0 */
    int32_t ham;         /* popcount(hash ^ prev_hash)     (video.py:38); -1 for frame 0 */
    int32_t reserved;    /* non-zero: the pair (previous frame, this frame) was flagged ill-posed by the fast Farneback kernels and re-run exactly.
                          * bit k (0 = 320 px .. 3 = 40 px): level k met the solver's criterion; bit 4 + k: the border-sign criterion.  Levels a pair
                          * skipped because it was flagged already leave no bit. */
} avd_frame_record;      /* 32 bytes */

int avd_abi_version(void);

/* Lifetime.  avd_create binds HIP device `device_id`, creates a stream.  */
int avd_create(int device_id, avd_ctx** out);
void avd_destroy(avd_ctx* ctx);
const char* avd_last_error(const avd_ctx* ctx);

/* Replaces, for n decoded BGR frames (uint8, interleaved, frame f row y at
 * bgr + f*frame_stride + y*row_stride):
 *   cv2.cvtColor(BGR2GRAY) x3, cv2.resize(32x32, INTER_AREA) + mean threshold,
 *   cv2.resize(320x320) INTER_LINEAR, cv2.Laplacian(CV_64F) moments
 *   (video.py:4-8, 36, 43, 51-52).
 * Outputs (host, any may be NULL): small320 uint8[n][320*320], hash1024 uint8[n][1024]
 * (0/1), lap_sum / lap_sumsq int64[n]. */
int avd_preprocess_bgr(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w,
                       int64_t row_stride, int64_t frame_stride,
                       uint8_t* small320, uint8_t* hash1024,
                       int64_t* lap_sum, int64_t* lap_sumsq);

/* Replaces cv2.calcOpticalFlowFarneback(prev, cur, None, 0.5, 3, 15, 3, 5, 1.2, 0)
 * + np.sqrt/np.mean/np.var (video.py:45-48) for the n-1 consecutive pairs of
 * n 320x320 uint8 images.  Outputs host float[n-1].  flow_out (host, may be
 * NULL) receives the dense flow float[n-1][320][320][2]. */
int avd_farneback_pairs(avd_ctx* ctx, const uint8_t* small320, int mem, int n,
                        float* flow_mean, float* flow_var, float* flow_out);

/* The whole per-frame pixel path in one call: preprocess + Hamming + Farneback
 * + flow statistics, everything resident in HBM in between (video.py:36-52).
 * records: host avd_frame_record[n]. */
int avd_analyze_frames(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w,
                       int64_t row_stride, int64_t frame_stride,
                       avd_frame_record* records);

/* Same, but only enqueues the work on ctx's stream and returns; records (any HOST
 * memory) are filled by avd_synchronize: the device-to-host copy lands in a pinned
 * buffer of the library, so the call never blocks on it and several contexts can
 * keep clips in flight on one GPU.  One call may be outstanding per context (a
 * second one drains the first).  With AVD_MEM_HOST input the frames are staged by
 * hipMemcpyAsync from the caller's buffer, which is only asynchronous if that buffer
 * is pinned. */
int avd_analyze_frames_async(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w,
                             int64_t row_stride, int64_t frame_stride,
                             avd_frame_record* records);
int avd_synchronize(avd_ctx* ctx);

/* A BATCH of clips in one call (BASELINE.json configs[2] / configs[4]: many concurrent clips, mixed resolutions).  The
 * reference analyses one file per request in a sequential loop (app/analyzers/video.py:27-58); a service that has several
 * short clips waiting hands them over together: preprocess / hash / Hamming run per clip with that clip's geometry (the
 * tables of the last few geometries are cached per context: no allocation in steady state), then ONE Farneback launch
 * sequence covers the pairs of all clips (that stage is 320 x 320 whatever the source resolution), so thirteen 20-frame
 * clips fill the chip like one long clip instead of running thirteen sequences of 19 pairs each.
 * A clip is BGR (uv == NULL: data = frames, row_stride / frame_stride in bytes) or NV12 (data = Y plane, uv = interleaved
 * chroma plane, the four strides as for avd_analyze_frames_nv12).  records: host, sum of clips[i].n entries, clip after
 * clip; the first record of every clip has ham = -1 and flow 0 (video.py:37-41, 55).  Results are identical to calling
 * avd_analyze_frames per clip. */
typedef struct avd_clip {
    const uint8_t* data;
    const uint8_t* uv;
    int mem, n, h, w;
    int64_t row_stride, frame_stride, uv_row_stride, uv_frame_stride;
} avd_clip;
int avd_analyze_batch(avd_ctx* ctx, const avd_clip* clips, int nclips, avd_frame_record* records);
int avd_analyze_batch_async(avd_ctx* ctx, const avd_clip* clips, int nclips, avd_frame_record* records);

/* NV12 input (decode -> ingest).  Hardware decoders (VCN / rocDecode) and most software decoders
 * produce YUV 4:2:0, not BGR; the reference gets BGR because cv2.VideoCapture.retrieve() runs libswscale on the
 * decoded picture (reference app/analyzers/video.py:28-32).  These entry points take the decoder's surface
 * directly -- a Y plane uint8[h][w] and an interleaved U,V plane uint8[h/2][w], per frame at
 * y + f*y_frame_stride + row*y_row_stride and uv + f*uv_frame_stride + (row/2)*uv_row_stride -- and form, per pixel
 * and in registers, the BGR triple libswscale's C converter would have written (yuv2rgb.c tables: BT.601 limited
 * range, nearest chroma) before cv2's BGR2GRAY: 1.5 bytes per pixel cross PCIe / HBM instead of 3, and no BGR frame
 * exists anywhere.  Results equal avd_analyze_frames on the BGR frames of oracle/avd_oracle.c's avdo_nv12_to_bgr24
 * bit for bit; parity with a real libswscale is UNPINNED (restated from memory, x86 builds also dispatch to SIMD
 * code that differs by +-1).  Width and height must be even.  Outputs as for the BGR entry points. */
int avd_preprocess_nv12(avd_ctx* ctx, const uint8_t* y, const uint8_t* uv, int mem, int n, int h, int w,
                        int64_t y_row_stride, int64_t uv_row_stride, int64_t y_frame_stride, int64_t uv_frame_stride,
                        uint8_t* small320, uint8_t* hash1024, int64_t* lap_sum, int64_t* lap_sumsq);
int avd_analyze_frames_nv12(avd_ctx* ctx, const uint8_t* y, const uint8_t* uv, int mem, int n, int h, int w,
                            int64_t y_row_stride, int64_t uv_row_stride, int64_t y_frame_stride, int64_t uv_frame_stride,
                            avd_frame_record* records);
int avd_analyze_frames_nv12_async(avd_ctx* ctx, const uint8_t* y, const uint8_t* uv, int mem, int n, int h, int w,
                                  int64_t y_row_stride, int64_t uv_row_stride, int64_t y_frame_stride,
                                  int64_t uv_frame_stride, avd_frame_record* records);

/* ViT-B/16 patch embedding on the matrix cores -- a BUILD-DEFINED EXTENSION.  The reference
 * contains no learned model (its per-frame "model" is the closed form of app/analyzers/video.py:54-56); BASELINE.json's
 * north_star / configs[3] ask for this stage, so it exists, with caller-supplied weights, and is never part of
 * ai_score / timeline.  avd_vit_set_weights: weight_bf16 = bf16 bits [768 out][768 in], in = c*256 + py*16 + px (the
 * layout of a [768][3][16][16] conv weight, channels RGB), bias float[768] or NULL; both host pointers, copied.
 * avd_vit_patch_embed: each BGR frame is resized to 224x224 (bilinear), normalised ((x/255 - mean)/std, ImageNet
 * constants), cut into 196 patches of 16x16x3, rounded to bf16 and multiplied with the weights (f32 accumulation):
 * tokens [n][196][768], float (tokens_bf16 = 0) or bf16 bit patterns (tokens_bf16 = 1, round to nearest even), host or
 * device (tokens_mem).  If timing_reps > 0 and gemm_ms != NULL the GEMM kernel alone is launched timing_reps more times
 * between two HIP events and its mean duration is returned (bench hook). */
int avd_vit_set_weights(avd_ctx* ctx, const uint16_t* weight_bf16, const float* bias);
int avd_vit_patch_embed(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w, int64_t row_stride,
                        int64_t frame_stride, void* tokens, int tokens_mem, int tokens_bf16, int timing_reps, float* gemm_ms);

/* CNN extension (build-defined: the reference has no learned model; BASELINE.json's north_star
 * names a "CNN (ResNet-50-style) forward" on the matrix cores).  Never part of ai_score / timeline.  Topology: 7x7/2
 * stem of 64 channels + ReLU, 3x3/2 max pool, bottleneck stages [3, 4, 6, 3] of widths 64/128/256/512 (x4 out, the stride
 * on the 3x3 convolution, projection shortcut in the first block of a stage), global average pool, linear 2048 -> 1000;
 * batch norm folded into weights + bias; bf16 weights and activations, f32 accumulation.
 * avd_cnn_param_counts: elements of the flat parameter arrays.  avd_cnn_set_weights (host pointers, copied): weights =
 * bf16 bits, convolutions in forward order (stem; per block conv1 1x1, conv2 3x3, conv3 1x1, then the projection shortcut
 * where there is one), each [cout][kh][kw][cin], then the linear layer [1000][2048]; biases f32 in the same order.
 * avd_cnn_forward: each BGR frame is resized to 224x224 and normalised as in avd_vit_patch_embed; logits: host float
 * [n][1000]; any n (more than 128 frames are processed in passes of 128).  If timing_reps > 0 and forward_ms != NULL the whole forward pass (input conversion to logits, 57 launches)
 * is run timing_reps more times between two HIP events and its mean duration is returned (bench hook).
 * avd_cnn_conv: ONE convolution layer on host tensors (test entry): x NHWC bf16 [n][hin][win][cin], w [cout][k][k][cin],
 * bias f32[cout], optional residual NHWC [n][hout][wout][cout] added before the optional ReLU, y NHWC bf16; pad = k / 2;
 * cin % 32 == 0, cout % 64 == 0, ksize 1 or 3, stride 1 or 2. */
int avd_cnn_param_counts(size_t* n_weights, size_t* n_biases);
int avd_cnn_set_weights(avd_ctx* ctx, const uint16_t* weights_bf16, size_t n_weights, const float* biases, size_t n_biases);
int avd_cnn_forward(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w, int64_t row_stride, int64_t frame_stride,
                    float* logits, int timing_reps, float* forward_ms);
int avd_cnn_conv(avd_ctx* ctx, const uint16_t* x, int n, int hin, int win, int cin, const uint16_t* w, const float* bias,
                 int cout, int ksize, int stride, int relu, const uint16_t* residual, uint16_t* y);

/* LayerNorm and softmax -- the remaining elements of north_star's "conv / GEMM / LayerNorm / softmax stack"; BUILD-DEFINED
 * EXTENSIONS like the two above (the reference has no learned model), never part of ai_score.  One wave per row, the row
 * held in registers, wave-level shuffle reductions, one pass over HBM.
 * avd_layernorm: y = (x - mean) / sqrt(var + eps) * gamma + beta over the last dimension (biased variance, float32
 * statistics: torch.nn.functional.layer_norm); x / y [rows][cols] float32 (bf16 = 0) or bf16 bit patterns (bf16 = 1), both
 * host or both device (mem); cols in {256, 512, 768, 1024, 2048} (768 = the ViT-B/16 token width); gamma / beta host
 * float[cols].  avd_softmax: y = exp(x - max) / sum over rows of cols float32 logits (cols % 4 == 0, <= 4096; 1000 = the
 * CNN's classes).  If timing_reps > 0 and ms != NULL the kernel alone is launched timing_reps more times between two HIP
 * events and its mean duration is returned (bench hook). */
int avd_layernorm(avd_ctx* ctx, const void* x, int mem, int bf16, int64_t rows, int cols, const float* gamma, const float* beta,
                  float eps, void* y, int timing_reps, float* ms);
int avd_softmax(avd_ctx* ctx, const float* x, int mem, int64_t rows, int cols, float* y, int timing_reps, float* ms);

/* Audio analyzer: the per-window loop of reference app/analyzers/audio.py:40-61 for every window of a
 * mono float32 waveform at once.  wav: n samples (host or device); win: samples per window (the reference uses
 * int(sr * 0.5) = 8000 at 16 kHz; at most 8192); windows: host array of ceil(n / win) records, filled in order (the
 * last window may be shorter).  From a record the reference's per-window values follow as
 *   rms = sqrt(sumsq / length); zcr = float32(zero_cross) / float32(length - 1) / 2;
 *   flatness = exp(sum_log / nbins) / (sum_mag / nbins); rolloff = rolloff_index / max(1, nbins);
 *   centroid = sum_fmag / sum_mag          (mag = |rfft(seg * hanning)| + 1e-9, all sums in double)
 * and the scalar tail (audio.py:63-110) is host numpy (avd_hip/audio.py).  sumsq is accumulated in double where
 * audio.py:44 squares and averages in float32 (a deliberate deviation, inside 1e-6 of the reference's rms). */
typedef struct avd_audio_window {
    double sumsq;            /* sum of seg^2                                             (audio.py:44) */
    double sum_log;          /* sum of log(mag)                                          (audio.py:50) */
    double sum_mag;          /* sum of mag                                               (audio.py:50,51,61) */
    double sum_fmag;         /* sum of linspace(0,1,nbins) * mag                         (audio.py:60-61) */
    int32_t zero_cross;      /* sum of |diff(sign(seg))|                                 (audio.py:45) */
    int32_t length;          /* samples in this window */
    int32_t rolloff_index;   /* first k with running sum >= 0.85 * sum_mag, else 0       (audio.py:51-58) */
    int32_t nbins;           /* length / 2 + 1 */
} avd_audio_window;          /* 48 bytes */
int avd_audio_features(avd_ctx* ctx, const float* wav, int mem, int64_t n, int win, avd_audio_window* windows, int max_windows);

/* The one exchange step of the path: frames and clips shard across GPUs with no data-path collective;
 * only the 32-byte records are reassembled, with ONE RCCL all-gather over xGMI per batch (a few KB: latency-bound).
 * RCCL is bound at run time (dlopen), so single-GPU use never loads it.  avd_comm_unique_id: one rank creates the
 * 128-byte id and the caller distributes it (file, environment, torch's store ...); avd_comm_init: every rank, with
 * its own context (one process per GPU); avd_allgather_records: every rank passes `count` records (the same count on
 * every rank: whole clips per rank, or shards padded by the caller) and receives world * count in rank order.
 * Host pointers; the call blocks until the gathered records are in `all`. */
int avd_comm_unique_id(void* id128);
int avd_comm_init(avd_ctx* ctx, int rank, int world, const void* id128);
int avd_allgather_records(avd_ctx* ctx, const avd_frame_record* local, int count, avd_frame_record* all);
/* The same exchange straight from the device: the first `count` records of this context's LAST avd_analyze_* call
 * (blocking or asynchronous) are gathered from where k_records left them in HBM -- the collective is enqueued on the
 * context's stream behind the analysis, one copy brings the world * count gathered records to `all` (host).  Blocks until
 * they are there; an outstanding asynchronous call is drained too (its own records buffer is filled). */
int avd_allgather_last_records(avd_ctx* ctx, int count, avd_frame_record* all);

/* Stream ordering for AVD_MEM_DEVICE inputs.  A context launches on its own non-blocking stream, so device memory
 * that another stream is still writing (e.g. torch's current stream: a freshly computed tensor, a .contiguous()
 * copy, a decoder's colour-conversion kernel) must be ordered explicitly: everything enqueued on `producer_stream`
 * (a hipStream_t passed as a plain pointer; NULL = the default stream) before this call completes before anything
 * submitted to ctx afterwards starts.  No host synchronisation.  The caller keeps the input buffer alive until
 * avd_synchronize / the blocking call returns. */
int avd_wait_stream(avd_ctx* ctx, void* producer_stream);

/* Free the context's scratch memory (about 1.5 GB after a 120-frame 1080p clip) but keep the context; the next
 * call reserves it again.  Weights uploaded with avd_cnn_set_weights / avd_vit_set_weights are state, not scratch: they
 * stay.  For services that keep a pool of idle contexts. */
int avd_release_workspace(avd_ctx* ctx);

/* HIP-event timing of the work enqueued on ctx's stream between the two calls
 * (milliseconds).  avd_timer_stop synchronizes the stream. */
int avd_timer_start(avd_ctx* ctx);
int avd_timer_stop(avd_ctx* ctx, float* elapsed_ms);
/* Per-stage device time (ms, HIP events on ctx's stream) of the LAST avd_analyze_frames*
 * call when profiling was enabled with avd_set_profiling(ctx, 1): stage 0 = staging copies, fused
 * preprocess kernel (+ the 2 KB moment memset) and aHash kernel of every clip of the call, 1 = upload of the clip-start
 * table (calls with several clips; otherwise empty), 2 = Farneback (pyramid .. flow) + flow statistics + the record kernel
 * (Hamming distances, record assembly), 3 = records copy-out; 4 = duration of the fused level kernel at
 * 320x320 (all iterations of pyramid level 0 in one launch, events around the launch; with the two-kernel path selected
 * by AVD_FB_FUSED: mean duration of one k_uv launch), 5 = mean duration of one k_hscan<320> launch (two-kernel path only,
 * otherwise 0). */
int avd_set_profiling(avd_ctx* ctx, int enable);
/* Tuning / test switches.  "fb_mode": 1 (default; environment AVD_FB_MODE=fast) = the fast Farneback level kernel
 * (csrc/avd_fbfast.hip: a pair is spread over several workgroups; cv2's vertical running sums are kept literally, the
 * horizontal 15-column window sums are formed directly in double instead of as cv2's running double sum: the flow equals
 * the exact kernels' bit for bit on well-posed inputs, within 1e-5 px otherwise) WITH the exact re-run of the pairs it cannot follow.
 * Two criteria, evaluated by the level kernels at every pixel, iteration and level (derivation: profiles/r04_experiments.md section 1,
 * profiles/r05_experiments.md section 1; tools/experiments/fb_illposed_run.py):
 *   solver      the 2 x 2 normal equations are singular over whole regions -- determinant cancellation above 2000, or a displacement
 *               above 0.3 of the level width: the reference's own flow is chaotic there (ramps, stripes, isolated straight edges);
 *   border sign a flow component at the top / left image border is smaller than 1e-12 px (the size of the rounding residue of cv2's own
 *               running sums) while cv2's warp decides "inside the image" / "outside" by its SIGN and the two branches differ there:
 *               exactly periodic or static content whose true flow is zero (checkerboards; bit-identical frames but for a small patch).
 *               A pair of bit-identical frames is exempt (its zero flow is structural in cv2 too).
 * The host reads the flag words (with the records) and sends flagged pairs through the exact kernels before anything is handed to the caller,
 * so their results are the exact kernels', bit for bit: nothing is launched when nothing is flagged; up to 32 flagged pairs of a chunk run the
 * 160- / 320-px levels through the two-kernel path (a pair spread over many workgroups), more run the fused kernels (one workgroup per pair: a
 * clip of nothing but flagged pairs costs the exact mode's time -- a flagged pair leaves the fast launches at once).
 * avd_frame_record.reserved is non-zero for the frame that closes such a pair; avd_get_option "rerun_pairs" counts
 * them for the last drained call.  Stated guarantee of the mode, with no exception for any content: flow_mean / flow_var within rel 1e-6 and
 * ai_susp within 1e-6 of the oracle (north_star: 1e-4), tests/test_gpu_fbfast.py + the content soak over 28 families in tests/test_gpu_soak.py.
 * "fb_rerun" (default 1, environment AVD_FB_RERUN): 0 switches the re-run off (A/B, tests).  "fb_rerun_fused" (default 0xC): level mask of the
 * fused kernel in the few-pairs re-run (bit 3 is always set); no effect on results.
 * "tail_help" (default 1; no effect on results): the flag words of an asynchronous call's last chunk reach the host with its records, and the host
 * enqueues the re-run.  With the option on, a thread that waits in avd_synchronize -- for its own fast pass or its own re-run -- does that for the
 * OTHER contexts of the device whose records have arrived, instead of each clip's re-run starting only when its own avd_synchronize is reached
 * (one host thread, three fully flagged 120-frame clips in flight: 76 k -> 111 k frames/s; nothing flagged: no difference).  0 = wait in the runtime.
 * 0 (AVD_FB_MODE=exact) = the exact kernels, bit-identical to the oracle everywhere, one workgroup per pair.
 * "fb_fold_up" (fast mode, bit mask, default 5, environment AVD_FB_FOLD_UP; no effect on results): 1 = the first launch of the 320-px level
 * resizes the 160-px level's flow itself instead of reading the output of a separate resize launch; 2 = the 160- and 80-px levels do so in a
 * prologue of their first launch; 4 = the 80- and 40-px levels (a pair is one workgroup there) run their three iterations in one launch.
 * "fb_fold_blur" (default 1, environment AVD_FB_FOLD_BLUR; no effect on results): the 3 x 3 Gaussian of the 320-px pyramid scale is formed inside
 * the polynomial expansion (same two float passes, same operation order) instead of being written by the pyramid kernel and read back;
 * avd_debug_fetch "pyr0" exists with the option off only (an error otherwise: the buffer is not even allocated).
 * "fb_wide160" (fast mode, environment AVD_FB_WIDE160): 1 = the 160-px level runs a pair as ONE strip of three 64-column blocks
 * (119 workgroups, fewer CU-microseconds: +2.4 % frames/s with clips in flight), 0 = as two 80-column strips (238 workgroups, each
 * launch 10 us shorter: one clip alone finishes ~25 us sooner), 2 (default) = chosen when the call is enqueued: one strip if another context of the
 * process holds an undrained avd_analyze_* call, two strips if this clip has the chip to itself.  Same guarantee; the two shapes group the solver's
 * window sums differently (bit-identical on well-posed content, tests/test_gpu_fbfast.py).
 * "fb_fused" (exact mode only, no effect on results): bit k set = pyramid level k (0 = 320x320 .. 3 = 40x40)
 * of the Farneback stage runs the fused level kernel (default 0xF, or the environment variable AVD_FB_FUSED at
 * avd_create); clear = the two-kernel path that exchanges its double intermediate through HBM.  "cnn_tiles": tiling of the
 * CNN extension's convolutions, 0 = by layer shape (default), 1 = 256-pixel tiles everywhere, 2 = 128 x 128 tiles wherever the
 * channel count allows (the accumulation order of an output does not depend on the tiling: results are bit-identical).
 * "cnn_fuse" (default 2): a bottleneck block's 3x3 and its expanding 1x1 run as ONE launch in the 56x56 and 28x28 stages
 * (the mid activation stays in LDS; bit-identical to the two layers); 2 = in their stride-1 blocks the 3x3 also reads its nine taps from one
 * copy of its input in LDS instead of gathering them tap by tap; 1 = gathering form everywhere; 0 = layer by layer.  "cnn_chunk" (default 128, 1 ... 1024):
 * frames per forward pass of avd_cnn_forward (activation scratch: 4 x 1.6 MB per frame).
 * avd_get_option returns the value an option has now (environment defaults included) and the read-only "rerun_pairs". */
int avd_set_option(avd_ctx* ctx, const char* name, int value);
int avd_get_option(avd_ctx* ctx, const char* name, int* value);
int avd_stage_ms(avd_ctx* ctx, int stage, float* ms);
/* Per-kernel device time (ms) of the LAST drained avd_analyze_* call with profiling on: HIP events on the context's stream in
 * front of every kernel (group) of the path; a level's figure is the sum of its launches (fast mode: three, one per blur
 * iteration).  bench.py's roofline.kernels is built from these, with clips run alone.  A call with more kernel regions than the library
 * records (96: ~40 clips in one batch) makes avd_kernel_ms fail rather than report partial sums. */
enum avd_kernel_id {
    AVD_K_PREPROCESS = 0,  /* k_preprocess_vec / k_preprocess_nv12 (+ staging copies of host input) */
    AVD_K_HASH,            /* k_hash: 32 x 32 INTER_AREA cells, mean threshold, moment reduction */
    AVD_K_PYRAMID,         /* k_pyramid_all: Gaussian blur + decimation, four scales */
    AVD_K_POLYEXP,         /* k_polyexp_all: polynomial expansion, four scales */
    AVD_K_LEVEL40, AVD_K_FLOWUP80, AVD_K_LEVEL80, AVD_K_FLOWUP160, AVD_K_LEVEL160, AVD_K_FLOWUP320, AVD_K_LEVEL320,
    AVD_K_RERUN,           /* exact re-run of the pairs the fast level kernels flagged (launched when the call is drained; 0 when none was) */
    AVD_K_STATS,           /* k_stats_pair (exact mode: + k_mag) */
    AVD_K_RECORDS,         /* k_records: Hamming distances, record assembly */
    AVD_K_OTHER,           /* clip-table upload, records copy-out */
    AVD_K_COUNT
};
int avd_kernel_ms(avd_ctx* ctx, int kernel_id, float* ms);

/* Test hook: copy an internal device buffer of the last call to host.
 * name: "area" uint8[n][1024]; "pyr<L>" float[n][hL][wL]; "poly<L>" float[n][hL][wL][5];
 * "flow<L>" float[n-1][2][hL][wL] (planar, after the last iteration at level L).
 * Returns the number of bytes copied (>=0) or a negative status. */
int64_t avd_debug_fetch(avd_ctx* ctx, const char* name, void* out, size_t out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* AVD_H */
