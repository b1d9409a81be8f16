// avd_capi.hip -- C-ABI of libavd_hip.so (include/avd.h): context, workspace, entry points.
// Host C++; every kernel lives in avd_preprocess.hip / avd_farneback.hip.
#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <new>
#include <thread>
#include "avd_internal.h"

namespace {

constexpr int kFbChunk = 128;      // Farneback pairs the scratch holds at first (one 120-frame clip)
constexpr int kFbChunkMax = 512;   // ... and at most (2.6 GB): a batch of short clips runs as ONE launch sequence up to here;
                                   // longer calls are processed in chunks of the reserved size with a one-frame overlap

int rows_per_band_for(int w)
{
    // LDS tile = (rows+2) * pitch bytes, kept under 48 KiB so that >= 3 workgroups fit a CU
    const int pitch = ((w + 32 + 15) / 16) * 16;
    const bool big = [] { const char* e = std::getenv("AVD_PRE_NT"); return e && std::atoi(e) == 512; }();
    int r = ((big ? 72 : 48) * 1024) / pitch - 2;
    int cap = 14;                      // 16 tile rows + LDS tables = 37 KiB at 1080p: 4 workgroups per CU
    if (w % 16 == 0 && w / 16 <= 256) {
        // aligned fast path: a lane holds at most 9 row chunks (3 x 16 B each) in registers, so the
        // tile may have at most 9 * (rows covered per pass) rows -- 7-row bands at 4K
        const int rpp = (big ? 512 : 256) / (w / 16);
        cap = std::min(big ? 30 : cap, 9 * rpp - 2);
    }
    if (const char* e = std::getenv("AVD_ROWS_PER_BAND")) cap = std::max(1, std::atoi(e));   // tuning knob
    r = std::min(r, cap);
    return std::max(r, 1);
}

struct TableBlob {
    std::vector<uint8_t> bytes;
    template <typename T>
    size_t push(const std::vector<T>& v)
    {
        size_t off = (bytes.size() + 15) / 16 * 16;
        bytes.resize(off + v.size() * sizeof(T));
        std::memcpy(bytes.data() + off, v.data(), v.size() * sizeof(T));
        return off;
    }
};

// keep_weights: avd_release_workspace gives SCRATCH back; the weights a caller uploaded (avd_cnn_set_weights,
// avd_vit_set_weights) are state, not scratch, and stay
void free_ws(Workspace& ws, bool keep_weights = false)
{
    auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    uint16_t *kw = ws.d_cnn_w, *vw = ws.d_vit_w;
    float *kb = ws.d_cnn_b, *vb = ws.d_vit_bias;
    const int vhb = ws.vit_has_bias;
    std::vector<size_t> woff = ws.cnn_w_off;
    const size_t fcoff = ws.cnn_fc_off;
    if (keep_weights) { ws.d_cnn_w = nullptr; ws.d_cnn_b = nullptr; ws.d_vit_w = nullptr; ws.d_vit_bias = nullptr; }
    F(ws.d_stage); F(ws.d_small); F(ws.d_rowbuf); F(ws.d_area); F(ws.d_hash); F(ws.d_ham); F(ws.d_lap); F(ws.d_lap_part);
    for (Geom& g : ws.geoms) F(g.d_tables);
    F(ws.d_clipstart);
    if (ws.h_clipstart) (void)hipHostFree(ws.h_clipstart);
    for (int k = 0; k < AVD_FB_LEVELS; k++) { F(ws.d_pyr[k]); F(ws.d_poly[k]); F(ws.d_flow[k]); F(ws.d_flow2[k]); }
    F(ws.d_vs); F(ws.d_vs0); F(ws.d_flow_il); F(ws.d_stats); F(ws.d_rec); F(ws.d_mag); F(ws.d_fbflags); F(ws.d_pairdiff);
    F(ws.d_rlist); F(ws.d_vs_rerun); F(ws.d_vs0_rerun);
    if (ws.h_rlist) (void)hipHostFree(ws.h_rlist);
    F(ws.d_vit_w); F(ws.d_vit_bias); F(ws.d_vit_patches); F(ws.d_vit_tokens);
    F(ws.d_audio_tab); F(ws.d_audio_buf); F(ws.d_audio_out);
    F(ws.d_cnn_w); F(ws.d_cnn_b); F(ws.d_cnn_img); F(ws.d_cnn_pool); F(ws.d_cnn_logits);
    for (int i = 0; i < 4; i++) F(ws.d_cnn_act[i]);
    if (ws.h_rec) (void)hipHostFree(ws.h_rec);
    ws = Workspace{};
    if (keep_weights) {
        ws.d_cnn_w = kw; ws.d_cnn_b = kb; ws.d_vit_w = vw; ws.d_vit_bias = vb; ws.vit_has_bias = vhb;
        ws.cnn_w_off = woff; ws.cnn_fc_off = fcoff;
    }
}

int check_geometry(avd_ctx* ctx, int n, int h, int w, int64_t row_stride, int64_t frame_stride)
{
    if (n < 0 || h <= 0 || w <= 0 || h > 16384 || w > 16384) { ctx->err = "bad frame geometry"; return AVD_ERR_ARG; }
    if (row_stride < (int64_t)w * 3 || (n > 1 && frame_stride < row_stride * (h - 1) + (int64_t)w * 3)) {
        ctx->err = "strides smaller than the frame"; return AVD_ERR_ARG;
    }
    if (h < AVD_HASH || w < AVD_HASH) { ctx->err = "frame smaller than 32x32: INTER_AREA upscaling is not on the path"; return AVD_ERR_UNSUPPORTED; }
    return 0;
}

}  // namespace

// ---- workspace -----------------------------------------------------------------------
// Build the tables of geometry (h, w) into cache entry g (its old table buffer is re-used when large enough).
static int build_geom(avd_ctx* ctx, Geom& g, int h, int w)
{
    LinearTab lt;
    AreaTab at;
    build_linear_tab(h, w, AVD_SMALL, AVD_SMALL, lt);
    int rc = build_area_tab(h, w, AVD_HASH, AVD_HASH, at);
    if (rc) { ctx->err = "unsupported geometry for INTER_AREA"; return rc; }
    const int R = rows_per_band_for(w);
    const int nbands = (h + R - 1) / R;
    std::vector<int> band_dy(nbands + 1, AVD_SMALL);
    for (int b = 0; b <= nbands; b++) {
        int d = 0;
        while (d < AVD_SMALL && lt.y0[d] < b * R) d++;
        band_dy[b] = d;
    }
    band_dy[nbands] = AVD_SMALL;
    TableBlob tb;
    std::vector<LinTap> lxt(AVD_SMALL), lyt(AVD_SMALL);
    for (int d = 0; d < AVD_SMALL; d++) {
        lxt[d] = LinTap{(short)lt.x0[d], (short)lt.x1[d], lt.a0[d], lt.a1[d]};
        lyt[d] = LinTap{(short)lt.y0[d], (short)lt.y1[d], lt.b0[d], lt.b1[d]};
    }
    const size_t o_lxt = tb.push(lxt), o_lyt = tb.push(lyt);
    const size_t o_band = tb.push(band_dy);
    const size_t o_axb = tb.push(at.x.begin), o_axc = tb.push(at.x.count);
    const size_t o_axf = tb.push(at.x.w_first), o_axm = tb.push(at.x.w_mid), o_axl = tb.push(at.x.w_last);
    const size_t o_ayb = tb.push(at.y.begin), o_ayc = tb.push(at.y.count);
    const size_t o_ayf = tb.push(at.y.w_first), o_aym = tb.push(at.y.w_mid), o_ayl = tb.push(at.y.w_last);
    g.h = g.w = 0;                                     // not valid until everything below succeeded
    uint8_t* dt = (uint8_t*)g.d_tables;
    if (g.tables_bytes < tb.bytes.size()) {
        // an evicted entry's tables may still be read by a kernel in flight on this context's stream: drain it first
        // (only on the first use of a FIFTH distinct geometry, or of one with larger tables)
        if (dt) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        g.tables_bytes = 0;
        if (int e = dev_alloc(ctx, dt, tb.bytes.size())) { g.d_tables = nullptr; return e; }
        g.d_tables = dt; g.tables_bytes = tb.bytes.size();
    } else if (dt) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // re-using the buffer of an evicted geometry
    }
    HIP_TRY(ctx, hipMemcpyAsync(dt, tb.bytes.data(), tb.bytes.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // tb goes out of scope
    PreParams& P = g.pre;
    P = PreParams{};
    P.lxt = (const LinTap*)(dt + o_lxt); P.lyt = (const LinTap*)(dt + o_lyt);
    P.band_dy = (const int*)(dt + o_band);
    P.ax_begin = (const int*)(dt + o_axb); P.ax_count = (const int*)(dt + o_axc);
    P.ax_first = (const float*)(dt + o_axf); P.ax_mid = (const float*)(dt + o_axm); P.ax_last = (const float*)(dt + o_axl);
    P.area_fast = at.fast;
    P.area_x_uniform4 = 1;
    for (int d = 0; d < AVD_HASH; d++)
        if ((at.x.begin[d] & 3) || (at.x.count[d] & 3) || at.x.count[d] < 4 || at.x.w_first[d] != at.x.w_mid[d] ||
            at.x.w_last[d] != at.x.w_mid[d])
            P.area_x_uniform4 = 0;
    P.h = h; P.w = w; P.rows_per_band = R; P.nbands = nbands;
    P.pitch = ((w + 32 + 15) / 16) * 16;
    HashParams& H = g.hsh;
    H = HashParams{};
    H.ay_begin = (const int*)(dt + o_ayb); H.ay_count = (const int*)(dt + o_ayc);
    H.ay_first = (const float*)(dt + o_ayf); H.ay_mid = (const float*)(dt + o_aym); H.ay_last = (const float*)(dt + o_ayl);
    H.area_fast = at.fast;
    H.fast_area = at.iscale_x * at.iscale_y;
    H.fast_simd_w = (at.fast && at.iscale_x == 2 && at.iscale_y == 2) ? (AVD_HASH & ~7) : 0;
    H.h = h;
    g.h = h; g.w = w;
    return 0;
}

// Make (h, w) the current geometry: ws.pre / ws.hsh are copies of its cache entry.  A hit costs nothing; a miss builds the
// tables in a free entry, or in the least recently used one.
int avd_ws_geometry(avd_ctx* ctx, int h, int w)
{
    Workspace& ws = ctx->ws;
    Geom* hit = nullptr;
    Geom* victim = nullptr;                            // a free entry, else the least recently used one
    for (Geom& g : ws.geoms) {
        if (g.h == h && g.w == w) { hit = &g; break; }
        if (!victim || (g.h == 0 && victim->h != 0) || (g.h != 0 && victim->h != 0 && g.stamp < victim->stamp)) victim = &g;
    }
    if (!hit) {
        ws.h = ws.w = 0;
        if (int e = build_geom(ctx, *victim, h, w)) return e;
        hit = victim;
    }
    hit->stamp = ++ws.geom_clock;
    ws.pre = hit->pre; ws.hsh = hit->hsh;
    ws.h = h; ws.w = w;
    return 0;
}

// Per-frame buffers for n frames (all clips of a call), `rowbuf_elems` floats of INTER_AREA row partials and `lappart_elems`
// long longs of Laplacian partial moments.  Grow-only: in steady state nothing is allocated or freed.
int avd_ws_reserve_frames(avd_ctx* ctx, int n, size_t rowbuf_elems, size_t lappart_elems)
{
    Workspace& ws = ctx->ws;
    if (n > ws.cap_n) {
        const int cap = std::max(n, 1);
        ws.cap_n = 0;                                  // not valid again until every buffer below exists
        if (int e = dev_alloc(ctx, ws.d_small, (size_t)cap * AVD_NPIX)) return e;
        if (int e = dev_alloc(ctx, ws.d_area, (size_t)cap * 1024)) return e;
        if (int e = dev_alloc(ctx, ws.d_hash, (size_t)cap * 1024)) return e;
        if (int e = dev_alloc(ctx, ws.d_ham, (size_t)cap)) return e;
        if (int e = dev_alloc(ctx, ws.d_lap, (size_t)cap * 2)) return e;
        if (int e = dev_alloc(ctx, ws.d_rec, (size_t)cap)) return e;
        if (int e = dev_alloc(ctx, ws.d_clipstart, (size_t)cap)) return e;
        if (ws.h_rec) { (void)hipHostFree(ws.h_rec); ws.h_rec = nullptr; }
        if (ws.h_clipstart) { (void)hipHostFree(ws.h_clipstart); ws.h_clipstart = nullptr; }
        if (hipHostMalloc((void**)&ws.h_rec, sizeof(avd_frame_record) * cap, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void**)&ws.h_clipstart, sizeof(int) * cap, hipHostMallocDefault) != hipSuccess) {
            ctx->err = "hipHostMalloc failed"; return AVD_ERR_NOMEM;
        }
        ws.cap_n = cap;
    }
    if (rowbuf_elems > ws.rowbuf_cap) {
        ws.rowbuf_cap = 0;
        if (int e = dev_alloc(ctx, ws.d_rowbuf, rowbuf_elems)) return e;
        ws.rowbuf_cap = rowbuf_elems;
    }
    if (lappart_elems > ws.lappart_cap) {
        ws.lappart_cap = 0;
        if (int e = dev_alloc(ctx, ws.d_lap_part, lappart_elems)) return e;
        ws.lappart_cap = lappart_elems;
    }
    return 0;
}

static size_t rowbuf_elems_for(const Workspace& ws, int n) { return (size_t)n * ws.pre.h * AVD_HASH; }
static size_t lappart_elems_for(const Workspace& ws, int n) { return (size_t)n * ws.pre.nbands * 8 * 2; }

// one clip at offset 0 of the buffers
int avd_ws_reserve(avd_ctx* ctx, int n, int h, int w)
{
    if (int e = avd_ws_geometry(ctx, h, w)) return e;
    Workspace& ws = ctx->ws;
    ws.f0 = 0; ws.rowbuf_off = 0; ws.lappart_off = 0;
    return avd_ws_reserve_frames(ctx, n, rowbuf_elems_for(ws, n), lappart_elems_for(ws, n));
}

constexpr size_t kPairDiffTilesHost = 20;     // = kPairDiffTiles of avd_fb_device.h (the pyramid kernel's 160-px tiles per frame)

// Farneback scratch for min(n - 1, kFbChunkMax) pairs, at least kFbChunk; grows when a call brings more pairs
int avd_ws_reserve_fb(avd_ctx* ctx, int n)
{
    Workspace& ws = ctx->ws;
    const int want = std::max(kFbChunk, std::min(std::max(n - 1, 0), kFbChunkMax));
    if (want > ws.fb_cap) {
        const size_t nf = (size_t)want + 2, np = (size_t)want;       // +1 frame: the two segments of a chunk overlap by one frame
        ws.fb_cap = 0;
        for (int k = 0; k < AVD_FB_LEVELS; k++) {
            const size_t plane = (size_t)(AVD_SMALL >> k) * (AVD_SMALL >> k);
            // the 320-px scale of the pyramid exists only with fb_fold_blur off (the polynomial expansion forms that blur itself): 49 MB per 120 frames
            if (k > 0 || !ctx->fb_fold_blur) { if (int e = dev_alloc(ctx, ws.d_pyr[k], nf * plane)) return e; }
            else if (ws.d_pyr[0]) { (void)hipFree(ws.d_pyr[0]); ws.d_pyr[0] = nullptr; }
            if (int e = dev_alloc(ctx, ws.d_poly[k], nf * 5 * plane)) return e;
            if (int e = dev_alloc(ctx, ws.d_flow[k], np * 2 * plane)) return e;
            if (int e = dev_alloc(ctx, ws.d_flow2[k], np * 2 * plane)) return e;
            ws.flow_res[k] = nullptr;
        }
        if (int e = dev_alloc(ctx, ws.d_stats, np * 2)) return e;
        if (int e = dev_alloc(ctx, ws.d_mag, np * (size_t)AVD_NPIX)) return e;
        if (int e = dev_alloc(ctx, ws.d_fbflags, np)) return e;
        if (int e = dev_alloc(ctx, ws.d_pairdiff, np * kPairDiffTilesHost)) return e;
        if (ws.d_rlist) { (void)hipFree(ws.d_rlist); ws.d_rlist = nullptr; ws.rlist_cap = 0; }
        if (ws.h_rlist) { (void)hipHostFree(ws.h_rlist); ws.h_rlist = nullptr; }
        if (hipHostMalloc((void**)&ws.h_rlist, sizeof(int) * np) != hipSuccess) { ws.h_rlist = nullptr; ctx->err = "hipHostMalloc (re-run list)"; return AVD_ERR_NOMEM; }
        if (ws.d_vs) { (void)hipFree(ws.d_vs); ws.d_vs = nullptr; }
        if (ws.d_vs0) { (void)hipFree(ws.d_vs0); ws.d_vs0 = nullptr; }
        if (ws.d_flow_il) { (void)hipFree(ws.d_flow_il); ws.d_flow_il = nullptr; }
        ws.fb_cap = want;
    }
    if (!ctx->fb_fold_blur && !ws.d_pyr[0])
        if (int e = dev_alloc(ctx, ws.d_pyr[0], ((size_t)ws.fb_cap + 2) * AVD_NPIX)) return e;
    // the double intermediate of the two-kernel fallback (4 MB per pair): only when that path is selected
    const bool two_kernel = ctx->fb_mode == 0 && ctx->fb_fused != 0xF;
    if (two_kernel && !ws.d_vs) {
        const size_t np = (size_t)ws.fb_cap;
        if (int e = dev_alloc(ctx, ws.d_vs0, np * 5 * AVD_SMALL * 8)) return e;
        if (int e = dev_alloc(ctx, ws.d_vs, np * (5 * AVD_NPIX + 512))) return e;      // + one pad tile per pair
    }
    return 0;
}

// ---- record assembly -------------------------------------------------------------------
// One workgroup per frame, ONE launch per Farneback chunk (it follows the flow statistics): the frame's Laplacian moments,
// its Hamming distance to the previous frame (video.py:8: the 1024 aHash bits, one per byte, of both frames) and the
// statistics of pair (f - 1, f).  The first frame of a clip has no predecessor (ham = -1, flow 0; the pair the Farneback stage
// computed across a clip boundary is ignored): f == 0, or clipstart[f] != 0 when the call holds several clips
// (clipstart == nullptr: one clip).  with_flow == 0: a call without any pair (one frame).
__global__ __launch_bounds__(256) void k_records(const unsigned long long* __restrict__ lap, const uint8_t* __restrict__ bits,
                                                const float* __restrict__ stats, const int* __restrict__ flags, int stats_off,
                                                avd_frame_record* __restrict__ rec, int f0, const int* __restrict__ clipstart, int with_flow)
{
    __shared__ int wsum[4];
    const int f = f0 + blockIdx.x, tid = threadIdx.x;
    const bool first = f == 0 || (clipstart && clipstart[f] != 0);
    int c = 0;
    if (!first) {
        const unsigned a = reinterpret_cast<const unsigned*>(bits + (int64_t)f * 1024)[tid];
        const unsigned b = reinterpret_cast<const unsigned*>(bits + (int64_t)(f - 1) * 1024)[tid];
        c = __popc(a ^ b);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if ((tid & 63) == 0) wsum[tid >> 6] = c;
    __syncthreads();
    if (tid != 0) return;
    avd_frame_record r;
    r.lap_sum = (int64_t)lap[2 * f];
    r.lap_sumsq = (int64_t)lap[2 * f + 1];
    r.ham = first ? -1 : wsum[0] + wsum[1] + wsum[2] + wsum[3];
    // fast Farneback mode: non-zero when pair (f - 1, f) met the ill-posedness criterion and was re-run by the exact kernels
    r.reserved = (first || !with_flow || !flags) ? 0 : flags[f - 1 - stats_off];
    r.flow_mean = (first || !with_flow) ? 0.f : stats[2 * (f - 1 - stats_off)];
    r.flow_var = (first || !with_flow) ? 0.f : stats[2 * (f - 1 - stats_off) + 1];
    rec[f] = r;
}

// Fast mode: the pairs of the chunk in the workspace whose flag word is set (h_flags[0 .. np), as the host has just read them) go through the
// exact kernels again (launch_farneback_rerun: enqueued, not drained).  Returns the number of such pairs, or a negative status.
static int rerun_flagged(avd_ctx* ctx, const int* h_flags, int stride, int np)
{
    Workspace& ws = ctx->ws;
    int m = 0;
    for (int i = 0; i < np; i++)
        if (h_flags[(size_t)i * stride] != 0) ws.h_rlist[m++] = i;
    if (m == 0) return 0;
    if (int e = launch_farneback_rerun(ctx, ctx->stream, ws.h_rlist, m, 0, np)) return e < 0 ? e : -1;
    return m;
}

__global__ void k_wake() {}

static void launch_records(avd_ctx* ctx, int p0, int np, int fa, const int* clipstart)
{
    Workspace& ws = ctx->ws;
    kmark(ctx, AVD_K_RECORDS);
    hipLaunchKernelGGL(k_records, dim3(p0 + np + 1 - fa), dim3(256), 0, ctx->stream, (const unsigned long long*)ws.d_lap,
                       (const uint8_t*)ws.d_hash, (const float*)ws.d_stats,
                       (const int*)(ctx->fb_mode == 1 && ctx->fb_rerun ? ws.d_fbflags : nullptr), p0, ws.d_rec, fa, clipstart, 1);
}

// Farneback + stats over all n-1 pairs in chunks; pair p = (frame p, frame p+1).
// Fast mode: the level kernels flag the pairs they cannot follow; the HOST reads the flag words and sends those pairs through the exact kernels
// (nothing is launched when nothing is flagged -- the usual case).  For the last chunk of an asynchronous call that happens when the call is
// drained (ctx->tail, impl_synchronize): the flags travel with the records.  A chunk that is followed by another one (> 512 pairs in a call), and
// every chunk of a call that hands statistics straight to the host, is settled here, before its scratch is reused.
static void tail_unregister(avd_ctx* ctx);
static int run_flow_chunks(avd_ctx* ctx, const uint8_t* d_small, int n, float* h_mean, float* h_var,
                           float* h_flow_out, bool into_records, const int* records_clipstart = nullptr)
{
    Workspace& ws = ctx->ws;
    tail_unregister(ctx);
    ctx->tail.active = 0;
    if (n < 2) return 0;
    if (int e = avd_ws_reserve_fb(ctx, n)) return e;
    const int chunk = ws.fb_cap;
    if (h_flow_out && !ws.d_flow_il)
        if (int e = dev_alloc(ctx, ws.d_flow_il, (size_t)chunk * AVD_NPIX * 2)) return e;
    float* saved_il = ws.d_flow_il;
    if (!h_flow_out) ws.d_flow_il = nullptr;
    const bool flagged_mode = ctx->fb_mode == 1 && ctx->fb_rerun;
    const bool host_wants = h_mean || h_var || h_flow_out;
    int rc = 0;
    for (int p0 = 0; p0 < n - 1 && rc == 0; p0 += chunk) {
        const int np = std::min(chunk, n - 1 - p0);
        const bool last = p0 + np >= n - 1;
        const uint8_t* base = d_small + (size_t)p0 * AVD_NPIX;
        rc = launch_farneback(ctx, ctx->stream, base, np + 1, 0, 0);
        if (rc) break;
        rc = launch_flow_stats(ctx, ctx->stream, np + 1, 0, 0);
        if (rc) break;
        if (flagged_mode && (!last || host_wants || !into_records)) {
            std::vector<int> fl((size_t)np, 0);
            hipError_t e = hipMemcpyAsync(fl.data(), ws.d_fbflags, sizeof(int) * np, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { ctx->err = hipGetErrorString(e); rc = AVD_ERR_DEVICE; break; }
            const int m = rerun_flagged(ctx, fl.data(), 1, np);
            if (m < 0) { rc = m; break; }
            if (!into_records) ctx->last_rerun += m;
        }
        // frames p0 + 1 .. p0 + np, and frame 0 with the first chunk
        const int fa = p0 == 0 ? 0 : p0 + 1;
        if (into_records) {
            launch_records(ctx, p0, np, fa, records_clipstart);
            if (flagged_mode && last && !host_wants) {
                ctx->tail.active = 1; ctx->tail.p0 = p0; ctx->tail.np = np; ctx->tail.fa = fa; ctx->tail.n = n; ctx->tail.clipstart = records_clipstart;
            }
        }
        if (host_wants) {
            std::vector<float> st((size_t)np * 2);
            hipError_t e = hipMemcpyAsync(st.data(), ws.d_stats, sizeof(float) * 2 * np, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess && h_flow_out)
                e = hipMemcpyAsync(h_flow_out + (size_t)p0 * AVD_NPIX * 2, ws.d_flow_il,
                                   sizeof(float) * 2 * AVD_NPIX * np, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { ctx->err = hipGetErrorString(e); rc = AVD_ERR_DEVICE; break; }
            for (int i = 0; i < np; i++) {
                if (h_mean) h_mean[p0 + i] = st[2 * i];
                if (h_var) h_var[p0 + i] = st[2 * i + 1];
            }
        }
    }
    ws.d_flow_il = saved_il;
    return rc;
}

// Make `src` (host or device) available on the device; host input is staged over PCIe.
static int stage_input(avd_ctx* ctx, const uint8_t* src, int mem, size_t bytes, const uint8_t** d_out)
{
    if (mem == AVD_MEM_DEVICE) { *d_out = src; return 0; }
    if (mem != AVD_MEM_HOST) { ctx->err = "mem must be AVD_MEM_HOST or AVD_MEM_DEVICE"; return AVD_ERR_ARG; }
    Workspace& ws = ctx->ws;
    if (ws.stage_bytes < bytes) {
        ws.stage_bytes = 0;                        // not valid again until the buffer exists
        if (int e = dev_alloc(ctx, ws.d_stage, bytes)) return e;
        ws.stage_bytes = bytes;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ws.d_stage, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *d_out = ws.d_stage;
    return 0;
}

static void stage_mark(avd_ctx* ctx, int i)
{
    if (!ctx->profiling) return;
    if (i == 0) ctx->stage_marks = 0;
    if (hipEventRecord(ctx->stage_ev[i], ctx->stream) == hipSuccess) ctx->stage_marks++;
}

// contexts of this process (any device) that hold an enqueued, not yet drained avd_analyze_* call: what "fb_wide160" = 2 decides by
std::atomic<int> g_calls_in_flight{0};
int avd_calls_in_flight() { return g_calls_in_flight.load(std::memory_order_relaxed); }

// ---- contexts whose last Farneback chunk awaits its flags (fast mode; ctx->tail) ---------------------------------------------------------
// The exact re-run of a call's flagged pairs needs the HOST (it reads the flag words and sizes the launches).  With one host thread driving
// several contexts in turn, each clip's re-run would start only when the thread reaches that clip's avd_synchronize: the chip sits on one
// re-run chain at a time (a fully flagged 120-frame clip: 76 k frames/s against 108 k with one thread per context).  So a thread that has to wait
// anyway -- for its own fast pass or its own re-run -- looks at the other registered contexts of its device and settles those whose records have
// arrived.  Lock order: ctx->api_mu (blocking, taken by every entry point) -> g_tail_mu -> another context's api_mu (try_lock only).
static std::mutex g_tail_mu;
static std::vector<avd_ctx*> g_tail_list;
static std::atomic<int> g_tail_count{0};

static void tail_register(avd_ctx* ctx)
{
    if (ctx->tail_registered || hipEventRecord(ctx->tail_ev, ctx->stream) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_tail_mu);
    g_tail_list.push_back(ctx);
    ctx->tail_registered = 1;
    g_tail_count.fetch_add(1, std::memory_order_relaxed);
}

static void tail_unregister(avd_ctx* ctx)
{
    std::lock_guard<std::mutex> lk(g_tail_mu);             // a helper writes tail_registered under this lock
    if (!ctx->tail_registered) return;
    for (size_t i = 0; i < g_tail_list.size(); i++)
        if (g_tail_list[i] == ctx) { g_tail_list[i] = g_tail_list.back(); g_tail_list.pop_back(); break; }
    ctx->tail_registered = 0;
    g_tail_count.fetch_sub(1, std::memory_order_relaxed);
}

// The records of ctx's last chunk (with the level kernels' flag words) are in the pinned buffer: the pairs they mark go through the exact kernels
// (the workspace still holds the chunk), the chunk's records are assembled again and fetched.  Enqueues only; the caller holds ctx->api_mu.
static int tail_settle(avd_ctx* ctx)
{
    ctx->tail.active = 0;
    Workspace& ws = ctx->ws;
    const int p0 = ctx->tail.p0, np = ctx->tail.np, fa = ctx->tail.fa;
    const int m = rerun_flagged(ctx, &ws.h_rec[p0 + 1].reserved, (int)(sizeof(avd_frame_record) / sizeof(int)), np);
    if (m < 0) return m;
    if (m > 0) {
        launch_records(ctx, p0, np, fa, ctx->tail.clipstart);
        HIP_TRY(ctx, hipMemcpyAsync(ws.h_rec + fa, ws.d_rec + fa, sizeof(avd_frame_record) * (size_t)(p0 + np + 1 - fa), hipMemcpyDeviceToHost, ctx->stream));
        if (ctx->profiling && ctx->kmark_used > 0) kmark(ctx, AVD_K_COUNT);     // close the re-run's region
    }
    return AVD_OK;
}

static void tail_help_others(avd_ctx* self)
{
    if (g_tail_count.load(std::memory_order_relaxed) == 0) return;
    std::unique_lock<std::mutex> lk(g_tail_mu, std::try_to_lock);
    if (!lk.owns_lock()) return;
    for (size_t i = 0; i < g_tail_list.size();) {
        avd_ctx* c = g_tail_list[i];
        if (c == self || c->device != self->device || !c->api_mu.try_lock()) { i++; continue; }
        bool settled = !c->tail.active;
        if (!settled && hipEventQuery(c->tail_ev) == hipSuccess) {
            int rc;
            try { rc = tail_settle(c); } catch (...) { c->err = "out of host memory"; rc = AVD_ERR_NOMEM; }
            c->tail.active = 0;
            c->tail_rc = rc;
            settled = true;
        }
        if (settled) {
            g_tail_list[i] = g_tail_list.back(); g_tail_list.pop_back();
            c->tail_registered = 0;
            g_tail_count.fetch_sub(1, std::memory_order_relaxed);
        } else {
            i++;
        }
        c->api_mu.unlock();
    }
}

// Wait for `ev` (or, with ev == nullptr, for everything on the context's stream).  While other contexts hold unsettled tails the wait is a poll
// that settles them as their fast passes finish; otherwise it blocks in the runtime.
static hipError_t wait_helping(avd_ctx* ctx, hipEvent_t ev)
{
    int spins = 0;
    while (ctx->tail_help && g_tail_count.load(std::memory_order_relaxed) - ctx->tail_registered > 0) {
        const hipError_t e = ev ? hipEventQuery(ev) : hipStreamQuery(ctx->stream);
        if (e != hipErrorNotReady) return e;
        tail_help_others(ctx);
        if (++spins > 64) std::this_thread::yield();
    }
    return ev ? hipEventSynchronize(ev) : hipStreamSynchronize(ctx->stream);
}

// ---- entry-point bodies (wrapped by the extern "C" functions at the end of the file) -----------------
static void impl_destroy(avd_ctx* ctx);
static int impl_synchronize(avd_ctx* ctx);


static int impl_create(int device_id, avd_ctx** out)
{
    if (!out) return AVD_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count)
        return AVD_ERR_DEVICE;
    avd_ctx* ctx = new (std::nothrow) avd_ctx();
    if (!ctx) return AVD_ERR_NOMEM;
    ctx->device = device_id;
    bool ok = hipSetDevice(device_id) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&ctx->ev0) == hipSuccess && hipEventCreate(&ctx->ev1) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->tail_ev, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < 5; i++) ok = hipEventCreate(&ctx->stage_ev[i]) == hipSuccess;
    for (int i = 0; ok && i < 12; i++) ok = hipEventCreate(&ctx->kern_ev[i]) == hipSuccess;
    if (ok) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->num_cus = prop.multiProcessorCount;
        if (const char* e = std::getenv("AVD_FB_FUSED")) ctx->fb_fused = (int)std::strtol(e, nullptr, 0) & 0xF;
        if (const char* e = std::getenv("AVD_FB_FOLD_UP")) ctx->fb_fold_up = std::atoi(e) & 7;
        if (const char* e = std::getenv("AVD_FB_WIDE160")) { const int v = std::atoi(e); ctx->fb_wide160 = v == 0 ? 0 : (v == 1 ? 1 : 2); }
        if (const char* e = std::getenv("AVD_FB_FOLD_BLUR")) ctx->fb_fold_blur = std::atoi(e) != 0;
        if (const char* e = std::getenv("AVD_FB_MODE")) ctx->fb_mode = (std::strcmp(e, "exact") == 0 || std::strcmp(e, "0") == 0) ? 0 : 1;
        if (const char* e = std::getenv("AVD_FB_RERUN")) ctx->fb_rerun = std::atoi(e) != 0;
        if (const char* e = std::getenv("AVD_GEMM_WAVES")) ctx->gemm_waves = std::atoi(e) == 16 ? 16 : 8;
        build_fb_consts(ctx->fbc);
        ok = hipMalloc(&ctx->d_fbc, sizeof(FbConsts)) == hipSuccess &&
             hipMemcpy(ctx->d_fbc, &ctx->fbc, sizeof(FbConsts), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { impl_destroy(ctx); return AVD_ERR_DEVICE; }
    *out = ctx;
    return AVD_OK;
}

static void impl_destroy(avd_ctx* ctx)
{
    if (!ctx) return;
    tail_unregister(ctx);                                   // no helper finds the context from here on ...
    { std::lock_guard<std::recursive_mutex> lk(ctx->api_mu); }   // ... and one that had found it has let go
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->counted_in_flight) { ctx->counted_in_flight = 0; g_calls_in_flight.fetch_sub(1, std::memory_order_relaxed); }
    comm_destroy(ctx);
    free_ws(ctx->ws);
    if (ctx->d_fbc) (void)hipFree(ctx->d_fbc);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
    if (ctx->tail_ev) (void)hipEventDestroy(ctx->tail_ev);
    for (auto& e : ctx->stage_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->kern_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->kmark_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int impl_preprocess_bgr(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w,
                       int64_t row_stride, int64_t frame_stride,
                       uint8_t* small320, uint8_t* hash1024, int64_t* lap_sum, int64_t* lap_sumsq)
{
    if (!ctx) return AVD_ERR_ARG;
    if (!bgr && n > 0) { ctx->err = "null frame pointer"; return AVD_ERR_ARG; }
    if (int e = check_geometry(ctx, n, h, w, row_stride, frame_stride)) return e;
    if (n == 0) return AVD_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int e = avd_ws_reserve(ctx, n, h, w)) return e;
    const uint8_t* d_bgr = nullptr;
    const size_t bytes = (size_t)frame_stride * (n - 1) + (size_t)row_stride * (h - 1) + (size_t)w * 3;
    if (int e = stage_input(ctx, bgr, mem, bytes, &d_bgr)) return e;
    if (int e = launch_preprocess(ctx, d_bgr, n, h, w, row_stride, frame_stride)) return e;
    if (int e = launch_hash(ctx, n)) return e;
    Workspace& ws = ctx->ws;
    std::vector<unsigned long long> lap((size_t)n * 2);
    if (small320) HIP_TRY(ctx, hipMemcpyAsync(small320, ws.d_small, (size_t)n * AVD_NPIX, hipMemcpyDeviceToHost, ctx->stream));
    if (hash1024) HIP_TRY(ctx, hipMemcpyAsync(hash1024, ws.d_hash, (size_t)n * 1024, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(lap.data(), ws.d_lap, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int f = 0; f < n; f++) {
        if (lap_sum) lap_sum[f] = (int64_t)lap[2 * f];
        if (lap_sumsq) lap_sumsq[f] = (int64_t)lap[2 * f + 1];
    }
    ctx->last_n = n;
    ctx->rec_n = 0;
    return AVD_OK;
}

static int impl_farneback_pairs(avd_ctx* ctx, const uint8_t* small320, int mem, int n,
                        float* flow_mean, float* flow_var, float* flow_out)
{
    if (!ctx) return AVD_ERR_ARG;
    if (n < 0 || (!small320 && n > 0)) { ctx->err = "bad arguments"; return AVD_ERR_ARG; }
    if (n < 2) return AVD_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint8_t* d_small = nullptr;
    if (int e = stage_input(ctx, small320, mem, (size_t)n * AVD_NPIX, &d_small)) return e;
    ctx->last_rerun = 0;
    ctx->kmark_used = 0;
    std::vector<float> fm_tmp;
    if (!flow_mean && !flow_var && !flow_out) { fm_tmp.resize((size_t)n - 1); flow_mean = fm_tmp.data(); }   // the chunks are drained either way
    if (int e = run_flow_chunks(ctx, d_small, n, flow_mean, flow_var, flow_out, false)) return e;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->last_n = std::min(n, ctx->ws.fb_cap + 1);
    ctx->rec_n = 0;
    return AVD_OK;
}

// ---- the whole per-frame path for a BATCH of clips ------------------------------------------------------------------------
// Frames of all clips are concatenated in the per-frame buffers (small320, hashes, moments, records): clip c occupies frames
// [f0_c, f0_c + n_c).  Preprocess / hash / Hamming run per clip with that clip's geometry (cached tables); the Farneback
// stage does not care where a 320 x 320 frame came from: it runs ONCE over all N - 1 consecutive pairs of the concatenation
// (the one pair per clip boundary it computes in vain is ignored by k_records), so K short clips cost one launch sequence
// over all their pairs instead of K sequences that each leave most of the chip idle.
struct Nv12Arg {
    const uint8_t *y, *uv;
    int64_t y_row, uv_row, y_frame, uv_frame;
};

static int check_nv12(avd_ctx* ctx, const Nv12Arg& a, int n, int h, int w)
{
    if (n < 0 || h <= 0 || w <= 0 || h > 16384 || w > 16384) { ctx->err = "bad frame geometry"; return AVD_ERR_ARG; }
    if ((h | w) & 1) { ctx->err = "NV12 needs even width and height"; return AVD_ERR_UNSUPPORTED; }
    if (h < AVD_HASH || w < AVD_HASH) { ctx->err = "frame smaller than 32x32: INTER_AREA upscaling is not on the path"; return AVD_ERR_UNSUPPORTED; }
    if ((!a.y || !a.uv) && n > 0) { ctx->err = "null plane pointer"; return AVD_ERR_ARG; }
    if (a.y_row < w || a.uv_row < w || (n > 1 && (a.y_frame < a.y_row * (h - 1) + w || a.uv_frame < a.uv_row * (h / 2 - 1) + w))) {
        ctx->err = "strides smaller than the planes"; return AVD_ERR_ARG;
    }
    return 0;
}

static size_t clip_stage_bytes(const avd_clip& c, size_t* chroma_off)
{
    if (c.mem != AVD_MEM_HOST || c.n <= 0) { if (chroma_off) *chroma_off = 0; return 0; }
    if (!c.uv) {
        if (chroma_off) *chroma_off = 0;
        return ((size_t)c.frame_stride * (c.n - 1) + (size_t)c.row_stride * (c.h - 1) + (size_t)c.w * 3 + 255) / 256 * 256;
    }
    const size_t ybytes = (size_t)c.frame_stride * (c.n - 1) + (size_t)c.row_stride * (c.h - 1) + (size_t)c.w;
    const size_t cbytes = (size_t)c.uv_frame_stride * (c.n - 1) + (size_t)c.uv_row_stride * (c.h / 2 - 1) + (size_t)c.w;
    const size_t coff = (ybytes + 255) / 256 * 256;
    if (chroma_off) *chroma_off = coff;
    return (coff + cbytes + 255) / 256 * 256;
}

static int impl_analyze_batch_async(avd_ctx* ctx, const avd_clip* clips, int nclips, avd_frame_record* records)
{
    if (!ctx) return AVD_ERR_ARG;
    if (nclips < 0 || (nclips > 0 && !clips)) { ctx->err = "bad clip list"; return AVD_ERR_ARG; }
    int64_t total = 0;
    for (int c = 0; c < nclips; c++) {
        const avd_clip& k = clips[c];
        if (k.mem != AVD_MEM_HOST && k.mem != AVD_MEM_DEVICE) { ctx->err = "mem must be AVD_MEM_HOST or AVD_MEM_DEVICE"; return AVD_ERR_ARG; }
        if (!k.data && k.n > 0) { ctx->err = "null frame pointer"; return AVD_ERR_ARG; }
        if (k.uv) {
            const Nv12Arg a{k.data, k.uv, k.row_stride, k.uv_row_stride, k.frame_stride, k.uv_frame_stride};
            if (int e = check_nv12(ctx, a, k.n, k.h, k.w)) return e;
        } else if (int e = check_geometry(ctx, k.n, k.h, k.w, k.row_stride, k.frame_stride)) return e;
        total += k.n;
    }
    if (total > (1 << 24)) { ctx->err = "too many frames in one call"; return AVD_ERR_ARG; }
    if (total == 0) return AVD_OK;
    if (!records) { ctx->err = "null pointer"; return AVD_ERR_ARG; }
    const int n = (int)total;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->pending_out) { if (int e = impl_synchronize(ctx)) return e; }     // a previous call was never drained
    Workspace& ws = ctx->ws;
    // pass 1: geometry tables (cached) and sizes -- everything is reserved before the first launch of the call
    size_t rowbuf_elems = 0, lappart_elems = 0, stage_bytes = 0;
    for (int c = 0; c < nclips; c++) {
        const avd_clip& k = clips[c];
        if (k.n == 0) continue;
        if (int e = avd_ws_geometry(ctx, k.h, k.w)) return e;
        rowbuf_elems += rowbuf_elems_for(ws, k.n);
        lappart_elems += lappart_elems_for(ws, k.n);
        stage_bytes += clip_stage_bytes(k, nullptr);
    }
    if (int e = avd_ws_reserve_frames(ctx, n, rowbuf_elems, lappart_elems)) return e;
    if (int e = avd_ws_reserve_fb(ctx, n)) return e;
    if (ws.stage_bytes < stage_bytes) {
        ws.stage_bytes = 0;
        if (int e = dev_alloc(ctx, ws.d_stage, stage_bytes)) return e;
        ws.stage_bytes = stage_bytes;
    }
    // pass 2: per clip, stage (host input) -> fused full-resolution kernel -> hash / Hamming, at the clip's offsets
    ctx->kmark_used = 0;
    // profiling only: an empty launch in front of the first mark, so that the first region is the first kernel and not the queue's wake-up from idle as well
    if (ctx->profiling) hipLaunchKernelGGL(k_wake, dim3(1), dim3(64), 0, ctx->stream);
    stage_mark(ctx, 0);
    int f0 = 0;
    size_t rb = 0, lp = 0, st = 0;
    for (int c = 0; c < nclips; c++) {
        const avd_clip& k = clips[c];
        if (k.n == 0) continue;
        if (int e = avd_ws_geometry(ctx, k.h, k.w)) return e;       // a cache hit (pass 1 built it) unless > kGeomCache geometries
        ws.f0 = f0; ws.rowbuf_off = rb; ws.lappart_off = lp;
        ws.h_clipstart[f0] = 1;
        for (int i = 1; i < k.n; i++) ws.h_clipstart[f0 + i] = 0;
        size_t coff = 0;
        const size_t sb = clip_stage_bytes(k, &coff);
        const uint8_t* d_in = k.data;
        const uint8_t* d_uv = k.uv;
        kmark(ctx, AVD_K_PREPROCESS);
        if (k.mem == AVD_MEM_HOST) {
            uint8_t* dst = ws.d_stage + st;
            if (!k.uv) {
                const size_t bytes = (size_t)k.frame_stride * (k.n - 1) + (size_t)k.row_stride * (k.h - 1) + (size_t)k.w * 3;
                HIP_TRY(ctx, hipMemcpyAsync(dst, k.data, bytes, hipMemcpyHostToDevice, ctx->stream));
            } else {
                const size_t ybytes = (size_t)k.frame_stride * (k.n - 1) + (size_t)k.row_stride * (k.h - 1) + (size_t)k.w;
                const size_t cbytes = (size_t)k.uv_frame_stride * (k.n - 1) + (size_t)k.uv_row_stride * (k.h / 2 - 1) + (size_t)k.w;
                HIP_TRY(ctx, hipMemcpyAsync(dst, k.data, ybytes, hipMemcpyHostToDevice, ctx->stream));
                HIP_TRY(ctx, hipMemcpyAsync(dst + coff, k.uv, cbytes, hipMemcpyHostToDevice, ctx->stream));
                d_uv = dst + coff;
            }
            d_in = dst;
            st += sb;
        }
        if (!k.uv) {
            if (int e = launch_preprocess(ctx, d_in, k.n, k.h, k.w, k.row_stride, k.frame_stride)) return e;
        } else {
            Nv12Params nv{};
            nv.uv = d_uv; nv.uv_row_stride = k.uv_row_stride; nv.uv_frame_stride = k.uv_frame_stride;
            build_yuv_consts(nv.k);
            if (int e = launch_preprocess_nv12(ctx, d_in, nv, k.n, k.h, k.w, k.row_stride, k.frame_stride)) return e;
        }
        kmark(ctx, AVD_K_HASH);
        if (int e = launch_hash(ctx, k.n, false)) return e;
        f0 += k.n;
        rb += rowbuf_elems_for(ws, k.n);
        lp += lappart_elems_for(ws, k.n);
    }
    ws.f0 = 0; ws.rowbuf_off = 0; ws.lappart_off = 0;
    kmark(ctx, AVD_K_OTHER);
    stage_mark(ctx, 1);
    const int* clipstart = nullptr;                        // one clip: frame 0 is the only one without a predecessor
    if (nclips > 1) {
        HIP_TRY(ctx, hipMemcpyAsync(ws.d_clipstart, ws.h_clipstart, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
        clipstart = ws.d_clipstart;
    }
    stage_mark(ctx, 2);
    if (n < 2)
        hipLaunchKernelGGL(k_records, dim3(n), dim3(256), 0, ctx->stream, (const unsigned long long*)ws.d_lap, (const uint8_t*)ws.d_hash,
                           (const float*)nullptr, (const int*)nullptr, 0, ws.d_rec, 0, clipstart, 0);
    else if (int e = run_flow_chunks(ctx, ws.d_small, n, nullptr, nullptr, nullptr, true, clipstart)) return e;
    kmark(ctx, AVD_K_OTHER);
    stage_mark(ctx, 3);
    // into PINNED memory: a device-to-host copy into the caller's pageable buffer would block this thread until
    // the whole call is done and it would not be asynchronous at all; avd_synchronize hands the records over
    HIP_TRY(ctx, hipMemcpyAsync(ws.h_rec, ws.d_rec, sizeof(avd_frame_record) * n, hipMemcpyDeviceToHost, ctx->stream));
    ctx->pending_out = records; ctx->pending_n = n;
    if (ctx->tail.active) tail_register(ctx);              // from here on a waiting thread may settle this call's tail
    if (!ctx->counted_in_flight) { ctx->counted_in_flight = 1; g_calls_in_flight.fetch_add(1, std::memory_order_relaxed); }
    kmark(ctx, AVD_K_COUNT);                               // end of the last region
    stage_mark(ctx, 4);
    ctx->last_n = n;
    ctx->rec_n = n;
    return AVD_OK;
}

static int impl_analyze_frames_async(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w,
                             int64_t row_stride, int64_t frame_stride, avd_frame_record* records)
{
    if (!ctx) return AVD_ERR_ARG;
    if ((!bgr || !records) && n > 0) { ctx->err = "null pointer"; return AVD_ERR_ARG; }
    avd_clip k{};
    k.data = bgr; k.uv = nullptr; k.mem = mem; k.n = n; k.h = h; k.w = w;
    k.row_stride = row_stride; k.frame_stride = frame_stride;
    return impl_analyze_batch_async(ctx, &k, 1, records);
}

static int impl_analyze_frames_nv12_async(avd_ctx* ctx, const Nv12Arg& a, int mem, int n, int h, int w, avd_frame_record* records)
{
    if (!ctx) return AVD_ERR_ARG;
    if (!records && n > 0) { ctx->err = "null pointer"; return AVD_ERR_ARG; }
    if (int e = check_nv12(ctx, a, n, h, w)) return e;
    avd_clip k{};
    k.data = a.y; k.uv = a.uv; k.mem = mem; k.n = n; k.h = h; k.w = w;
    k.row_stride = a.y_row; k.frame_stride = a.y_frame; k.uv_row_stride = a.uv_row; k.uv_frame_stride = a.uv_frame;
    return impl_analyze_batch_async(ctx, &k, 1, records);
}

// host planes are staged back to back (the chroma plane on a 256-byte boundary); device planes are used in place
static int stage_nv12(avd_ctx* ctx, const Nv12Arg& a, int mem, int n, int h, int w, const uint8_t** d_y, Nv12Params* nv)
{
    nv->uv_row_stride = a.uv_row; nv->uv_frame_stride = a.uv_frame;
    build_yuv_consts(nv->k);
    if (mem == AVD_MEM_DEVICE) { *d_y = a.y; nv->uv = a.uv; return 0; }
    if (mem != AVD_MEM_HOST) { ctx->err = "mem must be AVD_MEM_HOST or AVD_MEM_DEVICE"; return AVD_ERR_ARG; }
    const size_t ybytes = (size_t)a.y_frame * (n - 1) + (size_t)a.y_row * (h - 1) + (size_t)w;
    const size_t cbytes = (size_t)a.uv_frame * (n - 1) + (size_t)a.uv_row * (h / 2 - 1) + (size_t)w;
    const size_t coff = (ybytes + 255) / 256 * 256;
    Workspace& ws = ctx->ws;
    if (ws.stage_bytes < coff + cbytes) {
        ws.stage_bytes = 0;
        if (int e = dev_alloc(ctx, ws.d_stage, coff + cbytes)) return e;
        ws.stage_bytes = coff + cbytes;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ws.d_stage, a.y, ybytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ws.d_stage + coff, a.uv, cbytes, hipMemcpyHostToDevice, ctx->stream));
    *d_y = ws.d_stage; nv->uv = ws.d_stage + coff;
    return 0;
}

static int impl_preprocess_nv12(avd_ctx* ctx, const Nv12Arg& a, int mem, int n, int h, int w,
                                uint8_t* small320, uint8_t* hash1024, int64_t* lap_sum, int64_t* lap_sumsq)
{
    if (!ctx) return AVD_ERR_ARG;
    if (int e = check_nv12(ctx, a, n, h, w)) return e;
    if (n == 0) return AVD_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int e = avd_ws_reserve(ctx, n, h, w)) return e;
    const uint8_t* d_y = nullptr;
    Nv12Params nv{};
    if (int e = stage_nv12(ctx, a, mem, n, h, w, &d_y, &nv)) return e;
    if (int e = launch_preprocess_nv12(ctx, d_y, nv, n, h, w, a.y_row, a.y_frame)) return e;
    if (int e = launch_hash(ctx, n)) return e;
    Workspace& ws = ctx->ws;
    std::vector<unsigned long long> lap((size_t)n * 2);
    if (small320) HIP_TRY(ctx, hipMemcpyAsync(small320, ws.d_small, (size_t)n * AVD_NPIX, hipMemcpyDeviceToHost, ctx->stream));
    if (hash1024) HIP_TRY(ctx, hipMemcpyAsync(hash1024, ws.d_hash, (size_t)n * 1024, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(lap.data(), ws.d_lap, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int f = 0; f < n; f++) {
        if (lap_sum) lap_sum[f] = (int64_t)lap[2 * f];
        if (lap_sumsq) lap_sumsq[f] = (int64_t)lap[2 * f + 1];
    }
    ctx->last_n = n;
    ctx->rec_n = 0;
    return AVD_OK;
}

static int impl_synchronize(avd_ctx* ctx)
{
    if (!ctx) return AVD_ERR_ARG;
    struct Uncount {                                        // whatever happens below, the call is no longer in flight afterwards
        avd_ctx* c;
        ~Uncount() { if (c->counted_in_flight) { c->counted_in_flight = 0; g_calls_in_flight.fetch_sub(1, std::memory_order_relaxed); } }
    } uncount{ctx};
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->tail.active) {
        // fast Farneback mode: the records of the call's last chunk carry the level kernels' flag words (tail_settle) -- unless a thread that
        // was waiting for its own context has settled this one already
        const hipError_t e = ctx->tail_registered ? wait_helping(ctx, ctx->tail_ev) : hipStreamSynchronize(ctx->stream);
        tail_unregister(ctx);
        if (e != hipSuccess) { ctx->tail.active = 0; ctx->pending_out = nullptr; ctx->pending_n = 0; ctx->err = hipGetErrorString(e); return AVD_ERR_DEVICE; }
        if (ctx->tail.active) ctx->tail_rc = tail_settle(ctx);
    }
    if (ctx->tail_rc) {                                     // the re-run could not be enqueued (by this thread or by a helper): the call has failed
        const int rc = ctx->tail_rc;
        ctx->tail_rc = 0; ctx->pending_out = nullptr; ctx->pending_n = 0;
        return rc;
    }
    HIP_TRY(ctx, wait_helping(ctx, nullptr));
    if (ctx->pending_out) {
        std::memcpy(ctx->pending_out, ctx->ws.h_rec, sizeof(avd_frame_record) * ctx->pending_n);
        ctx->last_rerun = 0;
        for (int i = 0; i < ctx->pending_n; i++) ctx->last_rerun += ctx->ws.h_rec[i].reserved != 0;
        ctx->pending_out = nullptr; ctx->pending_n = 0;
    }
    if (ctx->profiling && ctx->kmark_used > 1) {
        for (float& v : ctx->kernel_ms) v = 0.f;
        for (int i = 0; i + 1 < ctx->kmark_used; i++) {
            float ms = 0.f;
            if (ctx->kmark_id[i] < AVD_K_COUNT && hipEventElapsedTime(&ms, ctx->kmark_ev[i], ctx->kmark_ev[i + 1]) == hipSuccess)
                ctx->kernel_ms[ctx->kmark_id[i]] += ms;
        }
        ctx->kmark_incomplete = ctx->kmark_overflow;
    }
    ctx->kmark_used = 0;
    ctx->kmark_overflow = 0;
    if (ctx->profiling && ctx->stage_marks == 5) {
        // stages: 0 preprocess, 1 hash+hamming+records, 2 farneback+stats (3 reported as copy-out)
        for (int i = 0; i < 4; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->stage_ev[i], ctx->stage_ev[i + 1]) == hipSuccess) ctx->stage_ms[i] = ms;
        }
        // 4 / 5: mean duration of one k_uv<320> / k_hscan<320> launch (first segment of the last chunk)
        float sum[2] = {0.f, 0.f}; int cnt[2] = {0, 0};
        for (int i = 0; i + 1 < ctx->kern_ev_used; i += 2) {     // recorded by blur_iteration<320> of the drained call
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->kern_ev[i], ctx->kern_ev[i + 1]) == hipSuccess) { sum[(i >> 1) & 1] += ms; cnt[(i >> 1) & 1]++; }
        }
        for (int k = 0; k < 2; k++) ctx->stage_ms[4 + k] = cnt[k] ? sum[k] / cnt[k] : 0.f;
        ctx->stage_marks = 0;                       // the events belong to the call that was just drained
    }
    return AVD_OK;
}

static int impl_analyze_frames(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w,
                       int64_t row_stride, int64_t frame_stride, avd_frame_record* records)
{
    int rc = impl_analyze_frames_async(ctx, bgr, mem, n, h, w, row_stride, frame_stride, records);
    if (rc) return rc;
    return impl_synchronize(ctx);
}

static int impl_timer_start(avd_ctx* ctx)
{
    if (!ctx) return AVD_ERR_ARG;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return AVD_OK;
}

static int impl_timer_stop(avd_ctx* ctx, float* elapsed_ms)
{
    if (!ctx || !elapsed_ms) return AVD_ERR_ARG;
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return AVD_OK;
}

static int impl_set_profiling(avd_ctx* ctx, int enable)
{
    if (!ctx) return AVD_ERR_ARG;
    ctx->profiling = enable != 0;
    return AVD_OK;
}

static int impl_stage_ms(avd_ctx* ctx, int stage, float* ms)
{
    if (!ctx || !ms || stage < 0 || stage > 5) return AVD_ERR_ARG;
    *ms = ctx->stage_ms[stage];
    return AVD_OK;
}

static int impl_kernel_ms(avd_ctx* ctx, int id, float* ms)
{
    if (!ctx || !ms || id < 0 || id >= AVD_K_COUNT) return AVD_ERR_ARG;
    if (ctx->kmark_incomplete) { ctx->err = "avd_kernel_ms: the profiled call had more kernel regions than the library records (96): split it"; return AVD_ERR_ARG; }
    *ms = ctx->kernel_ms[id];
    return AVD_OK;
}

static int64_t impl_debug_fetch(avd_ctx* ctx, const char* name, void* out, size_t out_bytes)
{
    if (!ctx || !name || !out) return AVD_ERR_ARG;
    Workspace& ws = ctx->ws;
    const int n = ctx->last_n;
    const void* src = nullptr;
    size_t bytes = 0;
    auto level = [&](const char* prefix) -> int {
        const size_t L = std::strlen(prefix);
        if (std::strncmp(name, prefix, L) == 0 && name[L] >= '0' && name[L] < '0' + AVD_FB_LEVELS && name[L + 1] == 0)
            return name[L] - '0';
        return -1;
    };
    int k;
    if (std::strcmp(name, "area") == 0) { src = ws.d_area; bytes = (size_t)n * 1024; }
    else if (std::strcmp(name, "small") == 0) { src = ws.d_small; bytes = (size_t)n * AVD_NPIX; }
    // the Farneback scratch holds ONE chunk (kFbChunk pairs): for longer clips these are the last chunk's buffers
    else if ((k = level("pyr")) >= 0 && k == 0 && ctx->fb_fold_blur) { ctx->err = "pyr0 does not exist while fb_fold_blur is on (the polynomial expansion forms the 320-px blur itself)"; return AVD_ERR_ARG; }
    else if ((k = level("pyr")) >= 0) { src = ws.d_pyr[k]; bytes = (size_t)std::min(n, ws.fb_cap + 1) * (AVD_NPIX >> (2 * k)) * 4; }
    else if ((k = level("poly")) >= 0) { src = ws.d_poly[k]; bytes = (size_t)std::min(n, ws.fb_cap + 1) * 5 * (AVD_NPIX >> (2 * k)) * 4; }
    else if ((k = level("flow")) >= 0) { src = ws.flow_res[k] ? ws.flow_res[k] : ws.d_flow[k]; bytes = (size_t)std::min(std::max(n - 1, 0), ws.fb_cap) * 2 * (AVD_NPIX >> (2 * k)) * 4; }
    else if (std::strcmp(name, "vs0") == 0) { src = ws.d_vs0; bytes = (size_t)ws.fb_cap * 5 * AVD_SMALL * 8 * 8; }
    else { ctx->err = "unknown debug buffer"; return AVD_ERR_ARG; }
    if (!src) { ctx->err = "buffer not allocated yet"; return AVD_ERR_ARG; }
    bytes = std::min(bytes, out_bytes);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
    return (int64_t)bytes;
}

// Wait for work the caller enqueued on ANOTHER stream (e.g. torch's current stream, which produced or is still
// producing a device input) before anything submitted to this context afterwards: an event on the producer stream,
// waited for by the context's stream.  A null handle names the legacy default stream.
static int impl_wait_stream(avd_ctx* ctx, void* producer)
{
    if (!ctx) return AVD_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_in, (hipStream_t)producer));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_in, 0));
    return AVD_OK;
}

// Give the scratch memory back (a service keeps idle contexts cheap); the next call re-reserves it.
static int impl_release_workspace(avd_ctx* ctx)
{
    if (!ctx) return AVD_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int e = impl_synchronize(ctx)) return e;
    free_ws(ctx->ws, true);
    ctx->rec_n = 0;
    return AVD_OK;
}

// Tuning / test switches.  "fb_fused": bit k set = pyramid level k (0 = 320x320) runs the fused level kernel.
static int impl_set_option(avd_ctx* ctx, const char* name, int value)
{
    if (!ctx || !name) return AVD_ERR_ARG;
    if (std::strcmp(name, "fb_fused") == 0) { ctx->fb_fused = value & 0xF; return AVD_OK; }
    if (std::strcmp(name, "fb_mode") == 0) { ctx->fb_mode = value ? 1 : 0; return AVD_OK; }
    if (std::strcmp(name, "fb_fold_up") == 0) { ctx->fb_fold_up = value & 7; return AVD_OK; }
    if (std::strcmp(name, "fb_rerun") == 0) { ctx->fb_rerun = value ? 1 : 0; return AVD_OK; }
    if (std::strcmp(name, "fb_rerun_fused") == 0) { ctx->fb_rerun_fused = (value & 0xF) | 8; return AVD_OK; }
    if (std::strcmp(name, "tail_help") == 0) { ctx->tail_help = value != 0; return AVD_OK; }
    if (std::strcmp(name, "fb_wide160") == 0) { ctx->fb_wide160 = value == 0 ? 0 : (value == 1 ? 1 : 2); return AVD_OK; }
    if (std::strcmp(name, "fb_fold_blur") == 0) { ctx->fb_fold_blur = value != 0; return AVD_OK; }
    if (std::strcmp(name, "gemm_waves") == 0) { ctx->gemm_waves = value == 16 ? 16 : 8; return AVD_OK; }
    if (std::strcmp(name, "cnn_tiles") == 0) { ctx->cnn_tiles = value; return AVD_OK; }
    if (std::strcmp(name, "cnn_fuse") == 0) { ctx->cnn_fuse = value < 0 ? 0 : (value > 2 ? 2 : value); return AVD_OK; }
    if (std::strcmp(name, "cnn_chunk") == 0) {
        if (value < 1 || value > 1024) { ctx->err = "cnn_chunk: 1 ... 1024 frames per forward pass"; return AVD_ERR_ARG; }
        ctx->cnn_chunk = value;
        return AVD_OK;
    }
    ctx->err = std::string("unknown option: ") + name;
    return AVD_ERR_ARG;
}

// The value an option has NOW (environment defaults included), and read-only counters: "rerun_pairs" = pairs of the last
// drained call that the fast level kernel flagged as ill-posed and the exact kernels re-ran.
static int impl_get_option(avd_ctx* ctx, const char* name, int* value)
{
    if (!ctx || !name || !value) return AVD_ERR_ARG;
    if (std::strcmp(name, "fb_fused") == 0) { *value = ctx->fb_fused; return AVD_OK; }
    if (std::strcmp(name, "fb_mode") == 0) { *value = ctx->fb_mode; return AVD_OK; }
    if (std::strcmp(name, "fb_fold_up") == 0) { *value = ctx->fb_fold_up; return AVD_OK; }
    if (std::strcmp(name, "fb_rerun") == 0) { *value = ctx->fb_rerun; return AVD_OK; }
    if (std::strcmp(name, "fb_rerun_fused") == 0) { *value = ctx->fb_rerun_fused; return AVD_OK; }
    if (std::strcmp(name, "tail_help") == 0) { *value = ctx->tail_help; return AVD_OK; }
    if (std::strcmp(name, "fb_wide160") == 0) { *value = ctx->fb_wide160; return AVD_OK; }
    if (std::strcmp(name, "fb_wide160_used") == 0) { *value = ctx->fb_wide160_used; return AVD_OK; }
    if (std::strcmp(name, "fb_fold_blur") == 0) { *value = ctx->fb_fold_blur; return AVD_OK; }
    if (std::strcmp(name, "gemm_waves") == 0) { *value = ctx->gemm_waves; return AVD_OK; }
    if (std::strcmp(name, "cnn_tiles") == 0) { *value = ctx->cnn_tiles; return AVD_OK; }
    if (std::strcmp(name, "cnn_fuse") == 0) { *value = ctx->cnn_fuse; return AVD_OK; }
    if (std::strcmp(name, "cnn_chunk") == 0) { *value = ctx->cnn_chunk; return AVD_OK; }
    if (std::strcmp(name, "rerun_pairs") == 0) { *value = ctx->last_rerun; return AVD_OK; }
    ctx->err = std::string("unknown option: ") + name;
    return AVD_ERR_ARG;
}

// ---- CNN extension (avd_cnn.hip; never part of ai_score) ---------------------------------------------------------
static int impl_cnn_set_weights(avd_ctx* ctx, const uint16_t* w, size_t n_w, const float* b, size_t n_b)
{
    if (!ctx || !w || !b) return AVD_ERR_ARG;
    size_t want_w = 0, want_b = 0;
    cnn_param_counts(&want_w, &want_b);
    if (n_w != want_w || n_b != want_b) { ctx->err = "avd_cnn_set_weights: parameter counts differ from avd_cnn_param_counts"; return AVD_ERR_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return cnn_set_weights(ctx, w, b);
}

static int impl_cnn_forward(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w, int64_t row_stride, int64_t frame_stride,
                            float* logits, int reps, float* forward_ms)
{
    if (!ctx) return AVD_ERR_ARG;
    if ((!bgr || !logits) && n > 0) { ctx->err = "null pointer"; return AVD_ERR_ARG; }
    if (n < 0 || h < 2 || w < 2 || row_stride < (int64_t)w * 3) { ctx->err = "bad frame geometry"; return AVD_ERR_ARG; }
    if (n == 0) return AVD_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Workspace& ws = ctx->ws;
    if (!ws.d_cnn_w) { ctx->err = "avd_cnn_set_weights has not been called"; return AVD_ERR_ARG; }
    // frames per forward pass: bounds the activation scratch (4 x 1.6 MB per frame); avd_set_option "cnn_chunk", 1 ... 1024
    // (32-bit byte offsets inside an activation); the late stages fill the chip better with more frames per pass
    const int kChunk = ctx->cnn_chunk;
    if (int e = cnn_reserve(ctx, std::min(n, kChunk))) return e;
    const uint8_t* d_bgr = nullptr;
    const size_t bytes = (size_t)frame_stride * (n - 1) + (size_t)row_stride * (h - 1) + (size_t)w * 3;
    if (int e = stage_input(ctx, bgr, mem, bytes, &d_bgr)) return e;
    float total_ms = 0.f;
    for (int f0 = 0; f0 < n; f0 += kChunk) {
        const int m = std::min(kChunk, n - f0);
        const uint8_t* src = d_bgr + (size_t)frame_stride * f0;
        if (int e = launch_cnn_forward(ctx, src, m, h, w, row_stride, frame_stride)) return e;
        if (reps > 0 && forward_ms) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
            for (int r = 0; r < reps; r++)
                if (int e = launch_cnn_forward(ctx, src, m, h, w, row_stride, frame_stride)) return e;
            HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
            HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
            float ms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            total_ms += ms / reps;
        }
        // the logits buffer is reused by the next chunk: the copy is ordered before it on the same stream
        HIP_TRY(ctx, hipMemcpyAsync(logits + (size_t)f0 * 1000, ws.d_cnn_logits, sizeof(float) * (size_t)m * 1000, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (reps > 0 && forward_ms) *forward_ms = total_ms;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AVD_OK;
}

// ---- ViT-B/16 patch embedding (extension; never part of ai_score) ---------------------------------------------
static int impl_vit_set_weights(avd_ctx* ctx, const uint16_t* w_bf16, const float* bias)
{
    if (!ctx || !w_bf16) return AVD_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Workspace& ws = ctx->ws;
    if (!ws.d_vit_w) if (int e = dev_alloc(ctx, ws.d_vit_w, (size_t)768 * 768)) return e;
    if (!ws.d_vit_bias) if (int e = dev_alloc(ctx, ws.d_vit_bias, (size_t)768)) return e;
    // the GEMM reads its operands in 1-KiB blocks (avd_vit.hip): re-tile the row-major weight once, here
    std::vector<uint16_t> blocked((size_t)768 * 768);
    gemm_block_operand(w_bf16, blocked.data(), 768, 768);
    HIP_TRY(ctx, hipMemcpyAsync(ws.d_vit_w, blocked.data(), sizeof(uint16_t) * 768 * 768, hipMemcpyHostToDevice, ctx->stream));
    if (bias) HIP_TRY(ctx, hipMemcpyAsync(ws.d_vit_bias, bias, sizeof(float) * 768, hipMemcpyHostToDevice, ctx->stream));
    ws.vit_has_bias = bias != nullptr;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AVD_OK;
}

static int impl_vit_patch_embed(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w, int64_t row_stride,
                                int64_t frame_stride, void* tokens, int tokens_mem, int tokens_bf16, int reps, float* gemm_ms)
{
    if (!ctx) return AVD_ERR_ARG;
    if ((!bgr || !tokens) && n > 0) { ctx->err = "null pointer"; return AVD_ERR_ARG; }
    if (n < 0 || h < 2 || w < 2 || row_stride < (int64_t)w * 3) { ctx->err = "bad frame geometry"; return AVD_ERR_ARG; }
    if (tokens_mem != AVD_MEM_HOST && tokens_mem != AVD_MEM_DEVICE) { ctx->err = "tokens_mem must be AVD_MEM_HOST or AVD_MEM_DEVICE"; return AVD_ERR_ARG; }
    if (n == 0) return AVD_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Workspace& ws = ctx->ws;
    if (!ws.d_vit_w) { ctx->err = "avd_vit_set_weights has not been called"; return AVD_ERR_ARG; }
    const size_t m = (size_t)n * 196;
    const size_t m_pad = (m + kGemmRowPad - 1) / kGemmRowPad * kGemmRowPad;   // the persistent GEMM reads whole 256-row tiles of A
    if (ws.vit_patch_elems < m_pad * 768) {
        ws.vit_patch_elems = 0;                               // not valid again until the buffer exists
        if (int e = dev_alloc(ctx, ws.d_vit_patches, m_pad * 768)) return e;
        HIP_TRY(ctx, hipMemsetAsync(ws.d_vit_patches, 0, m_pad * 768 * sizeof(uint16_t), ctx->stream));
        ws.vit_patch_elems = m_pad * 768;
    }
    void* d_tok = tokens;
    const size_t esz = tokens_bf16 ? sizeof(uint16_t) : sizeof(float);
    if (tokens_mem == AVD_MEM_HOST) {
        if (ws.vit_token_elems < m * 768) {
            ws.vit_token_elems = 0;
            if (int e = dev_alloc(ctx, ws.d_vit_tokens, m * 768)) return e;
            ws.vit_token_elems = m * 768;
        }
        d_tok = ws.d_vit_tokens;
    }
    const uint8_t* d_bgr = nullptr;
    const size_t bytes = (size_t)frame_stride * (n - 1) + (size_t)row_stride * (h - 1) + (size_t)w * 3;
    if (int e = stage_input(ctx, bgr, mem, bytes, &d_bgr)) return e;
    const float* d_bias = ws.vit_has_bias ? ws.d_vit_bias : nullptr;
    if (int e = launch_vit_patch_embed(ctx, d_bgr, n, h, w, row_stride, frame_stride, ws.d_vit_w, d_bias, d_tok, tokens_bf16, ws.d_vit_patches)) return e;
    if (reps > 0 && gemm_ms) {
        // the GEMM alone, `reps` launches between two events on the context's stream (the patches stay resident)
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        for (int r = 0; r < reps; r++)
            if (int e = launch_gemm_bf16_nt(ctx, ws.d_vit_patches, ws.d_vit_w, d_bias, d_tok, tokens_bf16, (int)m, 768, 768)) return e;
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        *gemm_ms = ms / reps;
    }
    if (tokens_mem == AVD_MEM_HOST)
        HIP_TRY(ctx, hipMemcpyAsync(tokens, d_tok, esz * m * 768, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AVD_OK;
}

// ---- LayerNorm / softmax (extensions; never part of ai_score) ------------------------------------------------------------
// x / y: host or device (both the same side), gamma / beta: host float[cols].  Host operands are staged through the
// extension scratch (d_vit_tokens, grown on demand).
static int impl_rowop(avd_ctx* ctx, int op, const void* x, int mem, int bf16, int64_t rows, int cols, const float* gamma, const float* beta,
                      float eps, void* y, int reps, float* ms)
{
    if (!ctx) return AVD_ERR_ARG;
    if (rows < 0 || cols <= 0 || (rows > 0 && (!x || !y)) || (op == 0 && (!gamma || !beta))) { ctx->err = "bad arguments"; return AVD_ERR_ARG; }
    if (mem != AVD_MEM_HOST && mem != AVD_MEM_DEVICE) { ctx->err = "mem must be AVD_MEM_HOST or AVD_MEM_DEVICE"; return AVD_ERR_ARG; }
    if (rows == 0) return AVD_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Workspace& ws = ctx->ws;
    const size_t esz = (op == 0 && bf16) ? 2 : 4, bytes = (size_t)rows * cols * esz;
    // layout in float units: gamma | beta | pad to a multiple of 64 floats | x (rounded up to 256 bytes) | y -- the size is computed from
    // the SAME expressions the pointers below use (round 3 reserved 2 * bytes + 8 * cols + 256 B, less than pad + round-up can need)
    const size_t gb_f = 2 * (size_t)cols + ((64 - (2 * cols) % 64) % 64);
    const size_t x_bytes = (bytes + 255) / 256 * 256;
    const bool host = mem == AVD_MEM_HOST;
    const size_t want = host ? gb_f + (x_bytes + bytes + 3) / 4 : gb_f;
    if (ws.vit_token_elems < want) {
        ws.vit_token_elems = 0;
        if (int e = dev_alloc(ctx, ws.d_vit_tokens, want)) return e;
        ws.vit_token_elems = want;
    }
    float* d_gb = ws.d_vit_tokens;                        // gamma | beta first (16-byte aligned), then x, then y
    char* d_x = (char*)x;
    char* d_y = (char*)y;
    if (op == 0) {
        HIP_TRY(ctx, hipMemcpyAsync(d_gb, gamma, sizeof(float) * cols, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_gb + cols, beta, sizeof(float) * cols, hipMemcpyHostToDevice, ctx->stream));
    }
    if (host) {
        d_x = (char*)(d_gb + gb_f);
        d_y = d_x + x_bytes;
        HIP_TRY(ctx, hipMemcpyAsync(d_x, x, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    auto run = [&]() -> int {
        return op == 0 ? launch_layernorm(ctx, d_x, d_y, bf16, rows, cols, d_gb, d_gb + cols, eps)
                       : launch_softmax(ctx, (const float*)d_x, (float*)d_y, rows, cols);
    };
    if (int e = run()) return e;
    if (reps > 0 && ms) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        for (int r = 0; r < reps; r++)
            if (int e = run()) return e;
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
        float t = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
        *ms = t / reps;
    }
    if (host) HIP_TRY(ctx, hipMemcpyAsync(y, d_y, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AVD_OK;
}

// ---- audio analyzer (row N3) -------------------------------------------------------------------------------------
static int impl_audio_features(avd_ctx* ctx, const float* wav, int mem, int64_t n, int win, avd_audio_window* windows, int max_windows)
{
    if (!ctx) return AVD_ERR_ARG;
    if (n < 0 || win < 1 || (n > 0 && (!wav || !windows))) { ctx->err = "bad arguments"; return AVD_ERR_ARG; }
    if (n == 0) return AVD_OK;
    const int64_t nw64 = (n + win - 1) / win;
    if (nw64 > max_windows || nw64 > (1 << 24)) { ctx->err = "windows array too small"; return AVD_ERR_ARG; }
    const int nwin = (int)nw64;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Workspace& ws = ctx->ws;
    const uint8_t* d_wav = nullptr;
    if (int e = stage_input(ctx, reinterpret_cast<const uint8_t*>(wav), mem, (size_t)n * sizeof(float), &d_wav)) return e;
    if (ws.audio_out_elems < (size_t)nwin) {
        ws.audio_out_elems = 0;
        if (int e = dev_alloc(ctx, ws.d_audio_out, (size_t)nwin)) return e;
        ws.audio_out_elems = (size_t)nwin;
    }
    if (int e = launch_audio_features(ctx, reinterpret_cast<const float*>(d_wav), n, win, ws.d_audio_out, nwin)) return e;
    HIP_TRY(ctx, hipMemcpyAsync(windows, ws.d_audio_out, sizeof(avd_audio_window) * nwin, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AVD_OK;
}

// Nothing may propagate across the C boundary: std::vector / std::string members of the context and the table
// builders can throw std::bad_alloc (or length_error), so every entry point runs inside this guard.
template <typename F>
static int guarded(avd_ctx* ctx, F&& f) noexcept
{
    try {
        if (!ctx) return f();
        std::lock_guard<std::recursive_mutex> lk(ctx->api_mu);     // one call at a time per context; helpers of other contexts only try_lock (tail_help_others)
        return f();
    } catch (const std::bad_alloc&) {
        if (ctx) { try { ctx->err = "out of host memory"; } catch (...) {} }
        return AVD_ERR_NOMEM;
    } catch (...) {
        if (ctx) { try { ctx->err = "unexpected C++ exception"; } catch (...) {} }
        return AVD_ERR_DEVICE;
    }
}

// ---- C-ABI ------------------------------------------------------------------------------
extern "C" {

int avd_abi_version(void) { return AVD_ABI_VERSION; }

const char* avd_last_error(const avd_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int avd_create(int device_id, avd_ctx** out)
{
    return guarded(nullptr, [&] { return impl_create(device_id, out); });
}

void avd_destroy(avd_ctx* ctx)
{
    (void)guarded(nullptr, [&] { impl_destroy(ctx); return 0; });
}

int avd_preprocess_bgr(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w, int64_t row_stride,
                       int64_t frame_stride, uint8_t* small320, uint8_t* hash1024, int64_t* lap_sum, int64_t* lap_sumsq)
{
    return guarded(ctx, [&] { return impl_preprocess_bgr(ctx, bgr, mem, n, h, w, row_stride, frame_stride, small320, hash1024, lap_sum, lap_sumsq); });
}

int avd_farneback_pairs(avd_ctx* ctx, const uint8_t* small320, int mem, int n, float* flow_mean, float* flow_var, float* flow_out)
{
    return guarded(ctx, [&] { return impl_farneback_pairs(ctx, small320, mem, n, flow_mean, flow_var, flow_out); });
}

int avd_analyze_frames_async(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w, int64_t row_stride,
                             int64_t frame_stride, avd_frame_record* records)
{
    return guarded(ctx, [&] { return impl_analyze_frames_async(ctx, bgr, mem, n, h, w, row_stride, frame_stride, records); });
}

int avd_synchronize(avd_ctx* ctx) { return guarded(ctx, [&] { return impl_synchronize(ctx); }); }

int avd_analyze_batch_async(avd_ctx* ctx, const avd_clip* clips, int nclips, avd_frame_record* records)
{
    return guarded(ctx, [&] { return impl_analyze_batch_async(ctx, clips, nclips, records); });
}

int avd_analyze_batch(avd_ctx* ctx, const avd_clip* clips, int nclips, avd_frame_record* records)
{
    return guarded(ctx, [&] {
        const int rc = impl_analyze_batch_async(ctx, clips, nclips, records);
        return rc ? rc : impl_synchronize(ctx);
    });
}

int avd_preprocess_nv12(avd_ctx* ctx, const uint8_t* y, const uint8_t* uv, int mem, int n, int h, int w, int64_t y_row_stride,
                        int64_t uv_row_stride, int64_t y_frame_stride, int64_t uv_frame_stride, uint8_t* small320,
                        uint8_t* hash1024, int64_t* lap_sum, int64_t* lap_sumsq)
{
    const Nv12Arg a{y, uv, y_row_stride, uv_row_stride, y_frame_stride, uv_frame_stride};
    return guarded(ctx, [&] { return impl_preprocess_nv12(ctx, a, mem, n, h, w, small320, hash1024, lap_sum, lap_sumsq); });
}

int avd_analyze_frames_nv12_async(avd_ctx* ctx, const uint8_t* y, const uint8_t* uv, int mem, int n, int h, int w,
                                  int64_t y_row_stride, int64_t uv_row_stride, int64_t y_frame_stride,
                                  int64_t uv_frame_stride, avd_frame_record* records)
{
    const Nv12Arg a{y, uv, y_row_stride, uv_row_stride, y_frame_stride, uv_frame_stride};
    return guarded(ctx, [&] { return impl_analyze_frames_nv12_async(ctx, a, mem, n, h, w, records); });
}

int avd_analyze_frames_nv12(avd_ctx* ctx, const uint8_t* y, const uint8_t* uv, int mem, int n, int h, int w,
                            int64_t y_row_stride, int64_t uv_row_stride, int64_t y_frame_stride, int64_t uv_frame_stride,
                            avd_frame_record* records)
{
    const Nv12Arg a{y, uv, y_row_stride, uv_row_stride, y_frame_stride, uv_frame_stride};
    return guarded(ctx, [&] {
        const int rc = impl_analyze_frames_nv12_async(ctx, a, mem, n, h, w, records);
        return rc ? rc : impl_synchronize(ctx);
    });
}

int avd_analyze_frames(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w, int64_t row_stride,
                       int64_t frame_stride, avd_frame_record* records)
{
    return guarded(ctx, [&] { return impl_analyze_frames(ctx, bgr, mem, n, h, w, row_stride, frame_stride, records); });
}

int avd_cnn_param_counts(size_t* n_weights, size_t* n_biases)
{
    if (!n_weights || !n_biases) return AVD_ERR_ARG;
    cnn_param_counts(n_weights, n_biases);
    return AVD_OK;
}

int avd_cnn_set_weights(avd_ctx* ctx, const uint16_t* weights_bf16, size_t n_weights, const float* biases, size_t n_biases)
{
    return guarded(ctx, [&] { return impl_cnn_set_weights(ctx, weights_bf16, n_weights, biases, n_biases); });
}

int avd_cnn_forward(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w, int64_t row_stride, int64_t frame_stride,
                    float* logits, int timing_reps, float* forward_ms)
{
    return guarded(ctx, [&] { return impl_cnn_forward(ctx, bgr, mem, n, h, w, row_stride, frame_stride, logits, timing_reps, forward_ms); });
}

int avd_cnn_conv(avd_ctx* ctx, const uint16_t* x, int n, int hin, int win, int cin, const uint16_t* w, const float* bias,
                 int cout, int ksize, int stride, int relu, const uint16_t* residual, uint16_t* y)
{
    return guarded(ctx, [&] {
        if (!ctx || !x || !w || !bias || !y) return (int)AVD_ERR_ARG;
        if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return (int)AVD_ERR_DEVICE; }
        return cnn_conv_host(ctx, x, n, hin, win, cin, w, bias, cout, ksize, stride, relu, residual, y);
    });
}

int avd_vit_set_weights(avd_ctx* ctx, const uint16_t* weight_bf16, const float* bias)
{
    return guarded(ctx, [&] { return impl_vit_set_weights(ctx, weight_bf16, bias); });
}

int avd_vit_patch_embed(avd_ctx* ctx, const uint8_t* bgr, int mem, int n, int h, int w, int64_t row_stride, int64_t frame_stride,
                        void* tokens, int tokens_mem, int tokens_bf16, int timing_reps, float* gemm_ms)
{
    return guarded(ctx, [&] { return impl_vit_patch_embed(ctx, bgr, mem, n, h, w, row_stride, frame_stride, tokens, tokens_mem, tokens_bf16, timing_reps, gemm_ms); });
}

int avd_layernorm(avd_ctx* ctx, const void* x, int mem, int bf16, int64_t rows, int cols, const float* gamma, const float* beta, float eps,
                  void* y, int timing_reps, float* ms)
{
    return guarded(ctx, [&] { return impl_rowop(ctx, 0, x, mem, bf16, rows, cols, gamma, beta, eps, y, timing_reps, ms); });
}

int avd_softmax(avd_ctx* ctx, const float* x, int mem, int64_t rows, int cols, float* y, int timing_reps, float* ms)
{
    return guarded(ctx, [&] { return impl_rowop(ctx, 1, x, mem, 0, rows, cols, nullptr, nullptr, 0.f, y, timing_reps, ms); });
}

int avd_audio_features(avd_ctx* ctx, const float* wav, int mem, int64_t n, int win, avd_audio_window* windows, int max_windows)
{
    return guarded(ctx, [&] { return impl_audio_features(ctx, wav, mem, n, win, windows, max_windows); });
}

int avd_comm_unique_id(void* id128)
{
    if (!id128) return AVD_ERR_ARG;
    return guarded(nullptr, [&] { std::string err; return comm_unique_id(err, id128); });
}
int avd_comm_init(avd_ctx* ctx, int rank, int world, const void* id128)
{
    if (!ctx) return AVD_ERR_ARG;
    return guarded(ctx, [&] { return comm_init(ctx, rank, world, id128); });
}
int avd_allgather_records(avd_ctx* ctx, const avd_frame_record* local, int count, avd_frame_record* all)
{
    if (!ctx) return AVD_ERR_ARG;
    return guarded(ctx, [&] { return comm_allgather_records(ctx, local, count, all); });
}

int avd_allgather_last_records(avd_ctx* ctx, int count, avd_frame_record* all)
{
    if (!ctx) return AVD_ERR_ARG;
    return guarded(ctx, [&] {
        // a pending asynchronous call is drained FIRST: the exact re-run of the pairs its fast level kernels flagged happens there, and the
        // records on the device are final only after it
        if (ctx->pending_out || ctx->tail.active) { if (int e = impl_synchronize(ctx)) return e; }
        return comm_allgather_last_records(ctx, count, all);
    });
}

int avd_wait_stream(avd_ctx* ctx, void* producer_stream) { return guarded(ctx, [&] { return impl_wait_stream(ctx, producer_stream); }); }
int avd_release_workspace(avd_ctx* ctx) { return guarded(ctx, [&] { return impl_release_workspace(ctx); }); }
int avd_timer_start(avd_ctx* ctx) { return guarded(ctx, [&] { return impl_timer_start(ctx); }); }
int avd_timer_stop(avd_ctx* ctx, float* elapsed_ms) { return guarded(ctx, [&] { return impl_timer_stop(ctx, elapsed_ms); }); }
int avd_set_option(avd_ctx* ctx, const char* name, int value) { return guarded(ctx, [&] { return impl_set_option(ctx, name, value); }); }
int avd_get_option(avd_ctx* ctx, const char* name, int* value) { return guarded(ctx, [&] { return impl_get_option(ctx, name, value); }); }
int avd_set_profiling(avd_ctx* ctx, int enable) { return guarded(ctx, [&] { return impl_set_profiling(ctx, enable); }); }
int avd_stage_ms(avd_ctx* ctx, int stage, float* ms) { return guarded(ctx, [&] { return impl_stage_ms(ctx, stage, ms); }); }
int avd_kernel_ms(avd_ctx* ctx, int kernel_id, float* ms) { return guarded(ctx, [&] { return impl_kernel_ms(ctx, kernel_id, ms); }); }

int64_t avd_debug_fetch(avd_ctx* ctx, const char* name, void* out, size_t out_bytes)
{
    int64_t got = AVD_ERR_DEVICE;
    const int rc = guarded(ctx, [&] { got = impl_debug_fetch(ctx, name, out, out_bytes); return 0; });
    return rc ? rc : got;
}

}  // extern "C"
