"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs.  Integer / byte outputs AND the float Farneback chain are compared BIT-EXACT
(the kernels reproduce cv2's operation order); the end-to-end ai_susp /
summary comparison additionally states the north_star tolerance (1e-4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from avd_hip import synth  # noqa: E402


GEOMS = [
    (3, 48, 64),        # tiny, vector path (w % 16 == 0)
    (2, 67, 101),       # odd sizes -> scalar load path, ragged quads, general area tables
    (2, 360, 640),      # 1/9 of 720p... integer x scale 20, fractional y scale 11.25
    (2, 720, 1280),     # cfg1 geometry
    (2, 1080, 1920),    # cfg2 geometry
    (1, 2160, 3840),    # cfg4 geometry
    (2, 640, 640),      # exact 2x2 decimation to 320 -> INTER_LINEAR rerouted to INTER_AREA fast
    (2, 64, 64),        # 2x2 area-fast for the hash as well
    (2, 96, 128),       # integer scales 3x4 -> area-fast general
    (2, 240, 426),      # upscaling in y for the 320x320 resize
]


@pytest.mark.parametrize("n,h,w", GEOMS)
def test_preprocess_bit_exact(ctx, oracle, n, h, w):
    frames = synth.random_frames(n, h, w, seed=h * 7 + w)
    small, hsh, s, q = ctx.preprocess_bgr(frames)
    o_small, o_hsh, o_s, o_q = oracle.preprocess_bgr(frames)
    area = ctx.debug_fetch("area", (n, 32, 32), np.uint8)
    o_area = np.stack([oracle.resize_area(oracle.bgr2gray(f), 32, 32) for f in frames])
    assert np.array_equal(area, o_area), f"area cells differ: {np.argwhere(area != o_area)[:5]}"
    assert np.array_equal(s, o_s) and np.array_equal(q, o_q)
    assert np.array_equal(small, o_small), f"{np.count_nonzero(small != o_small)} small320 pixels differ"
    assert np.array_equal(hsh, o_hsh)


def test_preprocess_random_geometries(ctx, oracle):
    """Seeded sweep over arbitrary (mostly odd) frame sizes: general INTER_AREA tables with partial cells,
    scalar load path, last band of any height."""
    rng = np.random.default_rng(99)
    for _ in range(12):
        h, w = int(rng.integers(32, 400)), int(rng.integers(32, 700))
        frames = synth.random_frames(2, h, w, seed=h * 1000 + w)
        got = ctx.preprocess_bgr(frames)
        want = oracle.preprocess_bgr(frames)
        for name, x, y in zip(("small", "hash", "lap_sum", "lap_sumsq"), got, want):
            assert np.array_equal(x, y), (h, w, name)


def test_preprocess_extreme_values(ctx, oracle):
    """All-0 / all-255 / checkerboard frames: saturating sums, maximal Laplacian magnitudes (|lap| = 1020)."""
    h, w = 128, 192
    chk = ((np.add.outer(np.arange(h), np.arange(w)) & 1) * 255).astype(np.uint8)
    frames = np.stack([np.zeros((h, w, 3), np.uint8), np.full((h, w, 3), 255, np.uint8),
                       np.repeat(chk[..., None], 3, axis=2), np.repeat((255 - chk)[..., None], 3, axis=2)])
    got = ctx.preprocess_bgr(frames)
    want = oracle.preprocess_bgr(frames)
    for x, y in zip(got, want):
        assert np.array_equal(x, y)
    assert got[3][2] == 1020 * 1020 * (h * w) and got[2][2] == 0        # checkerboard: |lap| = 1020 at every pixel


def test_preprocess_smooth_and_strided(ctx, oracle):
    clip = synth.make_clip(4, 270, 480, seed=3)
    # non-contiguous view: row stride larger than w*3 and a frame stride with a gap
    big = np.zeros((4, 280, 500, 3), np.uint8)
    big[:, :270, :480] = clip
    view = big[:, :270, :480]
    a = ctx.preprocess_bgr(view)
    b = oracle.preprocess_bgr(clip)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_preprocess_device_tensor(ctx, oracle):
    torch = pytest.importorskip("torch")
    clip = synth.make_clip(3, 360, 640, seed=5)
    t = torch.from_numpy(clip).to("cuda:0")
    a = ctx.preprocess_bgr(t)
    b = oracle.preprocess_bgr(clip)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def _smalls(oracle, n, seed):
    clip = synth.make_clip(n, 360, 640, seed=seed, dup_every=3)
    return np.stack([oracle.resize_linear(oracle.bgr2gray(f), 320, 320) for f in clip])


@pytest.mark.parametrize("fold_blur", [1, 0])
def test_farneback_stages_bit_exact(ctx, oracle, fold_blur):
    """fold_blur 1 (the default): the 320-px scale's 3 x 3 pyramid blur is formed inside the polynomial expansion and pyr0 is never written --
    its expansion poly0 must still equal the oracle's; 0: the pyramid kernel writes pyr0 as well."""
    small = _smalls(oracle, 3, seed=11)
    assert ctx.get_option("fb_fold_blur") == 1
    ctx.set_option("fb_fold_blur", fold_blur)
    try:
        fm, fv, flow = ctx.farneback_pairs(small, want_flow=True)
        if fold_blur:
            with pytest.raises(Exception, match="pyr0 does not exist"):       # not "stale memory with a success code"
                ctx.debug_fetch("pyr0", (3, 320, 320), np.float32)
        fetched = {k: (None if (k == 0 and fold_blur) else ctx.debug_fetch(f"pyr{k}", (3, 320 >> k, 320 >> k), np.float32),
                       ctx.debug_fetch(f"poly{k}", (3, 320 >> k, 320 >> k, 5), np.float32)) for k in range(4)}
    finally:
        ctx.set_option("fb_fold_blur", 1)
    # pyramid + polynomial expansion, per frame and level
    ks = {0: (3, 0.0), 1: (3, 0.5), 2: (9, 1.5), 3: (19, 3.5)}
    for k in (3, 2, 1, 0):
        wl = 320 >> k
        pyr, poly = fetched[k]
        for f in range(3):
            blur = oracle.gaussian_blur(small[f].astype(np.float32), *ks[k])
            o_pyr = oracle.resize_linear_f32(blur, wl, wl)
            if k > 0 or not fold_blur:
                assert np.array_equal(pyr[f], o_pyr), f"pyramid level {k} frame {f}"
            o_poly = oracle.poly_exp(o_pyr)
            assert np.array_equal(poly[f], o_poly), f"polyexp level {k} frame {f}"
    for p in range(2):
        o_flow = oracle.farneback(small[p], small[p + 1])
        assert np.array_equal(flow[p], o_flow), f"pair {p}: max |d| = {np.abs(flow[p] - o_flow).max()}"
        m, v = oracle.flow_stats(o_flow)
        assert fm[p] == m and fv[p] == v


def test_farneback_hard_inputs_bit_exact(oracle):
    """Content that drives the warp out of the image, saturates, or is flat: white noise (large
    erratic flow -> the out-of-range branch of UpdateMatrices), constant, half-saturated step edges,
    and a frame pair with a big global shift.  The EXACT level kernels (fb_mode 0); the fast kernel is held to its
    tolerance on the same inputs in test_gpu_fbfast.py (on the constant image cv2's running sums leave ~1e-32 where the
    direct sums give 0)."""
    import avd_hip
    ctx = avd_hip.Context(0)
    ctx.set_option("fb_mode", 0)
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (2, 320, 320), dtype=np.uint8)
    const = np.full((320, 320), 200, np.uint8)
    step = np.zeros((320, 320), np.uint8)
    step[:, 160:] = 255
    step2 = np.zeros((320, 320), np.uint8)
    step2[:, 190:] = 255                                   # 30 px jump of a saturated edge
    ramp = (np.add.outer(np.arange(320), np.arange(320)) % 256).astype(np.uint8)
    frames = np.stack([noise[0], noise[1], const, step, step2, ramp, np.roll(ramp, 25, axis=1), const])
    fm, fv, flow = ctx.farneback_pairs(frames, want_flow=True)
    for p in range(len(frames) - 1):
        o_flow = oracle.farneback(frames[p], frames[p + 1])
        assert np.array_equal(flow[p], o_flow), f"pair {p}: {np.count_nonzero(flow[p] != o_flow)} values differ"
        m, v = oracle.flow_stats(o_flow)
        assert fm[p] == m and fv[p] == v, p
    assert np.isfinite(flow).all()
    ctx.close()


def test_fused_and_two_kernel_paths_agree(oracle):
    """Every pyramid level can run the fused kernel (default) or the k_uv / k_uvp + k_hscan pair that exchanges the
    double intermediate through HBM: all 16 combinations give the oracle's flow, bit for bit."""
    import avd_hip
    small = _smalls(oracle, 3, seed=14)
    want = [oracle.farneback(small[p], small[p + 1]) for p in range(2)]
    with avd_hip.Context(0) as c:
        c.set_option("fb_mode", 0)                           # the exact kernels (the default is the fast level kernel)
        for mask in (0xF, 0x0, 0x1, 0x2, 0x4, 0x8, 0x5, 0xA, 0xE):
            c.set_option("fb_fused", mask)
            fm, fv, flow = c.farneback_pairs(small, want_flow=True)
            for p in range(2):
                assert np.array_equal(flow[p], want[p]), (hex(mask), p)
        with pytest.raises(avd_hip.AvdError):
            c.set_option("no_such_option", 1)


def test_flow_stats_match_numpy(ctx, oracle):
    small = _smalls(oracle, 4, seed=12)
    fm, fv, flow = ctx.farneback_pairs(small, want_flow=True)
    for p in range(3):
        mag = np.sqrt(flow[p][..., 0] ** 2 + flow[p][..., 1] ** 2)
        assert fm[p] == np.mean(mag) and fv[p] == np.var(mag)


def test_analyze_frames_end_to_end(ctx, oracle):
    import avd_hip
    clip = synth.make_clip(12, 360, 640, seed=21, dup_every=5)
    meta = {"width": 640, "height": 360, "fps": 30.0, "duration": 6.0}
    got = avd_hip.FrameAnalyzer(ctx=ctx).analyze(clip, meta)
    want = oracle.analyze_sampled_frames(clip, meta)
    assert got["timeline"] is got["timeline_ai"]
    # tolerance stated by north_star: 1e-4 on ai_score/timeline; here results are identical
    np.testing.assert_allclose(got["timeline"], want["timeline"], rtol=0, atol=1e-4)
    assert got["timeline"] == want["timeline"]
    for key, val in want["summary"].items():
        assert got["summary"][key] == pytest.approx(val, rel=1e-12, abs=1e-12), key
    assert got["summary"]["dup_density"] > 0          # duplicates were detected


def test_long_clip_crosses_farneback_chunk_boundary(ctx, oracle):
    """The Farneback workspace holds 128 pairs; longer clips are processed in chunks that overlap by
    one frame.  131 frames -> pairs 127|128 straddle the boundary: records there must equal the oracle."""
    base = synth.make_clip(8, 64, 96, seed=31, dup_every=0)
    idx = np.arange(131) % 8
    clip = base[idx]                                        # cheap long clip (8 distinct frames cycled)
    rec = ctx.analyze_frames(clip)
    small = np.stack([oracle.resize_linear(oracle.bgr2gray(f), 320, 320) for f in base])
    for p in (0, 126, 127, 128, 129):
        a, b = idx[p], idx[p + 1]
        m, v = oracle.flow_stats(oracle.farneback(small[a], small[b]))
        assert rec["flow_mean"][p + 1] == m and rec["flow_var"][p + 1] == v, p
    # the cycle repeats every 8 frames: equal pairs must give equal records on both sides of the boundary
    assert rec["flow_mean"][120 + 1] == rec["flow_mean"][128 + 1] and rec["flow_var"][121 + 1] == rec["flow_var"][129 + 1]
    assert rec["ham"][0] == -1 and np.all(rec["ham"][1:] >= 0)


def test_streaming_chunks_equal_one_shot(ctx):
    import avd_hip
    clip = synth.make_clip(9, 180, 320, seed=2)
    one = ctx.analyze_frames(clip)
    fa = avd_hip.FrameAnalyzer(ctx=ctx, chunk=4)
    two = fa.records_stream(iter(clip))
    assert np.array_equal(one, two)


def test_error_paths(ctx):
    import avd_hip
    with pytest.raises(avd_hip.AvdError):
        ctx.preprocess_bgr(np.zeros((1, 16, 16, 3), np.uint8))     # < 32x32: unsupported
    rec = ctx.analyze_frames(np.zeros((1, 64, 64, 3), np.uint8))    # single frame: no flow
    assert rec["ham"][0] == -1 and rec["flow_mean"][0] == 0 and rec["lap_sumsq"][0] == 0


def test_device_resident_inputs(ctx, oracle):
    """Inputs already in HBM (torch-ROCm tensors): a strided frame view and the 320x320 stack."""
    torch = pytest.importorskip("torch")
    clip = synth.make_clip(5, 200, 336, seed=41)
    big = torch.zeros((5, 208, 352, 3), dtype=torch.uint8, device="cuda:0")
    big[:, :200, :336] = torch.from_numpy(clip).to("cuda:0")
    rec = ctx.analyze_frames(big[:, :200, :336])                       # row stride 1056 B, frame stride with a gap
    assert np.array_equal(rec, ctx.analyze_frames(clip))
    small = np.stack([oracle.resize_linear(oracle.bgr2gray(f), 320, 320) for f in clip])
    fm, fv = ctx.farneback_pairs(torch.from_numpy(small).to("cuda:0"))
    assert np.array_equal(fm, rec["flow_mean"][1:]) and np.array_equal(fv, rec["flow_var"][1:])


def test_worker_threads_borrow_pooled_contexts(ctx):
    """reference api.py:133 runs the analyzer on worker threads: each request borrows a context from the bounded
    pool for its duration; more threads than contexts, results identical to the session context's."""
    import threading
    from avd_hip import analyzer
    clips = [synth.make_clip(6, 120, 200, seed=s) for s in (51, 52, 53, 54)]
    want = [ctx.analyze_frames(c) for c in clips]
    pool = analyzer.ContextPool(max_contexts=2, keep_warm=1)
    got = [None] * len(clips)
    used = set()

    def work(i):
        for _ in range(3):
            with pool.borrow(0) as c:
                used.add(id(c))
                got[i] = c.analyze_frames(clips[i])

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(clips))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert 1 <= len(used) <= 2 and pool.stats(0)["created"] == len(used)
    for i in range(len(clips)):
        assert np.array_equal(got[i], want[i])
    # a context whose workspace was released re-reserves it on its next use
    with pool.borrow(0) as c:
        c.release_workspace()
        assert np.array_equal(c.analyze_frames(clips[0]), want[0])
    pool.close()


def test_non_contiguous_and_fresh_device_tensors(ctx, oracle):
    """Device inputs that torch's CURRENT stream is still producing when the call is made: a planar tensor viewed
    as interleaved (needs ``.contiguous()``: a copy kernel on torch's stream) and a freshly computed one.  The
    context's own stream is ordered behind torch's with an event (avd_wait_stream), the contiguous temporary is
    kept alive by the binding until the clip is drained."""
    torch = pytest.importorskip("torch")
    import avd_hip
    clip = synth.make_clip(24, 1080, 1920, seed=61, dup_every=5)
    want = ctx.analyze_frames(clip)
    dev = torch.from_numpy(clip).to("cuda:0")
    planar = dev.permute(0, 3, 1, 2).contiguous()                     # [N,3,H,W]
    torch.cuda.synchronize()
    view = planar.permute(0, 2, 3, 1)                                 # uint8[N,H,W,3] with channel stride H*W
    assert not view.is_contiguous()
    assert np.array_equal(ctx.analyze_frames(view), want)
    # fresh: produced by kernels enqueued right before the call, no synchronisation in between
    for _ in range(3):
        fresh = (planar.to(torch.int16) + 0).to(torch.uint8).permute(0, 2, 3, 1)
        assert np.array_equal(ctx.analyze_frames(fresh), want)
    # asynchronous form: the temporary made by .contiguous() must survive until drain()
    runner = avd_hip.ClipsInFlight(device=0, depth=2)
    outs = list(runner.run((i, (planar + 0).permute(0, 2, 3, 1)) for i in range(4)))
    assert all(np.array_equal(rec, want) for _, rec in outs)
    # sliced rows / columns (strides, no copy needed) and a flipped batch (negative-free but re-ordered: copy)
    sub = dev[:, 100:900, 64:1600]
    assert np.array_equal(ctx.analyze_frames(sub), ctx.analyze_frames(np.ascontiguousarray(clip[:, 100:900, 64:1600])))


def test_full_size_clip_properties(ctx, oracle):
    """BASELINE.json configs[1] at its full size (120 x 1080p, 119 pairs): too big for the oracle as a whole,
    so parity is checked through properties that do not depend on the size -- locality (a frame's record only
    depends on the frame and its predecessor), order reversal, exact duplicates, a constant frame -- plus
    oracle spot checks of single frames and pairs."""
    n, h, w = 120, 1080, 1920
    clip = synth.make_clip(n, h, w, seed=0)                   # every 10th frame duplicates its predecessor
    clip[77] = 93                                             # one constant frame
    rec = ctx.analyze_frames(clip)
    assert rec["ham"][0] == -1 and np.all(rec["ham"][1:] >= 0)
    # exact duplicates: identical hash and moments; the flow is tiny but NOT zero (cv2 treats the last column
    # and row as warped outside the image, so the normal equations there see a difference)
    dups = [i for i in range(1, n) if np.array_equal(clip[i], clip[i - 1])]
    assert len(dups) >= 10
    for i in dups:
        assert rec["ham"][i] == 0 and 0.0 <= rec["flow_mean"][i] < 1e-2 and rec["flow_var"][i] < 1e-2
        assert rec["lap_sum"][i] == rec["lap_sum"][i - 1] and rec["lap_sumsq"][i] == rec["lap_sumsq"][i - 1]
    # constant frame: the Laplacian vanishes identically
    assert rec["lap_sum"][77] == 0 and rec["lap_sumsq"][77] == 0
    # locality: a sub-clip reproduces the corresponding records bit for bit (except its own first frame's pair)
    sub = ctx.analyze_frames(clip[40:61])
    for key in ("lap_sum", "lap_sumsq"):
        assert np.array_equal(sub[key], rec[key][40:61]), key
    for key in ("flow_mean", "flow_var", "ham"):
        assert np.array_equal(sub[key][1:], rec[key][41:61]), key
    # order reversal: per-frame moments reverse, the Hamming distance of a pair is symmetric
    rev = ctx.analyze_frames(clip[::-1])
    assert np.array_equal(rev["lap_sum"], rec["lap_sum"][::-1])
    assert np.array_equal(rev["lap_sumsq"], rec["lap_sumsq"][::-1])
    assert np.array_equal(rev["ham"][1:], rec["ham"][1:][::-1])
    # oracle spot checks: four single frames, two pairs (one across the scene cut in the middle)
    for i in (0, 59, 60, 119):
        small, hsh, s, q = oracle.preprocess_bgr(clip[i:i + 1])
        assert rec["lap_sum"][i] == s[0] and rec["lap_sumsq"][i] == q[0]
    for i in (33, 60, dups[0]):
        small, hsh, s, q = oracle.preprocess_bgr(clip[i - 1:i + 1])
        fm, fv = oracle.farneback_pairs(small)
        assert rec["flow_mean"][i] == fm[0] and rec["flow_var"][i] == fv[0]
        assert rec["ham"][i] == int(np.sum(hsh[0] ^ hsh[1]))


def test_clips_in_flight_equal_one_at_a_time(ctx):
    """Service throughput mode: several clips in flight on their own contexts return, in submission order, exactly
    the records of one-at-a-time analysis (different geometries and lengths, more clips than contexts)."""
    import avd_hip
    clips = [(f"clip{i}", synth.make_clip(n, h, w, seed=50 + i, dup_every=3))
             for i, (n, h, w) in enumerate([(7, 360, 640), (3, 240, 426), (12, 720, 1280), (2, 67, 101), (9, 360, 640),
                                            (5, 1080, 1920), (4, 96, 128)])]
    runner = avd_hip.ClipsInFlight(device=0, depth=3)
    got = list(runner.run(clips))
    assert [t for t, _ in got] == [t for t, _ in clips]
    for (tag, rec), (_, frames) in zip(got, clips):
        want = ctx.analyze_frames(frames)
        for key in ("lap_sum", "lap_sumsq", "flow_mean", "flow_var", "ham"):
            assert np.array_equal(rec[key], want[key]), (tag, key)
