"""The Farneback level kernels against the oracle (reference site: cv2.calcOpticalFlowFarneback at
app/analyzers/video.py:45).

  fb_mode = 1 "fast"  (default, csrc/avd_fbfast.hip): cv2's vertical running sums literally, the horizontal 15-column
                      window sums formed directly in double instead of as cv2's running double sum -- and every pair the
                      kernels flag as ILL-POSED re-run by the exact kernels (the host reads the flag words and launches
                      them for a compacted list: launch_farneback_rerun) before anything is handed to the caller.
                      Guarantee asserted here, with NO exception for any pair: flagged pairs bit-identical to the oracle
                      (they ARE the exact kernels' result); the others dense flow max |delta| <= 1e-5 px (in practice 0
                      differing values), flow_mean / flow_var relative 1e-6; ai_susp |delta| <= 1e-6 (north_star: 1e-4);
                      summary and the fused result of the hard-set clip equal to the oracle chain's.
  fb_rerun = 0        the fast kernels alone (A/B): on the ill-posed pairs of the hard set they differ from the oracle by
                      far more than any tolerance -- the negative control that shows the re-run is what closes the gap.
  fb_mode = 0 "exact" (csrc/avd_fbfused.hip and the two-kernel path): bit-identical, asserted with array_equal.

ILL-POSED pairs.  Three pairs of the hard set (a saturated step edge against an unrelated ramp, a ramp against itself
rolled by 25 px) make the 2 x 2 normal equations singular over whole regions: the oracle's own flow there is hundreds of
pixels on a 320-px image and moves by tens to hundreds of pixels when ONE ulp of noise is put on its pyramid.  The level
kernel recognises them (determinant cancellation > 2000 or a displacement > 0.3 of the level width; criterion derived in
tools/experiments/fb_illposed_run.py) and the exact kernels compute them again.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from avd_hip import synth  # noqa: E402

FLOW_TOL = 1e-5      # px
SUSP_TOL = 1e-6


@pytest.fixture(scope="module")
def ctxs():
    import avd_hip
    fast, exact, fastonly = avd_hip.Context(0), avd_hip.Context(0), avd_hip.Context(0)
    fast.set_option("fb_mode", 1)
    fast.set_option("fb_rerun", 1)
    exact.set_option("fb_mode", 0)
    fastonly.set_option("fb_mode", 1)
    fastonly.set_option("fb_rerun", 0)
    yield {"fast": fast, "exact": exact, "fastonly": fastonly}
    for c in (fast, exact, fastonly):
        c.close()


def _smalls(oracle, clip):
    return np.stack([oracle.resize_linear(oracle.bgr2gray(f), 320, 320) for f in clip])


def _hard_frames():
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (2, 320, 320), dtype=np.uint8)
    const = np.full((320, 320), 200, np.uint8)
    step = np.zeros((320, 320), np.uint8)
    step[:, 160:] = 255
    step2 = np.zeros((320, 320), np.uint8)
    step2[:, 190:] = 255
    ramp = (np.add.outer(np.arange(320), np.arange(320)) % 256).astype(np.uint8)
    xx = np.arange(320)
    stripes = [(127 + 120 * np.sin((xx[None, :] + 3 * k) * 0.2) * np.ones((320, 1))).astype(np.uint8) for k in range(2)]
    box = np.full((2, 320, 320), 128, np.uint8)
    box[0, 100:200, 100:200] = 140
    box[1, 103:203, 98:198] = 140
    return np.stack([noise[0], noise[1], const, step, step2, ramp, np.roll(ramp, 25, axis=1), const,
                     stripes[0], stripes[1], box[0], box[1]])


def _check_pairs(ctxs, oracle, frames, tag, expect_rerun=None):
    want = [oracle.farneback(frames[p], frames[p + 1]) for p in range(len(frames) - 1)]
    stats = [oracle.flow_stats(f) for f in want]
    # exact kernels: bit-identical
    fm, fv, flow = ctxs["exact"].farneback_pairs(frames, want_flow=True)
    for p, o in enumerate(want):
        assert np.array_equal(flow[p], o), (tag, "exact", p, int(np.count_nonzero(flow[p] != o)))
        assert fm[p] == stats[p][0] and fv[p] == stats[p][1], (tag, "exact", p)
    # default mode (fast kernels + re-run of flagged pairs): the stated tolerance on EVERY pair, no carve-out
    fm, fv, flow = ctxs["fast"].farneback_pairs(frames, want_flow=True)
    rerun = ctxs["fast"].get_option("rerun_pairs")
    ndiff, worst = 0, 0.0
    for p, o in enumerate(want):
        d = np.abs(flow[p].astype(np.float64) - o.astype(np.float64))
        ndiff += int(np.count_nonzero(flow[p] != o))
        worst = max(worst, float(d.max()))
        assert d.max() <= FLOW_TOL, (tag, "fast", p, float(d.max()))
        assert fm[p] == pytest.approx(stats[p][0], rel=1e-6, abs=1e-7), (tag, p)
        assert fv[p] == pytest.approx(stats[p][1], rel=1e-6, abs=1e-7), (tag, p)
    assert np.isfinite(flow).all()
    print(f"[fbfast] {tag}: {ndiff} of {flow.size} flow values differ from the oracle, max |delta| = {worst:.3g} px, {rerun} pairs re-run")
    if expect_rerun is not None:
        assert rerun == expect_rerun, (tag, rerun)
    return ndiff, worst


def test_flow_on_smooth_clips(ctxs, oracle):
    """The synthetic clips of the parity suite: translation, duplicates, one scene cut (flow up to ~25 px)."""
    clip = synth.make_clip(7, 360, 640, seed=11, dup_every=3)
    _check_pairs(ctxs, oracle, _smalls(oracle, clip), "smooth 360p", expect_rerun=0)
    clip = synth.make_clip(4, 720, 1280, seed=1, dup_every=0)
    _check_pairs(ctxs, oracle, _smalls(oracle, clip), "720p with scene cut", expect_rerun=0)


def test_flow_on_hard_inputs(ctxs, oracle):
    """White noise (erratic flow, warps leave the image), constant and saturated step images (every sum is zero or
    cancels), ramps with a 25 px shift, 1-D stripes (aperture problem: the 2 x 2 system is near singular), a flat image
    with one box.  Every pair within the stated tolerance; the ill-posed ones (4, 5, 6: step / ramp / rolled ramp) are
    flagged and re-run, so they are bit-identical."""
    frames = _hard_frames()
    _check_pairs(ctxs, oracle, frames, "hard set")
    want = [oracle.farneback(frames[p], frames[p + 1]) for p in range(len(frames) - 1)]
    fm, fv, flow = ctxs["fast"].farneback_pairs(frames, want_flow=True)
    for p in (4, 5, 6):
        assert np.array_equal(flow[p], want[p]), p
    # negative control: without the re-run the fast kernels cannot follow these pairs (this is what the re-run closes)
    _, _, flow0 = ctxs["fastonly"].farneback_pairs(frames, want_flow=True)
    assert ctxs["fastonly"].get_option("rerun_pairs") == 0
    assert max(float(np.abs(flow0[p] - want[p]).max()) for p in (4, 5, 6)) > 100 * FLOW_TOL
    # what the reference derives from the flow, for the hard set run as a "clip": timeline (video.py:54-57), summary
    # (video.py:61-71) and the fused result (fusion.py:16-109) -- equal to the oracle chain's
    import avd_hip
    from app.analyzers import fusion
    clip = np.repeat(frames[..., None], 3, axis=3)
    meta = {"width": 320, "height": 320, "fps": 30.0, "duration": 6.0}
    want_v = oracle.analyze_sampled_frames(clip, meta)
    got_v = avd_hip.FrameAnalyzer(ctx=ctxs["fast"]).analyze(clip, meta)
    np.testing.assert_allclose(got_v["timeline"], want_v["timeline"], rtol=0, atol=SUSP_TOL)
    for key, val in want_v["summary"].items():
        assert got_v["summary"][key] == pytest.approx(val, rel=1e-6, abs=1e-9), key
    audio = {"scores": {}, "flags_audio": {"error": "FileNotFoundError"}, "timeline": [0.5] * 6}
    hints = {"bpp": 0.1, "compression": "normal", "dup_avg": 0.0}
    fw = fusion.fuse(dict(audio, timeline=list(audio["timeline"])), {k: (list(v) if isinstance(v, list) else v) for k, v in want_v.items()}, hints)
    fg = fusion.fuse(dict(audio, timeline=list(audio["timeline"])), {k: (list(v) if isinstance(v, list) else v) for k, v in got_v.items()}, hints)
    assert fg["result"] == fw["result"]
    np.testing.assert_allclose(fg["timeline_binned"], fw["timeline_binned"], rtol=0, atol=SUSP_TOL)
    assert fg["peaks"] == fw["peaks"]


def test_rerun_records_mark_the_pairs(ctxs, oracle):
    """avd_analyze_frames: the record of the frame that closes a re-run pair carries the level mask in `reserved`, the
    counter avd_get_option("rerun_pairs") agrees, and a context with the re-run switched off marks nothing."""
    frames = _hard_frames()
    clip = np.repeat(frames[..., None], 3, axis=3)
    rec = ctxs["fast"].analyze_frames(clip)
    marked = [int(i) for i in np.nonzero(rec["reserved"])[0]]
    assert set((5, 6, 7)) <= set(marked), marked          # pairs 4, 5, 6 close at frames 5, 6, 7
    assert ctxs["fast"].get_option("rerun_pairs") == len(marked)
    rec0 = ctxs["fastonly"].analyze_frames(clip)
    assert not rec0["reserved"].any() and ctxs["fastonly"].get_option("rerun_pairs") == 0
    small, _, _, _ = oracle.preprocess_bgr(clip)
    fm, fv = oracle.farneback_pairs(small)
    np.testing.assert_allclose(rec["flow_mean"][1:], fm, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rec["flow_var"][1:], fv, rtol=1e-6, atol=1e-7)


def test_strip_seams_and_borders(ctxs, oracle):
    """A pair is split into column strips with recomputed halos (320 px: seam at column 160; 160 px: at column 80):
    the flow must not show the seam.  White noise makes every column's sums different."""
    frames = synth.random_frames(3, 320, 320, seed=77)[..., 1].copy()
    want = oracle.farneback(frames[0], frames[1])
    _, _, flow = ctxs["fastonly"].farneback_pairs(frames[:2], want_flow=True)
    d = np.abs(flow[0] - want)
    for cols in (slice(150, 170), slice(0, 8), slice(312, 320)):
        assert d[:, cols].max() <= FLOW_TOL
    assert d[:8].max() <= FLOW_TOL and d[-8:].max() <= FLOW_TOL


def test_folded_launches_are_bit_identical(oracle):
    """What the fast mode folds into its level launches (option fb_fold_up, a bit mask) never changes a bit of the flow: 1 = the
    320-px level's first launch forms its initial flow from the 160-px level's on the fly (cv2: resize x 2, INTER_LINEAR, times 2)
    instead of reading what k_flow_up wrote; 2 = the 160- and 80-px levels do the same in a prologue of their first launch; 4 = the
    80- and 40-px levels run their three iterations in ONE launch (flow handed over through L2 between iterations).  On smooth
    clips, on white noise and on the hard set (flows of hundreds of pixels, first / last columns and rows)."""
    import avd_hip
    sets = [_smalls(oracle, synth.make_clip(5, 360, 640, seed=21, dup_every=3)), synth.random_frames(4, 320, 320, seed=5)[..., 1].copy(),
            _hard_frames()]

    def levels(c, n):
        return [c.debug_fetch(f"flow{k}", (n - 1, 2, 320 >> k, 320 >> k), np.float32) for k in range(4)]

    with avd_hip.Context(0) as c:
        c.set_option("fb_mode", 1)
        assert c.get_option("fb_fold_up") == 5
        for si, frames in enumerate(sets):
            c.set_option("fb_rerun", 0)                    # the fast kernels themselves, on every pair
            c.set_option("fb_fold_up", 0)
            fm0, fv0, flow0 = c.farneback_pairs(frames, want_flow=True)
            lvl0 = levels(c, len(frames))
            for mask in (1, 2, 4, 6, 7):
                c.set_option("fb_fold_up", mask)
                fm1, fv1, flow1 = c.farneback_pairs(frames, want_flow=True)
                assert np.array_equal(flow0.view(np.uint32), flow1.view(np.uint32)), mask
                assert np.array_equal(fm0, fm1) and np.array_equal(fv0, fv1), mask
                for k, lv in enumerate(levels(c, len(frames))):            # and the final flow of every pyramid level
                    assert np.array_equal(lv.view(np.uint32), lvl0[k].view(np.uint32)), (mask, k)
            c.set_option("fb_fold_up", 5)


def _flagged_mix(n_pairs, seed):
    """frames whose consecutive pairs alternate between content the level kernels flag (ramps rolled, stripes, 2-px checkerboards against
    their inverse) and content they do not (smooth translation)"""
    from tests.content_families import families
    fam = families()
    rng = np.random.default_rng(seed)
    names = ["ramp_roll", "smooth_shift", "stripes", "checker", "pink_shift"]
    frames = []
    while len(frames) < n_pairs + 1:
        a, b = fam[names[(len(frames) // 2) % len(names)]](rng)
        frames += [a, b]
    return np.stack(frames[:n_pairs + 1])


def test_rerun_paths_are_the_exact_kernels(ctxs, oracle):
    """The exact re-run of flagged pairs from a compacted list: few pairs (the 160- / 320-px levels through the two-kernel path, every
    choice of the level mask fb_rerun_fused), many pairs (> 32 flagged: the fused kernels, one workgroup per pair), through both entry points
    (avd_farneback_pairs settles a chunk at once; avd_analyze_frames when the call is drained).  A flagged pair's flow is the exact mode's,
    bit for bit; the unflagged ones keep the tolerance."""
    import avd_hip
    for n_pairs, seed in ((9, 3), (80, 4)):
        frames = _flagged_mix(n_pairs, seed)
        xm, xv, xflow = ctxs["exact"].farneback_pairs(frames, want_flow=True)
        if n_pairs <= 9:
            for p in range(n_pairs):
                assert np.array_equal(xflow[p], oracle.farneback(frames[p], frames[p + 1])), p
        clip = np.repeat(frames[..., None], 3, axis=3)
        with avd_hip.Context(0) as c:
            for mask in ((0x8, 0xC, 0xF, 0xA) if n_pairs <= 9 else (0xC,)):
                c.set_option("fb_rerun_fused", mask)
                fm, fv, flow = c.farneback_pairs(frames, want_flow=True)
                m = c.get_option("rerun_pairs")
                rec = c.analyze_frames(clip)
                flagged = np.nonzero(rec["reserved"][1:])[0]
                assert len(flagged) == m == c.get_option("rerun_pairs") and m >= (3 if n_pairs <= 9 else 33), (n_pairs, hex(mask), m)
                for p in range(n_pairs):
                    if rec["reserved"][p + 1]:
                        assert np.array_equal(flow[p].view(np.uint32), xflow[p].view(np.uint32)), (n_pairs, hex(mask), p)
                        assert fm[p] == xm[p] and fv[p] == xv[p], (n_pairs, hex(mask), p)
                    else:
                        assert np.abs(flow[p] - xflow[p]).max() <= FLOW_TOL, (n_pairs, hex(mask), p)
                    assert rec["flow_mean"][p + 1] == fm[p] and rec["flow_var"][p + 1] == fv[p], (n_pairs, hex(mask), p)


def test_rerun_across_chunks_and_in_batches(ctxs):
    """More pairs than the Farneback scratch holds (512): every chunk's flagged pairs are settled before the scratch is reused; and a batch
    of clips (one launch sequence over all their pairs): the records equal the exact mode's on flagged pairs, the tolerance elsewhere."""
    import avd_hip
    frames = _flagged_mix(560, 9)
    xm, xv = ctxs["exact"].farneback_pairs(frames)
    with avd_hip.Context(0) as c:
        fm, fv = c.farneback_pairs(frames)
        assert c.get_option("rerun_pairs") >= 150
        np.testing.assert_allclose(fm, xm, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(fv, xv, rtol=1e-6, atol=1e-7)
        clip = np.repeat(frames[..., None], 3, axis=3)
        rec = c.analyze_frames(clip)
        flagged = rec["reserved"][1:] != 0
        assert flagged.sum() == c.get_option("rerun_pairs") >= 150
        assert np.array_equal(rec["flow_mean"][1:][flagged], xm[flagged]) and np.array_equal(rec["flow_var"][1:][flagged], xv[flagged])
        np.testing.assert_allclose(rec["flow_mean"][1:], xm, rtol=1e-6, atol=1e-7)
        # a batch: three clips of 7 / 30 / 12 frames cut out of the same material
        cuts = [(0, 7), (40, 70), (100, 112)]
        recs = c.analyze_batch([clip[a:b] for a, b in cuts])
        for (a, b), r in zip(cuts, recs):
            assert r["ham"][0] == -1 and r["flow_mean"][0] == 0
            f = r["reserved"][1:] != 0
            assert np.array_equal(r["flow_mean"][1:][f], xm[a:b - 1][f]) and np.array_equal(r["flow_var"][1:][f], xv[a:b - 1][f])
            np.testing.assert_allclose(r["flow_mean"][1:], xm[a:b - 1], rtol=1e-6, atol=1e-7)


def test_level_shapes_are_bit_identical(oracle):
    """The two shapes of the 160-px level -- one three-block strip per pair (fb_wide160, the default) or two strips -- differ only in
    which lanes hold which columns and in the grouping of the solver's window sums (four columns per lane or two).  On well-posed
    content the flow of EVERY level is bit-identical; on the hard set it is with the re-run on (the ill-posed pairs are re-run exactly
    in either shape)."""
    import avd_hip
    sets = [_smalls(oracle, synth.make_clip(6, 360, 640, seed=22, dup_every=4)), synth.random_frames(8, 320, 320, seed=6)[..., 1].copy(),
            _hard_frames()]

    def levels(c, n):
        return [c.debug_fetch(f"flow{k}", (n - 1, 2, 320 >> k, 320 >> k), np.float32) for k in range(4)]

    with avd_hip.Context(0) as c:
        c.set_option("fb_mode", 1)
        assert c.get_option("fb_wide160") == 2           # default: chosen per call by what else is in flight
        for si, frames in enumerate(sets):
            hard = si == 2
            c.set_option("fb_rerun", 1 if hard else 0)
            c.set_option("fb_wide160", 0)
            fm0, fv0, flow0 = c.farneback_pairs(frames, want_flow=True)
            lvl0 = levels(c, len(frames))
            n0 = c.get_option("rerun_pairs")
            c.set_option("fb_wide160", 1)
            fm1, fv1, flow1 = c.farneback_pairs(frames, want_flow=True)
            c.set_option("fb_wide160", 2)
            assert c.get_option("rerun_pairs") == n0
            assert np.array_equal(flow0.view(np.uint32), flow1.view(np.uint32)), si
            assert np.array_equal(fm0, fm1) and np.array_equal(fv0, fv1), si
            if not hard:
                for k, lv in enumerate(levels(c, len(frames))):
                    assert np.array_equal(lv.view(np.uint32), lvl0[k].view(np.uint32)), (si, k)


@pytest.mark.parametrize("n,h,w,dur", [(12, 360, 640, 6.0), (6, 720, 1280, 3.0), (5, 1080, 1920, 2.5), (3, 2160, 3840, 1.5)])
def test_ai_susp_within_tolerance_every_geometry(ctxs, oracle, n, h, w, dur):
    """End to end (video.py:36-83): timeline / summary of both modes against the oracle chain on the BASELINE geometries."""
    import avd_hip
    clip = synth.make_clip(n, h, w, seed=h + n, dup_every=5)
    meta = {"width": w, "height": h, "fps": 30.0, "duration": dur}
    want = oracle.analyze_sampled_frames(clip, meta)
    got = avd_hip.FrameAnalyzer(ctx=ctxs["exact"]).analyze(clip, meta)
    assert got["timeline"] == want["timeline"]
    got = avd_hip.FrameAnalyzer(ctx=ctxs["fast"]).analyze(clip, meta)
    np.testing.assert_allclose(got["timeline"], want["timeline"], rtol=0, atol=SUSP_TOL)
    for key, val in want["summary"].items():
        assert got["summary"][key] == pytest.approx(val, rel=1e-6, abs=1e-9), key
    print(f"[fbfast] {h}p: timeline identical = {got['timeline'] == want['timeline']}")


def test_modes_can_be_switched_on_one_context(oracle):
    """fb_mode is a per-context option; switching it between calls re-uses the workspace (the two-kernel scratch is only
    reserved when that path is selected)."""
    import avd_hip
    small = _smalls(oracle, synth.make_clip(3, 180, 320, seed=4, dup_every=0))
    want = [oracle.farneback(small[p], small[p + 1]) for p in range(2)]
    with avd_hip.Context(0) as c:
        for mode, fused in ((1, 0xF), (0, 0xF), (0, 0x0), (1, 0x0), (0, 0x5)):
            c.set_option("fb_mode", mode)
            c.set_option("fb_fused", fused)
            _, _, flow = c.farneback_pairs(small, want_flow=True)
            for p in range(2):
                if mode == 0:
                    assert np.array_equal(flow[p], want[p]), (mode, hex(fused), p)
                else:
                    assert np.abs(flow[p] - want[p]).max() <= FLOW_TOL, (mode, hex(fused), p)


def test_the_160_px_shape_follows_what_is_in_flight(oracle):
    """fb_wide160 = 2 (default): a call enqueued while another context of the process holds an undrained call runs the 160-px level as one strip per pair
    (fewer CU-microseconds), a call that has the chip to itself as two strips (shorter launches); the records are the same either way."""
    import avd_hip
    clip = synth.make_clip(6, 360, 640, seed=31, dup_every=0)
    with avd_hip.Context(0) as a, avd_hip.Context(0) as b:
        alone = b.analyze_frames(clip)
        assert b.get_option("fb_wide160_used") == 0
        rec = np.zeros(len(clip), avd_hip.RECORD_DTYPE)
        keep = a.analyze_frames_async(clip, rec)               # enqueued, not drained
        shared = b.analyze_frames(clip)
        assert b.get_option("fb_wide160_used") == 1
        a.synchronize()
        del keep
        assert np.array_equal(alone, shared) and np.array_equal(rec, alone)
        again = b.analyze_frames(clip)
        assert b.get_option("fb_wide160_used") == 0 and np.array_equal(again, alone)


def test_a_waiting_thread_settles_the_other_contexts_tails():
    """One host thread, several contexts in flight (avd_hip.ClipsInFlight, bench.py): while it waits in avd_synchronize of one context it sends
    the flagged pairs of the OTHERS through the exact kernels as their fast passes finish (avd_capi.hip, tail_help_others).  Who settles a call
    changes nothing: records identical to the same clips analysed one at a time, with the helper off (option tail_help = 0) and with one thread
    per context; clips of different length and flag density, drained in and out of submission order, many rounds."""
    import threading
    import avd_hip
    clips = [np.repeat(_flagged_mix(n, seed)[..., None], 3, axis=3) for n, seed in ((40, 21), (9, 22), (64, 23), (3, 24))]
    clips.append(np.repeat(_flagged_mix(30, 25)[::2][..., None], 3, axis=3))
    with avd_hip.Context(0) as c:
        c.set_option("tail_help", 0)
        want = [c.analyze_frames(k).copy() for k in clips]
    assert sum(int((w["reserved"] != 0).sum()) for w in want) >= 40 and all((w["reserved"] != 0).any() for w in want)
    ctxs = [avd_hip.Context(0) for _ in range(4)]
    try:
        assert all(c.get_option("tail_help") == 1 for c in ctxs)
        for help_on in (1, 0):
            for c in ctxs:
                c.set_option("tail_help", help_on)
            for rnd in range(6):
                order = [(rnd + j) % len(clips) for j in range(len(ctxs))]
                recs = [np.zeros(len(clips[k]), avd_hip.RECORD_DTYPE) for k in order]
                for j, k in enumerate(order):
                    ctxs[j].analyze_frames_async(clips[k], recs[j])
                drain = list(range(len(ctxs))) if rnd % 2 == 0 else list(reversed(range(len(ctxs))))
                for j in drain:
                    ctxs[j].synchronize()
                    assert ctxs[j].get_option("rerun_pairs") == int((want[order[j]]["reserved"] != 0).sum())
                for j, k in enumerate(order):
                    assert recs[j].tobytes() == want[k].tobytes(), (help_on, rnd, j, k)
        # one thread per context, all of them helping each other
        for c in ctxs:
            c.set_option("tail_help", 1)
        bad = []

        def work(j):
            rec = np.zeros(max(len(k) for k in clips), avd_hip.RECORD_DTYPE)
            for i in range(8):
                k = (i + j) % len(clips)
                r = rec[:len(clips[k])]
                ctxs[j].analyze_frames_async(clips[k], r)
                ctxs[j].synchronize()
                if r.tobytes() != want[k].tobytes():
                    bad.append((j, i, k))
        ths = [threading.Thread(target=work, args=(j,)) for j in range(len(ctxs))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not bad, bad
        # a context destroyed with an unsettled tail, while another one waits
        extra = avd_hip.Context(0)
        r0, r1 = np.zeros(len(clips[0]), avd_hip.RECORD_DTYPE), np.zeros(len(clips[2]), avd_hip.RECORD_DTYPE)
        extra.analyze_frames_async(clips[0], r0)
        ctxs[0].analyze_frames_async(clips[2], r1)
        extra.close()
        ctxs[0].synchronize()
        assert r1.tobytes() == want[2].tobytes()
    finally:
        for c in ctxs:
            c.close()
