"""avd_analyze_batch: several clips in one call (BASELINE.json configs[2] / configs[4]; the reference analyses one file per
request in a sequential loop, app/analyzers/video.py:27-58).  Batched == one at a time == oracle; geometry changes re-use
cached tables and allocate nothing in steady state."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from avd_hip import synth  # noqa: E402


def _records_equal(a, b, tag=""):
    for key in ("lap_sum", "lap_sumsq", "flow_mean", "flow_var", "ham"):
        assert np.array_equal(a[key], b[key]), (tag, key, np.flatnonzero(a[key] != b[key])[:8])


def test_13_short_720p_clips_batched_equal_one_at_a_time_and_oracle(oracle):
    """configs[0]-sized clips (20 sampled frames of 720p): 13 of them in one call = 259 consecutive pairs in ONE Farneback
    launch sequence (12 of them straddle a clip boundary and are ignored)."""
    import avd_hip
    from tests.test_host_and_abi import _records_from_oracle
    base = [synth.make_clip(20, 720, 1280, seed=100 + i, dup_every=7 if i % 2 else 0, scene_cut=bool(i % 3 == 0)) for i in range(3)]
    clips = [base[i % 3] if i < 3 else np.ascontiguousarray(base[i % 3][::-1] if i % 2 else np.roll(base[i % 3], i, axis=0)) for i in range(13)]
    with avd_hip.Context(0) as c:
        c.set_option("fb_mode", 0)                       # exact kernels: records must equal the oracle bit for bit
        batched = c.analyze_batch(clips)
        single = [c.analyze_frames(x) for x in clips]
        assert len(batched) == 13
        for i, (a, b) in enumerate(zip(batched, single)):
            assert len(a) == 20 and a["ham"][0] == -1 and a["flow_mean"][0] == 0
            _records_equal(a, b, f"clip {i}")
        for i in range(3):                               # the three distinct clips against the oracle chain
            _records_equal(batched[i], _records_from_oracle(oracle, clips[i]), f"oracle clip {i}")
        c.set_option("fb_mode", 1)                       # the fast kernel: the same batch, same records on these inputs
        fast = c.analyze_batch(clips)
        for i, (a, b) in enumerate(zip(fast, single)):
            np.testing.assert_allclose(a["flow_mean"], b["flow_mean"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(a["flow_var"], b["flow_var"], rtol=1e-6, atol=1e-7)
            assert np.array_equal(a["ham"], b["ham"]) and np.array_equal(a["lap_sumsq"], b["lap_sumsq"])


def test_mixed_geometries_and_nv12_in_one_batch(oracle):
    """configs[4]: 720p / 1080p / 4K / odd sizes and an NV12 clip in one call, twice (the second call re-uses every table)."""
    import avd_hip
    geoms = [(5, 720, 1280), (4, 1080, 1920), (2, 2160, 3840), (3, 67, 101), (6, 360, 640), (3, 720, 1280)]
    clips = [synth.make_clip(n, h, w, seed=h + n, dup_every=0) for n, h, w in geoms]
    y, uv = synth.bgr_to_nv12(synth.make_clip(4, 360, 640, seed=9, dup_every=0))
    items = clips[:3] + [(y, uv)] + clips[3:]
    with avd_hip.Context(0) as c:
        want = [c.analyze_frames_nv12(*x) if isinstance(x, tuple) else c.analyze_frames(x) for x in items]
        for rep in range(2):
            got = c.analyze_batch(items)
            for i, (a, b) in enumerate(zip(got, want)):
                _records_equal(a, b, f"rep {rep} item {i}")
        assert c.analyze_batch([]) == []
        one = c.analyze_batch([clips[0][:1]])             # a one-frame clip: no pair at all
        assert one[0]["ham"][0] == -1 and one[0]["flow_mean"][0] == 0


def test_batch_larger_than_the_farneback_scratch(oracle):
    """More pairs than the scratch holds at most (512): the call is processed in chunks with a one-frame overlap, also when
    a chunk boundary falls on a clip boundary."""
    import avd_hip
    base = synth.make_clip(8, 64, 96, seed=31, dup_every=0)
    clips = [np.ascontiguousarray(base[np.arange(n) % 8]) for n in (300, 213, 1, 140)]   # 654 frames; boundary at 513 = chunk edge
    with avd_hip.Context(0) as c:
        got = c.analyze_batch(clips)
        want = [c.analyze_frames(x) for x in clips]
        for i, (a, b) in enumerate(zip(got, want)):
            _records_equal(a, b, f"clip {i}")


def test_clips_in_flight_with_batches():
    import avd_hip
    clips = [synth.make_clip(6, 180, 320, seed=s, dup_every=0) for s in range(6)]
    runner = avd_hip.ClipsInFlight(device=0, depth=2)
    with avd_hip.Context(0) as c:
        want = [c.analyze_frames(x) for x in clips]
    runner.submit_batch(clips[:3], tag="a")
    runner.submit_batch(clips[3:], tag="b")
    for tag, recs in (runner.drain(), runner.drain()):
        off = 0 if tag == "a" else 3
        for i, r in enumerate(recs):
            _records_equal(r, want[off + i], f"{tag}{i}")
    for cx in runner.ctxs:
        cx.close()
