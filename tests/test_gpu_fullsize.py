"""BASELINE.json configs at their FULL sizes, through the product path, against the oracle chain.

  configs[0]  "single 10 s 720p30 clip through /analyze": 300 decoded frames, step 15 -> 20 sampled frames, tlen 10;
              from a decoded-frames .npy and from a .y4m (decoder pictures: the NV12 ingest kernel does the colour
              conversion) through ``pipeline.analyze_path`` (reference api.py:142-170): the whole JSON body.
  configs[3]  "4K30 120 s clip, dense 8 fps sampling": 3600 decoded frames, step 4 -> 900 sampled 4K frames (22.4 GB,
              built on the GPU from six base frames), device resident in one call and streamed through ``video.analyze``.

Both Farneback modes: exact (bit-identical) and fast (the default; tolerance 1e-6 on ai_susp, in practice identical).
Reference sites: app/analyzers/video.py:14-19 (sampler), :27-58 (loop), :61-83 (summary / timeline), fusion.py:16-109."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fresh_pool(monkeypatch, mode):
    """video.analyze borrows contexts from the default pool, and a context reads AVD_FB_MODE when it is created."""
    from avd_hip import analyzer
    monkeypatch.setenv("AVD_FB_MODE", mode)
    pool = analyzer.ContextPool()
    monkeypatch.setattr(analyzer, "_pool", pool)
    return pool


def _expected_body(oracle, sampled, meta, path):
    from app.analyzers import fusion, heuristics_v2
    from avd_hip import pipeline
    video = oracle.analyze_sampled_frames(sampled, meta)
    hints = heuristics_v2.compute_hints(dict(meta), path)
    fused = fusion.fuse(pipeline.audio_unavailable(path, meta), video, hints)       # pads video["timeline"] in place, as api.py:148 does
    return video, hints, fused


def _check_body(body, want_video, hints, fused, mode):
    assert list(body.keys()) == ["ok", "meta", "hints", "video", "audio", "result", "timeline_binned", "peaks"]
    assert body["ok"] is True and "video_error" not in body["hints"]
    assert len(body["video"]["timeline"]) == 10 and body["video"]["timeline"] is body["video"]["timeline_ai"]
    if mode == "exact":
        assert body["video"]["timeline"] == want_video["timeline"]
        assert body["timeline_binned"] == fused["timeline_binned"] and body["peaks"] == fused["peaks"]
    else:
        np.testing.assert_allclose(body["video"]["timeline"], want_video["timeline"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(body["timeline_binned"], fused["timeline_binned"], rtol=0, atol=1e-6)
        print("[fullsize] fast mode: timeline identical =", body["video"]["timeline"] == want_video["timeline"])
    assert body["video"]["summary"] == pytest.approx(want_video["summary"], rel=1e-6 if mode == "fast" else 1e-12, abs=1e-12)
    assert body["result"] == fused["result"]
    for k, v in hints.items():
        assert body["hints"][k] == v, k
    assert body["audio"]["timeline"] == [0.5] * 10 and "error" in body["audio"]["flags_audio"]


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_cfg0_10s_720p30_npy_through_analyze_path(tmp_path, oracle, monkeypatch, mode):
    from avd_hip import pipeline, synth
    pool = _fresh_pool(monkeypatch, mode)
    n, h, w = 20, 720, 1280
    sampled = synth.make_clip(n, h, w, seed=40, dup_every=6)
    path = str(tmp_path / "clip10s.npy")
    full = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint8, shape=(300, h, w, 3))      # 300 decoded frames, sparse
    for i in range(n):
        full[15 * i] = sampled[i]                                       # only the frames the reference retrieves (video.py:31-32)
    full.flush()
    del full
    meta = {"width": w, "height": h, "fps": 30.0, "duration": 10.0, "bit_rate": 4_000_000, "vcodec": "h264", "acodec": None,
            "format_name": "mov,mp4"}
    body = pipeline.analyze_path(path, meta)
    want_video, hints, fused = _expected_body(oracle, sampled, meta, path)
    _check_body(body, want_video, hints, fused, mode)
    assert body["video"]["summary"]["w"] == w and body["video"]["summary"]["fps"] == 30.0
    pool.close()


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_cfg0_10s_720p30_y4m_through_analyze_path(tmp_path, oracle, monkeypatch, mode):
    from avd_hip import pipeline, sources, synth
    pool = _fresh_pool(monkeypatch, mode)
    n, h, w = 20, 720, 1280
    ys, cs = synth.bgr_to_nv12(synth.make_clip(n, h, w, seed=41, dup_every=6))
    y = np.zeros((300, h, w), np.uint8)
    uv = np.full((300, h // 2, w), 128, np.uint8)
    y[::15], uv[::15] = ys, cs
    path = str(tmp_path / "clip10s.y4m")
    sources.write_y4m(path, y, uv, fps=(30, 1))
    meta = {"width": 0, "height": 0, "fps": 0.0, "duration": 0.0, "bit_rate": 0}       # ffprobe missing: the file's own header is used
    body = pipeline.analyze_path(path, meta)
    sampled = oracle.nv12_to_bgr(ys, cs)
    from app.analyzers import fusion
    want_video = oracle.analyze_sampled_frames(sampled, {"width": w, "height": h, "fps": 30.0, "duration": 10.0})
    full_meta = {**meta, "vcodec": None, "acodec": None, "format_name": None}
    hints = pipeline.hx.compute_hints(full_meta, path)
    ref = dict(want_video, timeline=list(want_video["timeline"]))
    fused = fusion.fuse(pipeline.audio_unavailable(path, full_meta), ref, hints)        # audio: tlen of the (zero) meta duration
    assert len(want_video["timeline"]) == 10
    if mode == "exact":
        assert body["video"]["timeline"][:10] == want_video["timeline"]
        assert body["timeline_binned"] == fused["timeline_binned"]
    else:
        np.testing.assert_allclose(body["video"]["timeline"][:10], want_video["timeline"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(body["timeline_binned"], fused["timeline_binned"], rtol=0, atol=1e-6)
    assert body["video"]["summary"] == pytest.approx(want_video["summary"], rel=1e-6, abs=1e-12)
    assert body["result"] == fused["result"] and "video_error" not in body["hints"]
    pool.close()


# ---- configs[3] at 900 sampled 4K frames --------------------------------------------------------------------------------
def _pattern(n, nbase):
    out = []
    for k in range(n):
        out.append(out[-1] if (k and k % 10 == 0) else (k * 5 + k // 7) % nbase)
    return np.array(out)


def _oracle_records_from_pattern(oracle, base, pattern):
    import avd_hip
    small, hsh, s, q = oracle.preprocess_bgr(base)
    want = np.zeros(len(pattern), avd_hip.RECORD_DTYPE)
    want["lap_sum"], want["lap_sumsq"] = s[pattern], q[pattern]
    want["ham"][0] = -1
    cache = {}
    for k in range(1, len(pattern)):
        a, b = int(pattern[k - 1]), int(pattern[k])
        if (a, b) not in cache:
            fm, fv = oracle.farneback_pairs(np.stack([small[a], small[b]]))
            cache[(a, b)] = (fm[0], fv[0], int(np.sum(hsh[a] ^ hsh[b])))
        want["flow_mean"][k], want["flow_var"][k], want["ham"][k] = cache[(a, b)]
    return want


def test_cfg3_900_device_resident_4k_frames(oracle):
    """120 s of 4K30 at 8 analysed frames per second: 900 frames = 22.4 GB resident in HBM, ONE call.  Every record is
    compared with the oracle (the per-pair values only depend on the two frames, and six base frames give at most 36
    distinct pairs); 899 pairs cross the Farneback scratch boundary (512 pairs) once."""
    torch = pytest.importorskip("torch")
    import avd_hip
    from avd_hip import synth
    from avd_hip.timeline import records_to_result
    n, h, w = 900, 2160, 3840
    base = synth.make_clip(6, h, w, seed=3, dup_every=0)
    pattern = _pattern(n, len(base))
    want = _oracle_records_from_pattern(oracle, base, pattern)
    dev_base = torch.from_numpy(base).to("cuda:0")
    clip = dev_base[torch.from_numpy(pattern).to("cuda:0")]              # 900 x 24.9 MB, a fresh tensor
    assert clip.shape == (n, h, w, 3) and clip.is_contiguous()
    ref = records_to_result(want, h * w, w, h, 30.0, 120.0)
    for mode in (0, 1):
        with avd_hip.Context(0) as c:
            c.set_option("fb_mode", mode)
            rec = c.analyze_frames(clip)
            for key in ("lap_sum", "lap_sumsq", "ham"):
                assert np.array_equal(rec[key], want[key]), (mode, key)
            if mode == 0:
                assert np.array_equal(rec["flow_mean"], want["flow_mean"]) and np.array_equal(rec["flow_var"], want["flow_var"])
            else:
                np.testing.assert_allclose(rec["flow_mean"], want["flow_mean"], rtol=1e-6, atol=1e-7)
                np.testing.assert_allclose(rec["flow_var"], want["flow_var"], rtol=1e-6, atol=1e-7)
            got = records_to_result(rec, h * w, w, h, 30.0, 120.0)
            np.testing.assert_allclose(got["timeline"], ref["timeline"], rtol=0, atol=1e-6)
            assert len(got["timeline"]) == 120
            # size-independent properties: locality across the scratch boundary, reversal
            sub = c.analyze_frames(clip[500:530])
            for key in ("flow_mean", "flow_var", "ham"):
                assert np.array_equal(sub[key][1:], rec[key][501:530]), (mode, key)
            rev = c.analyze_frames(torch.flip(clip[:300], dims=[0]))
            assert np.array_equal(rev["lap_sumsq"], rec["lap_sumsq"][:300][::-1])
            assert np.array_equal(rev["ham"][1:], rec["ham"][1:300][::-1])
    del clip, dev_base


def test_cfg3_900_frames_streamed_through_video_analyze(oracle, monkeypatch):
    """The same clip through the drop-in ``video.analyze`` (host frames, chunks of 128 sampled frames with a one-frame halo)."""
    from app.analyzers import video
    from avd_hip import sources, synth
    from avd_hip.timeline import records_to_result
    from tests.test_gpu_configs import _Synthetic4KSource
    pool = _fresh_pool(monkeypatch, "exact")
    n, h, w = 900, 2160, 3840
    base = synth.make_clip(6, h, w, seed=3, dup_every=0)
    pattern = _pattern(n, len(base))
    src = _Synthetic4KSource(base, pattern)
    monkeypatch.setattr(sources, "open_source", lambda path: src)
    monkeypatch.setenv("AVD_SAMPLES_PER_SECOND", "8")
    monkeypatch.setenv("AVD_CHUNK_FRAMES", "128")
    meta = {"width": w, "height": h, "fps": 30.0, "duration": 120.0}
    got = video.analyze("synthetic-4k30-120s.mp4", meta)
    assert src.steps_seen == [4] and src.closed
    want = _oracle_records_from_pattern(oracle, base, pattern)
    ref = records_to_result(want, h * w, w, h, 30.0, 120.0)
    assert got["timeline"] == ref["timeline"] and len(got["timeline"]) == 120
    assert got["summary"] == pytest.approx(ref["summary"], rel=1e-12, abs=1e-12)
    pool.close()
