"""Frame sources of the drop-in ``video.analyze`` (reference app/analyzers/video.py:11-17,
27-33, 59).  No decoder exists in the build image, so the cv2 and ffmpeg sources are exercised against FAKES that speak
the same interface: a stand-in ``cv2`` module (VideoCapture with grab / retrieve / get / release) and stand-in
``ffprobe`` / ``ffmpeg`` executables that emit JSON and raw bgr24 frames.  What is checked is the reference's contract:
every frame is grabbed, only every ``step``-th is retrieved, capture properties are the fallback for missing metadata,
an unopenable file is not an error."""
import contextlib
import json
import os
import stat
import sys
import types

import numpy as np
import pytest

from avd_hip import sources


def _frames(t, h, w):
    return (np.arange(t * h * w * 3, dtype=np.uint32) % 251).astype(np.uint8).reshape(t, h, w, 3)


def _fake_cv2(frames, fps=25.0, opened=True, retrieve_fails_at=None):
    log = {"grabs": 0, "retrieves": [], "released": 0}
    mod = types.ModuleType("cv2")
    mod.CAP_PROP_FPS, mod.CAP_PROP_FRAME_WIDTH, mod.CAP_PROP_FRAME_HEIGHT, mod.CAP_PROP_FRAME_COUNT = 5, 3, 4, 7

    class VideoCapture:
        def __init__(self, path):
            self.path, self.pos = path, -1

        def isOpened(self):
            return opened

        def get(self, prop):
            return {5: fps, 3: float(frames.shape[2]), 4: float(frames.shape[1]), 7: float(len(frames))}[prop]

        def grab(self):
            if self.pos + 1 >= len(frames):
                return False
            self.pos += 1
            log["grabs"] += 1
            return True

        def retrieve(self):
            log["retrieves"].append(self.pos)
            if retrieve_fails_at is not None and self.pos == retrieve_fails_at:
                return False, None
            return True, frames[self.pos].copy()

        def release(self):
            log["released"] += 1

    mod.VideoCapture = VideoCapture
    return mod, log


@contextlib.contextmanager
def _module(name, mod):
    saved = sys.modules.get(name)
    sys.modules[name] = mod
    try:
        yield
    finally:
        if saved is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = saved


def test_cv2_source_grabs_everything_and_retrieves_every_step():
    fr = _frames(40, 6, 8)
    mod, log = _fake_cv2(fr, fps=25.0)
    with _module("cv2", mod):
        src = sources.open_source("clip.mp4")
        assert isinstance(src, sources.Cv2Source)
        assert (src.fps, src.width, src.height, src.frame_count) == (25.0, 8, 6, 40)
        got = list(src.sampled(12))                                   # fps 25 -> step 12 (video.py:19)
        src.close()
    assert log["grabs"] == 40 and log["retrieves"] == [0, 12, 24, 36] and log["released"] == 1
    assert all(np.array_equal(g, fr[i]) for g, i in zip(got, (0, 12, 24, 36)))


def test_cv2_capture_not_opened_and_failed_retrieve():
    mod, _ = _fake_cv2(_frames(5, 4, 4), opened=False)
    with _module("cv2", mod):
        assert sources.open_source("missing.mp4") is None            # video.py:12-13: not an error
    mod, log = _fake_cv2(_frames(30, 4, 4), retrieve_fails_at=15)
    with _module("cv2", mod):
        src = sources.open_source("clip.mp4")
        got = list(src.sampled(15))                                   # `if not ok: break` (video.py:33)
    assert len(got) == 1 and log["retrieves"] == [0, 15]


def _fake_ffmpeg(tmp_path, frames, fps="30000/1001"):
    bindir = tmp_path / "bin"
    bindir.mkdir()
    raw = tmp_path / "frames.raw"
    raw.write_bytes(frames.tobytes())
    probe = {"streams": [{"width": frames.shape[2], "height": frames.shape[1], "r_frame_rate": fps, "nb_frames": str(len(frames))}]}
    (bindir / "ffprobe").write_text("#!/bin/sh\ncat <<'EOF'\n" + json.dumps(probe) + "\nEOF\n")
    (bindir / "ffmpeg").write_text(f"#!/bin/sh\ncat '{raw}'\n")
    for exe in ("ffprobe", "ffmpeg"):
        p = bindir / exe
        p.chmod(p.stat().st_mode | stat.S_IEXEC)
    return str(bindir)


def test_ffmpeg_pipe_source(tmp_path, monkeypatch):
    fr = _frames(33, 4, 6)
    monkeypatch.setenv("PATH", _fake_ffmpeg(tmp_path, fr) + os.pathsep + "/usr/bin:/bin")
    monkeypatch.setitem(sys.modules, "cv2", None)                    # `import cv2` -> ImportError: fall through to the CLI source
    src = sources.open_source("clip.mkv")
    assert isinstance(src, sources.FfmpegSource)
    assert (src.width, src.height, src.frame_count) == (6, 4, 33) and src.fps == pytest.approx(29.97, abs=1e-3)
    got = list(src.sampled(15))
    src.close()
    assert len(got) == 3 and all(np.array_equal(g, fr[i]) for g, i in zip(got, (0, 15, 30)))


def test_no_decoder_at_all_means_capture_not_opened(monkeypatch, tmp_path):
    monkeypatch.setenv("PATH", str(tmp_path))
    monkeypatch.setitem(sys.modules, "cv2", None)
    assert sources.open_source("clip.mp4") is None


def test_analyze_uses_capture_properties_when_metadata_is_missing(monkeypatch):
    """video.py:14-17: meta first, capture properties as the fallback; duration = frame_count / fps."""
    import avd_hip
    from app.analyzers import video
    from avd_hip import analyzer
    fr = _frames(50, 40, 48)
    mod, log = _fake_cv2(fr, fps=25.0)
    seen = {}

    class FakeAnalyzer:
        def __init__(self, chunk=64, ctx=None):
            pass

        def records_stream(self, frames):
            got = list(frames)
            seen["n"] = len(got)
            rec = np.zeros(len(got), avd_hip.RECORD_DTYPE)
            rec["ham"][0] = -1
            return rec

    @contextlib.contextmanager
    def borrow(device=0):
        yield object()

    monkeypatch.setattr(analyzer, "FrameAnalyzer", FakeAnalyzer)
    monkeypatch.setattr(analyzer.default_pool(), "borrow", borrow)
    with _module("cv2", mod):
        out = video.analyze("clip.mp4", {"fps": 0, "width": None, "height": 0, "duration": 0.0})
    assert seen["n"] == 5 and log["retrieves"] == [0, 12, 24, 36, 48] and log["released"] == 1
    assert (out["summary"]["w"], out["summary"]["h"], out["summary"]["fps"]) == (48, 40, 25.0)
    assert len(out["timeline"]) == 2 and out["timeline"] is out["timeline_ai"]      # duration 50 / 25 = 2 s -> tlen 2
    with _module("cv2", mod):
        out = video.analyze("clip.mp4", {"fps": 30.0, "width": 1920, "height": 1080, "duration": 4.4})
    assert (out["summary"]["w"], out["summary"]["fps"]) == (1920, 30.0) and len(out["timeline"]) == 4   # meta wins; step 15


# ---- YUV4MPEG2: decoder pictures before the colour conversion (rows A1 + N1 together, no fakes) --------------------
def _write_y4m_clip(tmp_path, n=7, h=96, w=128, fps=(30, 1), seed=3):
    from avd_hip import synth
    clip = synth.make_clip(n, h, w, seed=seed, dup_every=3)
    y, uv = synth.bgr_to_nv12(clip)
    path = str(tmp_path / "clip.y4m")
    sources.write_y4m(path, y, uv, fps=fps)
    return path, y, uv


def test_y4m_source_reads_back_what_was_written(tmp_path):
    path, y, uv = _write_y4m_clip(tmp_path, n=7, fps=(30000, 1001))
    src = sources.open_source(path)
    assert isinstance(src, sources.Y4mSource) and src.surface == "nv12"
    assert (src.width, src.height, src.frame_count) == (128, 96, 7) and abs(src.fps - 29.97) < 0.01
    got = list(src.sampled(3))                                         # frames 0, 3, 6
    assert len(got) == 3
    for (gy, guv), i in zip(got, (0, 3, 6)):
        assert np.array_equal(gy, y[i]) and np.array_equal(guv, uv[i])
    src.close()


def test_y4m_rejects_what_it_cannot_read(tmp_path):
    bad = tmp_path / "bad.y4m"
    bad.write_bytes(b"RIFF....")
    assert sources.open_source(str(bad)) is None                       # "capture not opened", not an exception
    deep = tmp_path / "deep.y4m"
    deep.write_bytes(b"YUV4MPEG2 W64 H64 F25:1 C420p10\n")
    assert sources.open_source(str(deep)) is None
    odd = tmp_path / "odd.y4m"
    odd.write_bytes(b"YUV4MPEG2 W63 H64 F25:1 C420\n")
    assert sources.open_source(str(odd)) is None
    cut = tmp_path / "cut.y4m"                                          # a truncated last picture is dropped
    path, y, uv = _write_y4m_clip(tmp_path, n=3)
    data = open(path, "rb").read()
    cut.write_bytes(data[:-100])
    src = sources.open_source(str(cut))
    assert src.frame_count == 2


@pytest.mark.gpu
def test_analyze_a_y4m_file_end_to_end(tmp_path, oracle, monkeypatch):
    """video.analyze on a .y4m: the NV12 ingest kernel does the colour conversion; the result equals the oracle's chain
    swscale-style NV12 -> BGR, then video.py on the sampled frames -- across a streaming chunk boundary with its halo."""
    from avd_hip import synth
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ai-video-detector_amd"))
    from app.analyzers import video
    n, h, w = 40, 96, 128
    clip = synth.make_clip(n, h, w, seed=21, dup_every=4)
    y, uv = synth.bgr_to_nv12(clip)
    path = str(tmp_path / "clip.y4m")
    sources.write_y4m(path, y, uv, fps=(8, 1))                          # 8 fps: step = round(8 / 2) = 4 -> frames 0, 4, .., 36
    monkeypatch.setenv("AVD_CHUNK_FRAMES", "4")                        # 10 sampled frames in chunks of 4 + halo
    meta = {"width": 0, "height": 0, "fps": 0.0, "duration": 0.0}      # ffprobe missing: the source's properties are used
    got = video.analyze(path, meta)
    sampled = oracle.nv12_to_bgr(y[::4], uv[::4])
    want = oracle.analyze_sampled_frames(sampled, {"fps": 8.0, "width": w, "height": h, "duration": n / 8.0})
    assert got["timeline"] == want["timeline"] and got["timeline"] is got["timeline_ai"]
    assert got["summary"] == want["summary"]
