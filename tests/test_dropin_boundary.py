"""Drop-in boundary on the host side (CPU): module-path resolution next to a reference checkout
(reference api.py:14-18), the forensic passthrough (api.py:163-169), ffprobe metadata parsing
(api.py:46-89) and the bounded context pool behind ``app.analyzers.video.analyze`` (api.py:133 runs it
on worker threads)."""
import json
import os
import stat
import subprocess
import sys
import textwrap
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ai-video-detector_amd")

# exactly the import block of reference api.py:14-18
API_IMPORTS = textwrap.dedent("""
    from app.analyzers import audio as audio_an
    from app.analyzers import video as video_an
    from app.analyzers import fusion as fusion_an
    from app.analyzers import heuristics_v2 as hx
    from app.analyzers import meta as meta_an
""")


def _fake_reference_tree(base):
    """The reference's layout with stub modules: every module also exists in the hot-path subset, so the test
    can tell which copy was imported."""
    an = base / "app" / "analyzers"
    an.mkdir(parents=True)
    (base / "app" / "__init__.py").write_text("")
    # the reference's package init imports all six analyzers eagerly; it must never run when the drop-in is first
    (an / "__init__.py").write_text("raise ImportError('the reference package init ran: cv2/soundfile would be needed')\n")
    for name in ("audio", "meta", "forensic", "video", "fusion", "heuristics_v2"):
        (an / f"{name}.py").write_text(f"ORIGIN = 'reference-stub'\nNAME = {name!r}\n"
                                       "def forensic_summary(path):\n    return {'c2pa': {'present': False}, 'path': path}\n")
    return base


def _run(code, pythonpath):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(pythonpath))
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)


def test_api_import_block_resolves_next_to_a_reference_checkout(tmp_path):
    ref = _fake_reference_tree(tmp_path / "ref")
    code = API_IMPORTS + textwrap.dedent("""
        import json, os
        print(json.dumps({"audio": audio_an.__file__, "video": video_an.__file__, "fusion": fusion_an.__file__,
                          "hints": hx.__file__, "meta": meta_an.__file__,
                          "forensic": meta_an.forensic_summary("x.mp4")}))
    """)
    r = _run(code, [PKG, str(ref)])
    assert r.returncode == 0, r.stderr
    where = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("video", "fusion", "hints", "audio"):              # this build (audio: SURVEY.md 8f row N3)
        assert where[key].startswith(PKG), (key, where[key])
    assert where["meta"].startswith(str(ref)), where["meta"]       # everything else: the reference's own modules
    assert where["forensic"]["path"] == "x.mp4"


def test_without_a_reference_checkout_only_the_hot_path_modules_exist():
    r = _run("from app.analyzers import video, fusion, heuristics_v2, audio\nfrom app.analyzers import meta", [PKG])
    assert r.returncode != 0 and "cannot import name 'meta'" in r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/app/analyzers"), reason="reference checkout not present")
def test_real_reference_modules_resolve_behind_the_dropin():
    """With the real checkout behind the package: meta / forensic import from it (stdlib only); video and audio are
    this build's (the reference's audio.py would need soundfile, which this image lacks)."""
    code = textwrap.dedent("""
        from app.analyzers import video, fusion, heuristics_v2, meta, forensic, audio
        print(meta.__file__); print(forensic.__file__); print(video.__file__); print(audio.__file__)
    """)
    r = _run(code, [PKG, "/root/reference"])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "/root/reference/app/analyzers/meta.py" and lines[1] == "/root/reference/app/analyzers/forensic.py"
    assert lines[2].startswith(PKG)
    assert lines[3].startswith(PKG)


# ---- forensic passthrough (api.py:163-169) --------------------------------------------------
def _video(path, meta):
    return {"timeline": [0.4, 0.6], "summary": {}, "timeline_ai": [0.4, 0.6]}


def test_forensic_passthrough():
    from avd_hip import pipeline
    meta = {"duration": 2.0}
    body = pipeline.analyze_path("x.mp4", meta, video_analyzer=_video, forensic=lambda p: {"c2pa": {"present": True}, "p": p})
    assert list(body.keys())[-1] == "forensic" and body["forensic"] == {"c2pa": {"present": True}, "p": "x.mp4"}
    assert "forensic" not in pipeline.analyze_path("x.mp4", meta, video_analyzer=_video, forensic=lambda p: {})

    def boom(p):
        raise OSError("exiftool missing")

    quiet = pipeline.analyze_path("x.mp4", meta, video_analyzer=_video, forensic=boom)
    assert "forensic" not in quiet and "forensic_error" not in quiet and quiet["ok"] is True
    loud = pipeline.analyze_path("x.mp4", meta, video_analyzer=_video, forensic=boom, debug=True)
    assert "exiftool missing" in loud["forensic_error"]


# ---- ffprobe metadata (api.py:46-89) ---------------------------------------------------------
def _fake_ffprobe(tmp_path, report):
    bindir = tmp_path / "bin"
    bindir.mkdir()
    exe = bindir / "ffprobe"
    exe.write_text("#!/bin/sh\ncat <<'EOF'\n" + (report if isinstance(report, str) else json.dumps(report)) + "\nEOF\n")
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    return str(bindir)


def test_probe_basic_meta_parses_like_the_reference(tmp_path, monkeypatch):
    from avd_hip import pipeline
    report = {"streams": [{"codec_type": "video", "width": 0, "height": 0, "r_frame_rate": "0/0", "codec_name": "mjpeg"},
                          {"codec_type": "audio", "codec_name": "aac"},
                          {"codec_type": "video", "width": 1920, "height": 1080, "r_frame_rate": "30000/1001", "codec_name": "h264"},
                          {"codec_type": "video", "width": 640, "height": 360, "r_frame_rate": "25/1", "codec_name": "vp9"},
                          {"codec_type": "audio", "codec_name": "opus"}],
              "format": {"bit_rate": "8000000.0", "duration": "60.04", "format_name": "mov,mp4"}}
    monkeypatch.setenv("PATH", _fake_ffprobe(tmp_path, report) + os.pathsep + os.environ["PATH"])
    meta = pipeline.probe_basic_meta("whatever.mp4")
    assert list(meta.keys()) == list(pipeline.META_KEYS)
    # the zero-width cover-art stream is overwritten by the first real one; later streams are ignored
    assert meta == {"width": 1920, "height": 1080, "fps": 30000 / 1001, "duration": 60.04, "bit_rate": 8000000,
                    "vcodec": "h264", "acodec": "aac", "format_name": "mov,mp4"}
    assert type(meta["width"]) is int and type(meta["bit_rate"]) is int and type(meta["fps"]) is float


def test_probe_basic_meta_fallbacks(tmp_path, monkeypatch):
    from avd_hip import pipeline
    zero = {"width": 0, "height": 0, "fps": 0.0, "duration": 0.0, "bit_rate": 0, "vcodec": None, "acodec": None,
            "format_name": None}
    monkeypatch.setenv("PATH", str(tmp_path))                          # no ffprobe at all
    assert pipeline.probe_basic_meta("x") == zero
    monkeypatch.setenv("PATH", _fake_ffprobe(tmp_path, "this is not json") + os.pathsep + "/usr/bin:/bin")
    assert pipeline.probe_basic_meta("x") == zero
    sub = tmp_path / "b"
    sub.mkdir()
    rep = {"streams": [{"codec_type": "video", "width": 1280, "height": 720, "r_frame_rate": "junk"}],
           "format": {"duration": "N/A"}}
    monkeypatch.setenv("PATH", _fake_ffprobe(sub, rep) + os.pathsep + "/usr/bin:/bin")
    m = pipeline.probe_basic_meta("x")
    assert (m["width"], m["height"], m["fps"], m["duration"], m["bit_rate"]) == (1280, 720, 0.0, 0.0, 0)


# ---- bounded context pool ---------------------------------------------------------------------
class _FakeCtx:
    made = 0

    def __init__(self, device):
        type(self).made += 1
        self.device, self.released, self.closed = device, 0, False

    def release_workspace(self):
        self.released += 1

    def synchronize(self):
        pass

    def close(self):
        self.closed = True


def test_context_pool_is_bounded_and_trims_idle_workspaces(monkeypatch):
    from avd_hip import analyzer
    monkeypatch.setattr(analyzer._lib, "Context", _FakeCtx)
    _FakeCtx.made = 0
    pool = analyzer.ContextPool(max_contexts=3, keep_warm=1)
    peak, live, lock = [0], [0], threading.Lock()
    gate = threading.Barrier(8)

    def request():
        gate.wait()
        for _ in range(5):
            with pool.borrow(0) as ctx:
                with lock:
                    live[0] += 1
                    peak[0] = max(peak[0], live[0])
                assert not ctx.closed
                threading.Event().wait(0.002)
                with lock:
                    live[0] -= 1

    ts = [threading.Thread(target=request) for _ in range(8)]           # more worker threads than contexts
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in ts)
    assert _FakeCtx.made <= 3 and peak[0] <= 3                          # never more than max_contexts, whatever the threads do
    st = pool.stats(0)
    assert st["created"] == _FakeCtx.made and st["warm"] + st["cold"] == _FakeCtx.made and st["warm"] == 1
    warm, cold = list(pool._warm[0]), list(pool._cold[0])
    assert all(c.released for c in cold)                                # only keep_warm idle contexts keep their scratch
    pool.close()
    assert all(c.closed for c in warm + cold)


def test_context_pool_survives_a_failing_constructor(monkeypatch):
    from avd_hip import analyzer

    class Boom:
        def __init__(self, device):
            raise analyzer._lib.AvdError("no usable HIP device")

    monkeypatch.setattr(analyzer._lib, "Context", Boom)
    pool = analyzer.ContextPool(max_contexts=1)
    for _ in range(3):                                                  # the slot is given back each time: no deadlock
        with pytest.raises(analyzer._lib.AvdError):
            with pool.borrow(0):
                pass
    assert pool.stats(0)["created"] == 0


def test_context_pool_drops_a_context_whose_release_fails(monkeypatch, caplog):
    """A context whose stream reports an error when its scratch is released (a sticky HIP error after a fault) is logged, closed and
    NOT handed to a later request; its slot is free again, so the pool creates a fresh context."""
    from avd_hip import analyzer

    class Sick(_FakeCtx):
        def release_workspace(self):
            raise analyzer._lib.AvdError("hipStreamSynchronize: an illegal memory access was encountered")

    monkeypatch.setattr(analyzer._lib, "Context", Sick)
    Sick.made = 0
    pool = analyzer.ContextPool(max_contexts=2, keep_warm=1)
    with caplog.at_level("WARNING", logger="avd_hip"):
        with pool.borrow(0) as a:
            with pool.borrow(0) as b:
                pass
    # two contexts came back, one stays warm, the other's release failed: closed, gone from the pool, slot returned
    st = pool.stats(0)
    assert st == {"created": 1, "warm": 1, "cold": 0}
    assert sum(c.closed for c in (a, b)) == 1 and "release_workspace failed" in caplog.text
    with pool.borrow(0) as c1, pool.borrow(0) as c2:                       # a fresh one can be created again
        assert not c1.closed and not c2.closed
    pool.close()


def test_context_pool_drops_the_context_whose_own_call_faulted(monkeypatch, caplog):
    """The realistic path of a device fault: the BORROWER's call raises and the context's stream keeps reporting the (sticky) error.  That
    context must not go back to the most-recently-used end of the warm list, where the next request would pick it: it is probed, closed and its
    slot returned.  An ordinary failure of the request (bad argument) leaves a healthy context, which is reused."""
    from avd_hip import analyzer

    class Faulty(_FakeCtx):
        sick = False

        def synchronize(self):
            if self.sick:
                raise analyzer._lib.AvdError("hipStreamSynchronize: an illegal memory access was encountered")

    monkeypatch.setattr(analyzer._lib, "Context", Faulty)
    Faulty.made = 0
    pool = analyzer.ContextPool(max_contexts=2, keep_warm=2)
    with caplog.at_level("WARNING", logger="avd_hip"):
        with pytest.raises(analyzer._lib.AvdError):
            with pool.borrow(0) as a:
                a.sick = True
                raise analyzer._lib.AvdError("avd status -2: hipLaunchKernel: an illegal memory access was encountered")
    assert a.closed and pool.stats(0) == {"created": 0, "warm": 0, "cold": 0} and "dropped from the pool" in caplog.text
    with pytest.raises(ValueError):
        with pool.borrow(0) as b:                                          # a fresh context, whose request fails for an ordinary reason
            assert b is not a
            raise ValueError("frames must be uint8[N,H,W,3] (BGR)")
    assert not b.closed and pool.stats(0) == {"created": 1, "warm": 1, "cold": 0}
    with pool.borrow(0) as c:
        assert c is b                                                      # healthy: reused
    pool.close()
