"""Orchestration layer (reference api._analyze_path data path): CPU-side behaviours, and one
GPU end-to-end run from a decoded-frames file to the /analyze JSON body."""
import json

import numpy as np
import pytest

from avd_hip import pipeline, synth

KEYS = ["ok", "meta", "hints", "video", "audio", "result", "timeline_binned", "peaks"]


def test_unopenable_file_gives_reference_neutral_body(tmp_path):
    """Empty ffprobe meta + capture not opened.  fuse() alone gives 0.45 for empty hints;
    through the pipeline the zero metadata also classifies as very_heavy compression with bpp 0, so both
    quality penalties apply: 0.5 * (0.39 + 0.51 + 0.10) - 0.10 = 0.40."""
    body = pipeline.analyze_path(str(tmp_path / "nope.mp4"))
    assert list(body.keys()) == KEYS
    assert body["video"] == {"timeline": [], "summary": {}, "timeline_ai": []}      # empty lists are falsy: fuse pads a copy
    assert body["result"] == {"label": "uncertain", "ai_score": 0.4, "confidence": 0.1, "reason": "segnali misti o neutri"}
    assert body["timeline_binned"] == [0.4] and body["peaks"] == []
    assert body["meta"]["source_url"] is None and body["hints"]["compression"] == "very_heavy"
    json.dumps(body)                                                     # JSON-serialisable, plain Python types


def test_video_exception_becomes_neutral_timeline():
    def boom(path, meta):
        raise RuntimeError("device lost")

    body = pipeline.analyze_path("x.mp4", {"width": 1920, "height": 1080, "fps": 30.0, "duration": 4.4, "bit_rate": 8_000_000},
                                 video_analyzer=boom)
    assert body["hints"]["video_error"] == "RuntimeError"
    assert body["video"]["summary"] == {"error": "RuntimeError"}
    assert body["video"]["timeline"] == [0.5] * 4 and body["audio"]["timeline"] == [0.5] * 4
    assert "video_traceback" not in body["hints"]
    dbg = pipeline.analyze_path("x.mp4", {"duration": 1.0}, video_analyzer=boom, debug=True)
    assert "device lost" in dbg["hints"]["video_traceback"]


def test_audio_plugin_and_url_fields():
    audio = lambda p, m: {"scores": {"tts_like": 0.97}, "flags_audio": {"speech_ratio": 0.9, "tts_like": 0.97},
                          "timeline": [0.9, 0.95, 0.92]}
    video = lambda p, m: {"timeline": [0.8, 0.9, 0.85], "summary": {"dup_density": 0.3}, "timeline_ai": [0.8, 0.9, 0.85]}
    body = pipeline.analyze_path("x.mp4", {"duration": 3.0}, audio_analyzer=audio, video_analyzer=video,
                                 source_url="https://e/x", resolved_url="https://cdn/x.mp4")
    assert body["result"]["label"] == "ai" and "audio TTS-like elevato" in body["result"]["reason"]
    assert body["meta"]["source_url"] == "https://e/x" and body["meta"]["resolved_url"] == "https://cdn/x.mp4"


@pytest.mark.gpu
def test_analyze_path_end_to_end_on_gpu(tmp_path, oracle):
    """decoded-frames file -> sampler (step 15) -> HIP kernels -> fusion, against the oracle chain."""
    from app.analyzers import fusion, heuristics_v2
    sampled = synth.make_clip(4, 180, 320, seed=17, dup_every=3)
    full = np.repeat(sampled, 15, axis=0)[: 3 * 15 + 1]                  # 46 decoded frames, every 15th sampled
    path = tmp_path / "clip.npy"
    np.save(path, full)
    meta = {"width": 320, "height": 180, "fps": 30.0, "duration": 46 / 30.0, "bit_rate": 2_000_000}
    body = pipeline.analyze_path(str(path), meta)
    want_video = oracle.analyze_sampled_frames(sampled, meta)
    hints = heuristics_v2.compute_hints({**meta, "vcodec": None, "acodec": None, "format_name": None}, str(path))
    want = fusion.fuse(pipeline.audio_unavailable(str(path), meta), want_video, hints)
    assert body["video"]["timeline"] == want_video["timeline"]
    assert body["video"]["summary"] == pytest.approx(want_video["summary"], rel=1e-12, abs=1e-12)
    assert body["result"] == want["result"] and body["timeline_binned"] == want["timeline_binned"]
    assert "video_error" not in body["hints"]
