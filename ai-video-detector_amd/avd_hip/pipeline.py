"""Per-file orchestration around the video analyzer -- the data path of reference
``api._analyze_path`` (api.py:142-170) without the HTTP layer.

    meta  -> hints (heuristics_v2.compute_hints)
          -> audio result (pluggable; the audio analyzer is outside this build's scope)
          -> video result (app.analyzers.video.analyze, HIP kernels), with the reference's
             "safe" wrapper: ANY exception becomes the neutral 0.5 timeline + hints["video_error"]
             (api.py:130-140)
          -> fusion.fuse  -> the /analyze JSON body (same keys, same order, same Python types).

The HTTP routes themselves (upload spooling, yt-dlp, CORS, error bodies; api.py:213-279) are
control plane and stay the reference's: run the reference ``api.py`` with this package first on
PYTHONPATH and every route keeps working, backed by the MI355X analyzer.
"""
from __future__ import annotations

import json
import shutil
import subprocess
import traceback
from typing import Any, Callable, Dict, Optional

from app.analyzers import fusion as fusion_an
from app.analyzers import heuristics_v2 as hx
from app.analyzers import video as video_an

META_KEYS = ("width", "height", "fps", "duration", "bit_rate", "vcodec", "acodec", "format_name")


def probe_basic_meta(path: str) -> Dict[str, Any]:
    """ffprobe-based container metadata with the reference's fallbacks (api.py:46-89): when
    ffprobe is missing or fails every field is zero / None and the analyzers fall back to the
    frame source's own properties."""
    info: Dict[str, Any] = {}
    if shutil.which("ffprobe"):
        try:
            out = subprocess.check_output(
                ["ffprobe", "-v", "error", "-show_entries",
                 "format=bit_rate,duration,format_name:stream=codec_name,codec_type,width,height,r_frame_rate",
                 "-of", "json", path], text=True, stderr=subprocess.DEVNULL, timeout=30)
            info = json.loads(out)
        except Exception:
            info = {}
    width = height = fps = 0.0
    vcodec = acodec = None
    duration = 0.0
    for s in info.get("streams") or []:
        if s.get("codec_type") == "video" and not width:
            width = float(s.get("width") or 0)
            height = float(s.get("height") or 0)
            try:
                num, den = (s.get("r_frame_rate") or "0/1").split("/")
                fps = float(num) / max(1.0, float(den))
            except Exception:
                fps = 0.0
            vcodec = s.get("codec_name")
        elif s.get("codec_type") == "audio" and not acodec:
            acodec = s.get("codec_name")
    bit_rate, fmt = 0, None
    if info.get("format"):
        bit_rate = int(float(info["format"].get("bit_rate") or 0))
        fmt = info["format"].get("format_name")
        try:
            duration = float(info["format"].get("duration") or 0.0)
        except Exception:
            duration = 0.0
    return {"width": int(width), "height": int(height), "fps": fps, "duration": duration,
            "bit_rate": bit_rate, "vcodec": vcodec, "acodec": acodec, "format_name": fmt}


def _tlen(meta: dict) -> int:
    return int(max(1, round(meta.get("duration") or 0.0)))


def audio_unavailable(path: str, meta: dict) -> dict:
    """What the reference's audio analyzer returns when it cannot run (audio.py:112-118): a
    neutral timeline and an error flag.  Used when no audio analyzer is plugged in."""
    return {"scores": {}, "flags_audio": {"error": "audio analyzer not available in this build"},
            "timeline": [0.5] * _tlen(meta)}


def safe_call(kind: str, fn: Callable[[str, dict], dict], path: str, meta: dict, debug: bool = False):
    """reference _safe_audio/_safe_video (api.py:118-140): any exception -> neutral result."""
    extra: Dict[str, Any] = {}
    try:
        return fn(path, meta), extra
    except Exception as e:          # noqa: BLE001 -- the reference catches everything here
        tlen = _tlen(meta)
        name = str(e.__class__.__name__)
        if kind == "audio":
            neutral = {"scores": {}, "flags_audio": {"error": name}, "timeline": [0.5] * tlen}
        else:
            neutral = {"timeline": [0.5] * tlen, "summary": {"error": name}, "timeline_ai": [0.5] * tlen}
        extra[f"{kind}_error"] = name
        if debug:
            extra[f"{kind}_traceback"] = traceback.format_exc()
        return neutral, extra


def analyze_path(path: str, meta: Optional[dict] = None, *, audio_analyzer: Optional[Callable[[str, dict], dict]] = None,
                 video_analyzer: Optional[Callable[[str, dict], dict]] = None, source_url: Optional[str] = None,
                 resolved_url: Optional[str] = None, debug: bool = False) -> Dict[str, Any]:
    """The body of POST /analyze for one file (api.py:142-162), synchronous."""
    meta = dict(meta) if meta is not None else probe_basic_meta(path)
    for k in META_KEYS:
        meta.setdefault(k, None if k in ("vcodec", "acodec", "format_name") else 0)
    hints = hx.compute_hints(meta, path)
    audio, a_hint = safe_call("audio", audio_analyzer or audio_unavailable, path, meta, debug)
    video, v_hint = safe_call("video", video_analyzer or video_an.analyze, path, meta, debug)
    hints.update(a_hint)
    hints.update(v_hint)
    fused = fusion_an.fuse(audio, video, hints)
    return {
        "ok": True,
        "meta": {**meta, "source_url": source_url, "resolved_url": resolved_url},
        "hints": hints,
        "video": video,
        "audio": audio,
        "result": fused["result"],
        "timeline_binned": fused["timeline_binned"],
        "peaks": fused["peaks"],
    }
