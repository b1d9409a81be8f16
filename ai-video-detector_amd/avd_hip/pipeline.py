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
PYTHONPATH (the ``app`` package extends its search path to the reference checkout, so
``audio`` / ``meta`` still come from there) and every route keeps working, backed by the MI355X analyzer.
"""
from __future__ import annotations

import json
import shutil
import subprocess
import traceback
from typing import Any, Callable, Dict, Optional

from app.analyzers import audio as audio_an
from app.analyzers import fusion as fusion_an
from app.analyzers import heuristics_v2 as hx
from app.analyzers import video as video_an

META_KEYS = ("width", "height", "fps", "duration", "bit_rate", "vcodec", "acodec", "format_name")


def _ffprobe_json(path: str) -> Dict[str, Any]:
    """Raw ffprobe report, or {} when ffprobe is absent / fails / times out (api.py:46-56)."""
    if not shutil.which("ffprobe"):
        return {}
    fields = "format=bit_rate,duration,format_name:stream=codec_name,codec_type,width,height,r_frame_rate"
    try:
        report = subprocess.check_output(["ffprobe", "-v", "error", "-show_entries", fields, "-of", "json", path],
                                         text=True, stderr=subprocess.DEVNULL, timeout=30)
        return json.loads(report)
    except Exception:          # noqa: BLE001 -- the reference swallows everything here
        return {}


def _frame_rate(text) -> float:
    """'30000/1001' -> 29.97; the denominator is floored at 1 and anything unparsable is 0 (api.py:67-72)."""
    try:
        num, den = (text or "0/1").split("/")
        return float(num) / max(1.0, float(den))
    except Exception:          # noqa: BLE001
        return 0.0


def probe_basic_meta(path: str) -> Dict[str, Any]:
    """Container metadata with the reference's fallbacks (api.py:58-89): every field is zero / None when
    ffprobe cannot be used, and the analyzers then fall back to the frame source's own properties.

    Stream selection as the reference's loop does it: video streams are taken in order until one reports a
    non-zero width (that one wins; leading zero-width streams are overwritten), the first audio stream that
    is seen while no audio codec is known gives ``acodec``.  A ``bit_rate`` that is not a number raises, as it
    does in the reference (the request then fails with the service's 500 body)."""
    report = _ffprobe_json(path)
    meta: Dict[str, Any] = {"width": 0, "height": 0, "fps": 0.0, "duration": 0.0, "bit_rate": 0,
                            "vcodec": None, "acodec": None, "format_name": None}
    for stream in report.get("streams") or ():
        kind = stream.get("codec_type")
        if kind == "video" and not meta["width"]:
            meta["width"] = int(float(stream.get("width") or 0))
            meta["height"] = int(float(stream.get("height") or 0))
            meta["fps"] = _frame_rate(stream.get("r_frame_rate"))
            meta["vcodec"] = stream.get("codec_name")
        elif kind == "audio" and not meta["acodec"]:
            meta["acodec"] = stream.get("codec_name")
    container = report.get("format")
    if container:
        meta["bit_rate"] = int(float(container.get("bit_rate") or 0))
        meta["format_name"] = container.get("format_name")
        try:
            meta["duration"] = float(container.get("duration") or 0.0)
        except Exception:      # noqa: BLE001
            meta["duration"] = 0.0
    return meta


def _tlen(meta: dict) -> int:
    return int(max(1, round(meta.get("duration") or 0.0)))


def audio_unavailable(path: str, meta: dict) -> dict:
    """What the reference's audio analyzer returns when it cannot run (audio.py:112-118): a
    neutral timeline and an error flag (bench.py uses it: its synthetic clips have no sound track)."""
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


def _reference_forensic() -> Optional[Callable[[str], dict]]:
    """``meta.forensic_summary`` of a reference checkout behind this package on sys.path (api.py:18,164), if any."""
    try:
        from app.analyzers import meta as meta_an       # resolved through the extended package path
    except Exception:          # noqa: BLE001 -- no reference checkout: the key is simply absent
        return None
    return getattr(meta_an, "forensic_summary", None)


def analyze_path(path: str, meta: Optional[dict] = None, *, audio_analyzer: Optional[Callable[[str, dict], dict]] = None,
                 video_analyzer: Optional[Callable[[str, dict], dict]] = None, source_url: Optional[str] = None,
                 resolved_url: Optional[str] = None, debug: bool = False,
                 forensic: Optional[Callable[[str], dict]] = None) -> Dict[str, Any]:
    """The body of POST /analyze for one file (api.py:142-170), synchronous.

    ``forensic``: the exiftool / C2PA summary is control-plane I/O outside this build; pass the reference's
    ``meta.forensic_summary`` (or leave None: it is picked up automatically when a reference checkout sits behind
    this package on sys.path).  As in api.py:163-169 a falsy summary adds no key, and an exception adds
    ``forensic_error`` only in debug mode."""
    meta = dict(meta) if meta is not None else probe_basic_meta(path)
    for k in META_KEYS:
        meta.setdefault(k, None if k in ("vcodec", "acodec", "format_name") else 0)
    hints = hx.compute_hints(meta, path)
    audio, a_hint = safe_call("audio", audio_analyzer or audio_an.analyze, path, meta, debug)
    video, v_hint = safe_call("video", video_analyzer or video_an.analyze, path, meta, debug)
    hints.update(a_hint)
    hints.update(v_hint)
    fused = fusion_an.fuse(audio, video, hints)
    out = {
        "ok": True,
        "meta": {**meta, "source_url": source_url, "resolved_url": resolved_url},
        "hints": hints,
        "video": video,
        "audio": audio,
        "result": fused["result"],
        "timeline_binned": fused["timeline_binned"],
        "peaks": fused["peaks"],
    }
    summarize = forensic if forensic is not None else _reference_forensic()
    if summarize is not None:
        try:
            report = summarize(path)
            if report:
                out["forensic"] = report
        except Exception:      # noqa: BLE001 -- api.py:167
            if debug:
                out["forensic_error"] = traceback.format_exc()
    return out
