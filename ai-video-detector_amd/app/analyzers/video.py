"""MI355X drop-in for reference ``app/analyzers/video.py`` -- same name, signature, result.

``analyze(path, meta)`` keeps the reference's contract (video.py:10-83):
  * returns ``{"timeline": [...], "summary": {...}, "timeline_ai": <same list object>}``;
  * an unopenable file is NOT an error: ``{"timeline": [], "summary": {}, "timeline_ai": []}``;
  * ``meta`` values win over capture properties (fps, width, height, duration);
  * may raise (e.g. :class:`avd_hip.AvdError` when the HIP library reports a failure) --
    reference api.py:134-140 turns any exception into the neutral 0.5 timeline.
All pixel work (gray, aHash, 320x320 resize, Farneback, Laplacian moments) runs in the HIP
kernels behind the C-ABI (include/avd.h); there is no CPU fallback.
"""
from __future__ import annotations

import os

from avd_hip import analyzer as _analyzer
from avd_hip import sources as _sources
from avd_hip.timeline import records_to_result, sample_step

_DEVICE = int(os.getenv("AVD_DEVICE", "0"))
_CHUNK = int(os.getenv("AVD_CHUNK_FRAMES", "64"))      # sampled frames per HIP call (host memory bound)
_SAMPLES_PER_SECOND = float(os.getenv("AVD_SAMPLES_PER_SECOND", "2"))   # extension knob; 2 = the reference (video.py:19)


def analyze(path: str, meta: dict):
    src = _sources.open_source(path)
    if src is None:
        return {"timeline": [], "summary": {}, "timeline_ai": []}
    try:
        fps = meta.get("fps") or src.fps or 0.0
        w = meta.get("width") or int(src.width or 0)
        h = meta.get("height") or int(src.height or 0)
        duration = meta.get("duration") or (src.frame_count / fps if fps > 0 else 0.0)
        step = sample_step(fps, _SAMPLES_PER_SECOND)

        seen = {}

        def frames():
            for fr in src.sampled(step):
                seen.setdefault("npix", int(fr.shape[0]) * int(fr.shape[1]))
                yield fr

        rec = _analyzer.FrameAnalyzer(device=_DEVICE, chunk=_CHUNK).records_stream(frames())
    finally:
        src.close()
    return records_to_result(rec, seen.get("npix", 0), w, h, fps, duration)
