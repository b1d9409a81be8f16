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

def _settings():
    """Environment knobs, read per call (a service may be reconfigured without a restart; tests set them per case):
    AVD_DEVICE            HIP device index (default 0)
    AVD_CHUNK_FRAMES      sampled frames per HIP call (bounds host memory; default 64)
    AVD_SAMPLES_PER_SECOND  extension knob, 2 = the reference's hard-coded sampling (video.py:19); 8 = the dense
                          sampling of BASELINE.json configs[3]"""
    return (int(os.getenv("AVD_DEVICE", "0")), int(os.getenv("AVD_CHUNK_FRAMES", "64")),
            float(os.getenv("AVD_SAMPLES_PER_SECOND", "2")))


def analyze(path: str, meta: dict):
    device, chunk, per_second = _settings()
    src = _sources.open_source(path)
    if src is None:
        return {"timeline": [], "summary": {}, "timeline_ai": []}
    try:
        fps = meta.get("fps") or src.fps or 0.0
        w = meta.get("width") or int(src.width or 0)
        h = meta.get("height") or int(src.height or 0)
        duration = meta.get("duration") or (src.frame_count / fps if fps > 0 else 0.0)
        step = sample_step(fps, per_second)

        seen = {}

        nv12 = getattr(src, "surface", "bgr") == "nv12"      # decoder pictures: the colour conversion happens on the GPU

        def frames():
            for fr in src.sampled(step):
                plane = fr[0] if nv12 else fr
                seen.setdefault("npix", int(plane.shape[0]) * int(plane.shape[1]))
                yield fr

        # one pooled context for the duration of the request (bounded pool: api.py:133 runs this on worker threads)
        with _analyzer.default_pool().borrow(device) as ctx:
            fa = _analyzer.FrameAnalyzer(chunk=chunk, ctx=ctx)
            rec = fa.records_stream_nv12(frames()) if nv12 else fa.records_stream(frames())
    finally:
        src.close()
    return records_to_result(rec, seen.get("npix", 0), w, h, fps, duration)
