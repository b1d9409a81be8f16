"""FrameAnalyzer: decoded BGR frames -> HIP kernels -> per-frame records -> result dict.

This is the frames-level boundary of the hot path: everything the
reference does per sampled frame between ``cap.retrieve()`` and ``timeline_ai.append``
(reference app/analyzers/video.py:36-57), batched over all sampled frames of a clip.
"""
from __future__ import annotations

import threading
from typing import Iterable, Optional

import numpy as np

from . import _lib
from .timeline import records_to_result

_tls = threading.local()


def thread_context(device: int = 0) -> "_lib.Context":
    """One avd_ctx per (thread, device): reference api.py:133 runs the analyzer on worker
    threads (asyncio.to_thread), so contexts are never shared between threads."""
    cache = getattr(_tls, "ctx", None)
    if cache is None:
        cache = _tls.ctx = {}
    if device not in cache:
        cache[device] = _lib.Context(device)
    return cache[device]


class FrameAnalyzer:
    def __init__(self, device: int = 0, chunk: int = 64, ctx: Optional["_lib.Context"] = None):
        self.ctx = ctx or thread_context(device)
        self.chunk = max(2, int(chunk))

    # -- whole stack resident (numpy host array or torch-ROCm tensor) ----------------------
    def records(self, frames) -> np.ndarray:
        return self.ctx.analyze_frames(frames)

    def analyze(self, frames, meta: dict) -> dict:
        """frames: uint8[N,H,W,3] BGR, the SAMPLED frames of one clip in order."""
        n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        rec = self.records(frames) if n else np.zeros(0, _lib.RECORD_DTYPE)
        return records_to_result(rec, h * w, meta.get("width") or w, meta.get("height") or h,
                                 meta.get("fps") or 0.0, meta.get("duration") or 0.0)

    # -- streaming: bounded host memory, one-frame halo between chunks ----------------------
    def records_stream(self, frames: Iterable[np.ndarray]) -> np.ndarray:
        out = []
        buf = []
        carry = None                   # last frame of the previous chunk (flow / hash predecessor)
        for fr in frames:
            buf.append(fr)
            if len(buf) >= self.chunk:
                out.append(self._flush(buf, carry))
                carry, buf = buf[-1], []
        if buf:
            out.append(self._flush(buf, carry))
        return np.concatenate(out) if out else np.zeros(0, _lib.RECORD_DTYPE)

    def _flush(self, buf, carry):
        stack = np.stack(([carry] if carry is not None else []) + buf)
        rec = self.ctx.analyze_frames(stack)
        return rec[1:] if carry is not None else rec


def analyze_frames(frames, meta: Optional[dict] = None, device: int = 0) -> dict:
    """Convenience wrapper: the frames-level analogue of reference ``video.analyze``."""
    return FrameAnalyzer(device).analyze(frames, meta or {})
