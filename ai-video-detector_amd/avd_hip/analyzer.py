"""FrameAnalyzer: decoded BGR frames -> HIP kernels -> per-frame records -> result dict.

This is the frames-level boundary of the hot path: everything the
reference does per sampled frame between ``cap.retrieve()`` and ``timeline_ai.append``
(reference app/analyzers/video.py:36-57), batched over all sampled frames of a clip.
"""
from __future__ import annotations

import logging

import collections
import contextlib
import os
import threading
from typing import Iterable, Iterator, Optional, Tuple

import numpy as np

from . import _lib
from .timeline import records_to_result

class ContextPool:
    """Bounded pool of avd contexts for a threaded service.

    Reference api.py:133 runs the analyzer on ``asyncio.to_thread`` workers (up to 32 threads in the default
    executor) and a context is not re-entrant, so every request borrows one for its duration.  A context holds
    about 1.5 GB of HBM scratch after a 120-frame 1080p clip, so the pool is bounded: at most ``max_contexts``
    exist per device (further requests wait for one -- three or four clips in flight already saturate the GPU,
    DESIGN.md 4.5), and only ``keep_warm`` idle ones keep their workspace; the others give it back
    (``avd_release_workspace``) and re-reserve it on their next use.
    """

    def __init__(self, max_contexts: Optional[int] = None, keep_warm: Optional[int] = None):
        self.max_contexts = max(1, int(max_contexts if max_contexts is not None else os.getenv("AVD_MAX_CONTEXTS", "4")))
        self.keep_warm = max(0, int(keep_warm if keep_warm is not None else os.getenv("AVD_WARM_CONTEXTS", "2")))
        self._cv = threading.Condition()
        self._warm = {}            # device -> idle contexts that still hold their workspace, most recently used last
        self._cold = {}            # device -> idle contexts whose workspace was given back
        self._count = {}           # device -> contexts created

    @contextlib.contextmanager
    def borrow(self, device: int = 0):
        ctx = self._take(device)
        try:
            yield ctx
        except BaseException:
            # the request failed on this context: before it goes back to the MOST recently used end of the warm list -- the next request's
            # pick -- its stream is probed.  A device error is sticky on a HIP stream: such a context is closed and its slot given back (the
            # pool creates a fresh one); an ordinary error (bad argument, unreadable file) leaves a healthy context, which is reused.
            if self._healthy(ctx):
                self._give(device, ctx)
            else:
                self._drop(device, ctx, "the borrower's call failed and the context's stream reports an error")
            raise
        else:
            self._give(device, ctx)

    @staticmethod
    def _healthy(ctx) -> bool:
        try:
            ctx.synchronize()
            return True
        except Exception:
            return False

    def _drop(self, device, ctx, why):
        logging.getLogger("avd_hip").warning("context dropped from the pool: %s", why)
        try:
            ctx.close()
        except Exception:
            pass
        with self._cv:
            self._count[device] = max(0, self._count.get(device, 0) - 1)
            self._cv.notify()

    def _take(self, device):
        with self._cv:
            while True:
                warm, cold = self._warm.setdefault(device, []), self._cold.setdefault(device, [])
                if warm:
                    return warm.pop()                      # the most recently used one
                if cold:
                    return cold.pop()
                if self._count.get(device, 0) < self.max_contexts:
                    self._count[device] = self._count.get(device, 0) + 1
                    break
                self._cv.wait()
        try:
            return _lib.Context(device)
        except BaseException:
            with self._cv:
                self._count[device] -= 1
                self._cv.notify()
            raise

    def _give(self, device, ctx):
        trim = []
        with self._cv:
            warm = self._warm.setdefault(device, [])
            warm.append(ctx)
            while len(warm) > self.keep_warm:
                trim.append(warm.pop(0))                   # least recently used
            if warm:
                self._cv.notify()
        for c in trim:                                     # outside the lock: this synchronises the context's stream
            try:
                c.release_workspace()
            except Exception as exc:                       # must not mask the request's own result (or exception) ...
                # ... but a context whose stream reports an error (sticky after a fault) must not be handed to the next request: it is
                # logged, closed and its slot given back, so the pool creates a fresh one instead
                self._drop(device, c, "release_workspace failed: %r" % (exc,))
                continue
            with self._cv:
                self._cold.setdefault(device, []).append(c)
                self._cv.notify()

    def stats(self, device: int = 0):
        with self._cv:
            return {"created": self._count.get(device, 0), "warm": len(self._warm.get(device, [])),
                    "cold": len(self._cold.get(device, []))}

    def close(self):
        with self._cv:
            ctxs = [c for d in (self._warm, self._cold) for lst in d.values() for c in lst]
            self._warm.clear()
            self._cold.clear()
            self._count.clear()
        for c in ctxs:
            c.close()


_pool = ContextPool()


def default_pool() -> ContextPool:
    return _pool


class FrameAnalyzer:
    """Frames -> records on ONE context.  Pass ``ctx`` (tests, batch jobs) or borrow one from a pool
    (``with default_pool().borrow(device) as ctx: FrameAnalyzer(ctx=ctx)...``), which is what the drop-in
    ``app.analyzers.video.analyze`` does per request."""

    def __init__(self, device: int = 0, chunk: int = 64, ctx: Optional["_lib.Context"] = None):
        self._own = ctx is None
        self.ctx = ctx or _lib.Context(device)
        self.chunk = max(2, int(chunk))

    def close(self):
        if self._own and self.ctx is not None:
            self.ctx.close()
        self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- whole stack resident (numpy host array or torch-ROCm tensor) ----------------------
    def records(self, frames) -> np.ndarray:
        return self.ctx.analyze_frames(frames)

    def analyze(self, frames, meta: dict) -> dict:
        """frames: uint8[N,H,W,3] BGR, the SAMPLED frames of one clip in order."""
        n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        rec = self.records(frames) if n else np.zeros(0, _lib.RECORD_DTYPE)
        return records_to_result(rec, h * w, meta.get("width") or w, meta.get("height") or h,
                                 meta.get("fps") or 0.0, meta.get("duration") or 0.0)

    # -- several clips in one call (avd_analyze_batch): any mix of geometries; short clips fill the GPU together -------
    def records_many(self, clips) -> list:
        """clips: sequence of uint8[N,H,W,3] BGR stacks (or (y, uv) NV12 plane pairs) -> list of record arrays."""
        return self.ctx.analyze_batch(list(clips))

    def analyze_many(self, clips, metas) -> list:
        """The batched form of ``analyze``: one result dict per clip, identical to analysing them one by one."""
        clips = list(clips)
        out = []
        for frames, meta, rec in zip(clips, metas, self.records_many(clips)):
            y = frames[0] if isinstance(frames, tuple) else frames
            h, w = int(y.shape[1]), int(y.shape[2])
            out.append(records_to_result(rec, h * w, meta.get("width") or w, meta.get("height") or h,
                                         meta.get("fps") or 0.0, meta.get("duration") or 0.0))
        return out

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

    # -- the same for decoder surfaces: an iterable of (y uint8[H,W], uv uint8[H/2,W]) pairs ----------------
    def records_stream_nv12(self, surfaces) -> np.ndarray:
        out = []
        buf = []
        carry = None
        for sf in surfaces:
            buf.append(sf)
            if len(buf) >= self.chunk:
                out.append(self._flush_nv12(buf, carry))
                carry, buf = buf[-1], []
        if buf:
            out.append(self._flush_nv12(buf, carry))
        return np.concatenate(out) if out else np.zeros(0, _lib.RECORD_DTYPE)

    def _flush_nv12(self, buf, carry):
        items = ([carry] if carry is not None else []) + buf
        rec = self.ctx.analyze_frames_nv12(np.stack([y for y, _ in items]), np.stack([uv for _, uv in items]))
        return rec[1:] if carry is not None else rec


class ClipsInFlight:
    """Throughput mode for a service that analyses many clips on one GPU: up to ``depth`` clips are in
    flight, each on its own avd context (stream + workspace).  40 % of a clip's GPU time is launches that
    cannot fill the chip (the coarse Farneback levels, ~60 launch gaps, the host tail); other clips'
    bandwidth-bound kernels fill those holes: +18 % frames/s with 2, +24 % with 3 clips in flight (MI355X,
    1080p).  Results come back in submission order and are bit-identical to one-at-a-time analysis.

        runner = ClipsInFlight(device=0, depth=3)
        for tag, rec in runner.run((name, frames) for name, frames in clips):
            ...
    """

    def __init__(self, device: int = 0, depth: int = 3):
        self.ctxs = [_lib.Context(device) for _ in range(max(1, int(depth)))]
        self._pending = collections.deque()          # (slot, tag, records buffer, frames kept alive)
        self._free = collections.deque(range(len(self.ctxs)))

    @property
    def full(self) -> bool:
        return not self._free

    def submit(self, frames, tag=None) -> None:
        """Enqueue one clip (uint8[N,H,W,3] BGR, host numpy or torch-ROCm tensor) and return at once."""
        if self.full:
            raise RuntimeError("ClipsInFlight.submit: all contexts busy, drain() first")
        slot = self._free.popleft()
        rec = np.zeros(int(frames.shape[0]), _lib.RECORD_DTYPE)
        # `keep` is the buffer the kernels actually read (the input itself, or a contiguous copy of it): it must
        # stay allocated until the clip has been drained, or torch's caching allocator may hand it out again
        keep = self.ctxs[slot].analyze_frames_async(frames, rec)
        self._pending.append((slot, tag, rec, (frames, keep)))

    def submit_batch(self, clips, tag=None) -> None:
        """Enqueue SEVERAL clips as one call of one context (avd_analyze_batch_async): their Farneback pairs run as one
        launch sequence.  drain() returns (tag, [records of clip 0, records of clip 1, ...])."""
        if self.full:
            raise RuntimeError("ClipsInFlight.submit_batch: all contexts busy, drain() first")
        clips = list(clips)
        slot = self._free.popleft()
        total = sum(int((c[0] if isinstance(c, tuple) else c).shape[0]) for c in clips)
        rec = np.zeros(total, _lib.RECORD_DTYPE)
        keep, counts = self.ctxs[slot].analyze_batch_async(clips, rec)
        self._pending.append((slot, tag, (rec, counts), (clips, keep)))

    def drain(self) -> Tuple[object, np.ndarray]:
        """Wait for the OLDEST clip (or batch) in flight -> (tag, records)."""
        slot, tag, rec, _frames = self._pending.popleft()
        self.ctxs[slot].synchronize()
        self._free.append(slot)
        if isinstance(rec, tuple):
            rec, counts = rec
            return tag, (list(np.split(rec, np.cumsum(counts)[:-1])) if counts else [])
        return tag, rec

    def run(self, clips: Iterable[Tuple[object, object]]) -> Iterator[Tuple[object, np.ndarray]]:
        for tag, frames in clips:
            if self.full:
                yield self.drain()
            self.submit(frames, tag)
        while self._pending:
            yield self.drain()


def analyze_frames(frames, meta: Optional[dict] = None, device: int = 0) -> dict:
    """Convenience wrapper: the frames-level analogue of reference ``video.analyze``."""
    with default_pool().borrow(device) as ctx:
        return FrameAnalyzer(ctx=ctx).analyze(frames, meta or {})
