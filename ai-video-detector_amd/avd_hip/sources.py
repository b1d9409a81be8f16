"""Runtime-probed frame sources for ``app.analyzers.video.analyze(path, meta)``.

Container demux/decode (reference app/analyzers/video.py:11,28,32 -> cv2.VideoCapture /
libavcodec) is outside the parity boundary of this build: it is
third-party host plumbing, and none of cv2 / PyAV / ffmpeg exists in the build image.  A
source only has to yield what ``cap.retrieve()`` yields -- BGR uint8[H,W,3] -- for the frames
whose index is a multiple of ``step`` (every frame is still *grabbed*, i.e. decoded), plus the
capture properties the reference falls back to when ffprobe metadata is missing
(video.py:14-17).  Probed in this order:

  1. ``*.npy``  raw decoded stack uint8[T,H,W,3] (memory-mapped) -- used by tests / tools;
  2. ``cv2.VideoCapture`` if OpenCV is importable (identical decode to the reference);
  3. ``ffmpeg``/``ffprobe`` CLIs if on PATH (rawvideo bgr24 pipe).
If nothing can open the file the analyzer returns the reference's "capture not opened"
result (video.py:12-13).
"""
from __future__ import annotations

import json
import shutil
import subprocess
from typing import Iterator, Optional

import numpy as np


class FrameSource:
    fps: float = 0.0
    width: int = 0
    height: int = 0
    frame_count: int = 0

    def sampled(self, step: int) -> Iterator[np.ndarray]:
        raise NotImplementedError

    def close(self) -> None:
        pass


class NpySource(FrameSource):
    def __init__(self, path: str):
        arr = np.load(path, mmap_mode="r")
        if arr.ndim != 4 or arr.shape[-1] != 3 or arr.dtype != np.uint8:
            raise ValueError("expected uint8[T,H,W,3]")
        self._arr = arr
        self.frame_count, self.height, self.width = int(arr.shape[0]), int(arr.shape[1]), int(arr.shape[2])

    def sampled(self, step):
        for i in range(0, self.frame_count, step):
            yield np.ascontiguousarray(self._arr[i])


class Cv2Source(FrameSource):
    def __init__(self, path: str, cv2):
        self._cv2 = cv2
        self._cap = cv2.VideoCapture(path)
        if not self._cap.isOpened():
            raise ValueError("capture not opened")
        self.fps = float(self._cap.get(cv2.CAP_PROP_FPS) or 0.0)
        self.width = int(self._cap.get(cv2.CAP_PROP_FRAME_WIDTH) or 0)
        self.height = int(self._cap.get(cv2.CAP_PROP_FRAME_HEIGHT) or 0)
        self.frame_count = int(self._cap.get(cv2.CAP_PROP_FRAME_COUNT) or 0)

    def sampled(self, step):
        index = 0
        while self._cap.grab():
            if index % step == 0:
                ok, frame = self._cap.retrieve()
                if not ok:
                    break
                yield frame
            index += 1

    def close(self):
        self._cap.release()


class FfmpegSource(FrameSource):
    def __init__(self, path: str):
        info = json.loads(subprocess.check_output(
            ["ffprobe", "-v", "error", "-select_streams", "v:0", "-show_entries",
             "stream=width,height,r_frame_rate,nb_frames", "-of", "json", path], text=True, timeout=30))
        st = info["streams"][0]
        self.width, self.height = int(st["width"]), int(st["height"])
        num, den = (st.get("r_frame_rate") or "0/1").split("/")
        self.fps = float(num) / max(1.0, float(den))
        try:
            self.frame_count = int(st.get("nb_frames") or 0)
        except ValueError:
            self.frame_count = 0
        self._path = path
        self._proc = None

    def sampled(self, step):
        self._proc = subprocess.Popen(["ffmpeg", "-v", "error", "-i", self._path, "-f", "rawvideo",
                                       "-pix_fmt", "bgr24", "-"], stdout=subprocess.PIPE)
        nbytes = self.width * self.height * 3
        index = 0
        while True:
            buf = self._proc.stdout.read(nbytes)
            if len(buf) < nbytes:
                break
            if index % step == 0:
                yield np.frombuffer(buf, np.uint8).reshape(self.height, self.width, 3)
            index += 1

    def close(self):
        if self._proc is not None:
            self._proc.kill()
            self._proc.wait()


def open_source(path: str) -> Optional[FrameSource]:
    """First source that can open ``path``; None if none can (== capture not opened)."""
    if str(path).endswith(".npy"):
        try:
            return NpySource(path)
        except (OSError, ValueError):
            return None
    try:
        import cv2  # type: ignore
    except ImportError:
        cv2 = None
    if cv2 is not None:
        try:
            return Cv2Source(path, cv2)
        except ValueError:
            return None
    if shutil.which("ffmpeg") and shutil.which("ffprobe"):
        try:
            return FfmpegSource(path)
        except (subprocess.SubprocessError, KeyError, IndexError, ValueError, OSError):
            return None
    return None
