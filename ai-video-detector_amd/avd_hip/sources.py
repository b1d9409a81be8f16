"""Runtime-probed frame sources for ``app.analyzers.video.analyze(path, meta)``.

Container demux/decode (reference app/analyzers/video.py:11,28,32 -> cv2.VideoCapture /
libavcodec) is outside the parity boundary of this build: it is
third-party host plumbing, and none of cv2 / PyAV / ffmpeg exists in the build image.  A
source only has to yield what ``cap.retrieve()`` yields -- BGR uint8[H,W,3] -- for the frames
whose index is a multiple of ``step`` (every frame is still *grabbed*, i.e. decoded), plus the
capture properties the reference falls back to when ffprobe metadata is missing
(video.py:14-17).  Probed in this order:

  1. ``*.npy``  raw decoded stack uint8[T,H,W,3] (memory-mapped) -- used by tests / tools;
  2. ``*.y4m``  YUV4MPEG2, 8-bit 4:2:0 (what ``ffmpeg -i in.mp4 out.y4m`` writes, i.e. a decoder's pictures BEFORE the colour
     conversion): memory-mapped, handed over as NV12 surfaces to ``avd_analyze_frames_nv12`` -- the YUV->BGR step of
     ``cap.retrieve()`` happens in the HIP kernel, no BGR frame ever exists;
  3. ``cv2.VideoCapture`` if OpenCV is importable (identical decode to the reference);
  4. ``ffmpeg``/``ffprobe`` CLIs if on PATH (rawvideo bgr24 pipe).
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
    surface: str = "bgr"        # what sampled() yields: "bgr" = uint8[H,W,3]; "nv12" = (y uint8[H,W], uv uint8[H/2,W] interleaved U,V)
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


class Y4mSource(FrameSource):
    """YUV4MPEG2 file with 8-bit 4:2:0 pictures (C420, C420jpeg, C420mpeg2, C420paleo -- chroma siting does not enter
    libswscale's unscaled yuv420p -> bgr24 conversion, which takes the nearest chroma sample).  Frames are read through a
    memory map; ``sampled`` yields NV12 surfaces (the U and V planes interleaved on the host: 0.5 byte per pixel)."""
    surface = "nv12"

    def __init__(self, path: str):
        with open(path, "rb") as f:
            header = f.readline(4096)
        if not header.startswith(b"YUV4MPEG2 ") or not header.endswith(b"\n"):
            raise ValueError("not a YUV4MPEG2 stream")
        self._data_off = len(header)
        w = h = 0
        num, den, chroma = 0, 1, "420"
        for tok in header.split()[1:]:
            tag, val = tok[:1], tok[1:].decode("ascii", "replace")
            if tag == b"W":
                w = int(val)
            elif tag == b"H":
                h = int(val)
            elif tag == b"F":
                a, _, b = val.partition(":")
                num, den = int(a), int(b or 1)
            elif tag == b"C":
                chroma = val
        if w <= 0 or h <= 0 or (w | h) & 1:
            raise ValueError("bad or odd picture size")
        if not chroma.startswith("420") or "p1" in chroma:          # 420p10 / p12 / p16: more than 8 bits
            raise ValueError("only 8-bit 4:2:0 is supported")
        self.width, self.height = w, h
        self.fps = num / den if den else 0.0
        self._luma, self._chroma = w * h, (w // 2) * (h // 2)
        self._frame_bytes = self._luma + 2 * self._chroma
        self._map = np.memmap(path, dtype=np.uint8, mode="r")
        # every picture is preceded by "FRAME" [parameters] "\n"; the marker length is constant in files written by ffmpeg,
        # but it is read per frame anyway
        self._offsets = []
        pos = self._data_off
        size = self._map.shape[0]
        while pos + 6 <= size:
            end = pos
            limit = min(size, pos + 256)
            while end < limit and self._map[end] != 0x0A:
                end += 1
            if end >= limit or bytes(self._map[pos:pos + 5]) != b"FRAME":
                break
            if end + 1 + self._frame_bytes > size:
                break
            self._offsets.append(end + 1)
            pos = end + 1 + self._frame_bytes
        self.frame_count = len(self._offsets)

    def sampled(self, step):
        h, w = self.height, self.width
        for i in range(0, self.frame_count, step):
            o = self._offsets[i]
            y = np.asarray(self._map[o:o + self._luma]).reshape(h, w)
            u = self._map[o + self._luma:o + self._luma + self._chroma].reshape(h // 2, w // 2)
            v = self._map[o + self._luma + self._chroma:o + self._frame_bytes].reshape(h // 2, w // 2)
            uv = np.empty((h // 2, w), np.uint8)
            uv[:, 0::2] = u
            uv[:, 1::2] = v
            yield y, uv

    def close(self):
        self._map = None


def write_y4m(path: str, y: np.ndarray, uv: np.ndarray, fps=(30, 1)) -> None:
    """NV12 surfaces (y uint8[N,H,W], uv uint8[N,H/2,W] interleaved) -> a YUV4MPEG2 file (tests, tools)."""
    n, h, w = y.shape
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F%d:%d Ip A1:1 C420jpeg\n" % (w, h, fps[0], fps[1]))
        for i in range(n):
            f.write(b"FRAME\n")
            f.write(np.ascontiguousarray(y[i]).tobytes())
            f.write(np.ascontiguousarray(uv[i][:, 0::2]).tobytes())
            f.write(np.ascontiguousarray(uv[i][:, 1::2]).tobytes())


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
    if str(path).endswith(".y4m"):
        try:
            return Y4mSource(path)
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
