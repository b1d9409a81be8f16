"""ctypes binding of libavd_hip.so (C-ABI declared in include/avd.h).

The library is built in-tree by ``make -C ai-video-detector_amd/csrc`` (or
``__graft_entry__.build()``).  There is no CPU fallback: if the shared object is
missing, or no HIP device is usable, loading / context creation raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(_PKG, "csrc")
SO_PATH = os.path.join(_PKG, "lib", "libavd_hip.so")

AVD_MEM_HOST, AVD_MEM_DEVICE = 0, 1
SMALL, HASH = 320, 32

EXPORTS = (
    "avd_abi_version", "avd_create", "avd_destroy", "avd_last_error",
    "avd_preprocess_bgr", "avd_farneback_pairs", "avd_analyze_frames",
    "avd_analyze_frames_async", "avd_synchronize", "avd_analyze_batch", "avd_analyze_batch_async",
    "avd_wait_stream", "avd_release_workspace",
    "avd_preprocess_nv12", "avd_analyze_frames_nv12", "avd_analyze_frames_nv12_async",
    "avd_vit_set_weights", "avd_vit_patch_embed", "avd_audio_features", "avd_layernorm", "avd_softmax",
    "avd_cnn_param_counts", "avd_cnn_set_weights", "avd_cnn_forward", "avd_cnn_conv",
    "avd_comm_unique_id", "avd_comm_init", "avd_allgather_records", "avd_allgather_last_records",
    "avd_timer_start", "avd_timer_stop", "avd_set_option", "avd_get_option",
    "avd_set_profiling", "avd_stage_ms", "avd_kernel_ms", "avd_debug_fetch",
)

# numpy view of struct avd_audio_window (48 bytes)
AUDIO_WINDOW_DTYPE = np.dtype([("sumsq", "<f8"), ("sum_log", "<f8"), ("sum_mag", "<f8"), ("sum_fmag", "<f8"),
                               ("zero_cross", "<i4"), ("length", "<i4"), ("rolloff_index", "<i4"), ("nbins", "<i4")])

# numpy view of struct avd_frame_record (32 bytes)
RECORD_DTYPE = np.dtype([("lap_sum", "<i8"), ("lap_sumsq", "<i8"), ("flow_mean", "<f4"),
                         ("flow_var", "<f4"), ("ham", "<i4"), ("reserved", "<i4")])


def f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns (uint16), round to nearest even (finite inputs)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


class AvdClip(C.Structure):
    """struct avd_clip (include/avd.h): one clip of a batch, BGR (uv = NULL) or NV12."""
    _fields_ = [("data", C.c_void_p), ("uv", C.c_void_p), ("mem", C.c_int), ("n", C.c_int), ("h", C.c_int), ("w", C.c_int),
                ("row_stride", C.c_int64), ("frame_stride", C.c_int64), ("uv_row_stride", C.c_int64), ("uv_frame_stride", C.c_int64)]


class AvdError(RuntimeError):
    """Non-zero status from the C-ABI (the analyzer may raise; reference api.py:134-140
    turns any exception into the neutral 0.5 timeline)."""


def build(force: bool = False) -> str:
    """Compile the HIP extension for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "avd.h"))
    stale = (not os.path.exists(SO_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", CSRC] + (["-B"] if force else []), check=True, stdout=sys.stderr)   # keep stdout clean (bench.py prints one JSON line)
    return SO_PATH


_lib = None


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).
    Two HIP runtimes in one process cannot both own the GPU, and device pointers of torch
    tensors are only meaningful to the runtime that allocated them -- so when torch is
    installed, its runtime is loaded first and libavd_hip.so binds to it by SONAME."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    return None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(f"{SO_PATH} is missing: build it with `make -C {CSRC}` "
                          "(there is no CPU fallback for the HIP path)")
    _preload_hip_runtime()
    L = C.CDLL(SO_PATH)
    vp, u8p, f32p, i64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    L.avd_abi_version.restype = C.c_int
    if L.avd_abi_version() != 3:          # before any other symbol is bound: an older library fails HERE, with a message
        raise ImportError("libavd_hip.so ABI version mismatch (this package binds version 3): rebuild it with `make -C csrc`")
    L.avd_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.avd_destroy.argtypes = [vp]
    L.avd_destroy.restype = None
    L.avd_last_error.argtypes = [vp]
    L.avd_last_error.restype = C.c_char_p
    L.avd_preprocess_bgr.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                     u8p, u8p, i64p, i64p]
    L.avd_farneback_pairs.argtypes = [vp, u8p, C.c_int, C.c_int, f32p, f32p, f32p]
    L.avd_analyze_frames.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, vp]
    L.avd_analyze_frames_async.argtypes = L.avd_analyze_frames.argtypes
    nv12 = [vp, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64]
    L.avd_preprocess_nv12.argtypes = nv12 + [u8p, u8p, i64p, i64p]
    L.avd_analyze_frames_nv12.argtypes = nv12 + [vp]
    L.avd_analyze_frames_nv12_async.argtypes = nv12 + [vp]
    L.avd_vit_set_weights.argtypes = [vp, vp, vp]
    L.avd_vit_patch_embed.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, vp, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_float)]
    L.avd_cnn_param_counts.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.avd_cnn_set_weights.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
    L.avd_cnn_forward.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, vp, C.c_int, C.POINTER(C.c_float)]
    L.avd_cnn_conv.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.avd_audio_features.argtypes = [vp, vp, C.c_int, C.c_int64, C.c_int, vp, C.c_int]
    L.avd_layernorm.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int, vp, vp, C.c_float, vp, C.c_int, C.POINTER(C.c_float)]
    L.avd_softmax.argtypes = [vp, vp, C.c_int, C.c_int64, C.c_int, vp, C.c_int, C.POINTER(C.c_float)]
    L.avd_comm_unique_id.argtypes = [vp]
    L.avd_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.avd_allgather_records.argtypes = [vp, vp, C.c_int, vp]
    L.avd_allgather_last_records.argtypes = [vp, C.c_int, vp]
    L.avd_synchronize.argtypes = [vp]
    L.avd_analyze_batch.argtypes = [vp, vp, C.c_int, vp]
    L.avd_analyze_batch_async.argtypes = [vp, vp, C.c_int, vp]
    L.avd_wait_stream.argtypes = [vp, vp]
    L.avd_release_workspace.argtypes = [vp]
    L.avd_timer_start.argtypes = [vp]
    L.avd_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.avd_set_profiling.argtypes = [vp, C.c_int]
    L.avd_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.avd_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.avd_stage_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.avd_kernel_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.avd_debug_fetch.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    L.avd_debug_fetch.restype = C.c_int64
    for name in EXPORTS:
        if name not in ("avd_destroy", "avd_last_error", "avd_debug_fetch"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _is_torch_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class Context:
    """One avd_ctx: one device, one HIP stream, one workspace.  Not re-entrant -- use one
    Context per thread (ctypes releases the GIL for the duration of each call)."""

    def __init__(self, device: int = 0):
        self._L = load()
        h = C.c_void_p()
        rc = self._L.avd_create(int(device), C.byref(h))
        if rc != 0 or not h:
            raise AvdError(f"avd_create(device={device}) failed with status {rc}: no usable HIP device "
                           "(the HIP path has no CPU fallback)")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.avd_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            msg = self._L.avd_last_error(self._h)
            raise AvdError(f"avd status {rc}: {msg.decode() if msg else ''}")

    # -- buffers: numpy (host) or torch-ROCm tensors (device) ----------------------------
    def _after_torch_stream(self, t):
        """A device tensor may still be being written by torch's current stream (a fresh result, the copy kernel of
        ``.contiguous()``); the context launches on its own non-blocking stream, so order the two with an event
        (no host synchronisation).  The caller keeps the tensor alive until the work has been drained."""
        import torch
        if t.device.index is not None and t.device.index != self.device:
            raise ValueError(f"tensor lives on cuda:{t.device.index}, context on device {self.device}")
        self._check(self._L.avd_wait_stream(self._h, C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)))

    def _frames_ptr(self, frames):
        """-> (ptr, mem, n, h, w, row_stride, frame_stride, keepalive)"""
        if _is_torch_tensor(frames):
            t = frames
            if t.dim() != 4 or t.shape[-1] != 3 or str(t.dtype) != "torch.uint8":
                raise ValueError("frames must be uint8[N,H,W,3] (BGR)")
            n, h, w, _ = t.shape
            if t.stride(-1) != 1 or t.stride(-2) != 3 or t.stride(1) < 3 * w or (n > 1 and t.stride(0) < t.stride(1) * h):
                t = t.contiguous()
            mem = AVD_MEM_DEVICE if t.is_cuda else AVD_MEM_HOST
            if t.is_cuda:
                self._after_torch_stream(t)
            fs = t.stride(0) if n > 1 else h * t.stride(1)
            return t.data_ptr(), mem, n, h, w, t.stride(1), fs, t
        a = np.asarray(frames)
        if a.ndim != 4 or a.shape[-1] != 3 or a.dtype != np.uint8:
            raise ValueError("frames must be uint8[N,H,W,3] (BGR)")
        if a.strides[-1] != 1 or a.strides[-2] != 3 or a.strides[1] < a.shape[2] * 3 or (
                a.shape[0] > 1 and a.strides[0] < a.strides[1] * a.shape[1]):
            a = np.ascontiguousarray(a)
        n, h, w, _ = a.shape
        fs = a.strides[0] if n > 1 else h * a.strides[1]
        return a.ctypes.data, AVD_MEM_HOST, n, h, w, a.strides[1], fs, a

    def preprocess_bgr(self, frames):
        ptr, mem, n, h, w, rs, fs, keep = self._frames_ptr(frames)
        small = np.empty((n, SMALL, SMALL), np.uint8)
        hsh = np.empty((n, HASH * HASH), np.uint8)
        s = np.empty(n, np.int64)
        q = np.empty(n, np.int64)
        self._check(self._L.avd_preprocess_bgr(self._h, ptr, mem, n, h, w, rs, fs, small.ctypes.data,
                                               hsh.ctypes.data, s.ctypes.data, q.ctypes.data))
        return small, hsh, s, q

    def farneback_pairs(self, small, want_flow: bool = False):
        if _is_torch_tensor(small):
            t = small.contiguous()
            if t.is_cuda:
                self._after_torch_stream(t)
            ptr, mem, n, keep = t.data_ptr(), (AVD_MEM_DEVICE if t.is_cuda else AVD_MEM_HOST), t.shape[0], t
        else:
            a = np.ascontiguousarray(small, dtype=np.uint8)
            ptr, mem, n, keep = a.ctypes.data, AVD_MEM_HOST, a.shape[0], a
        m = max(n - 1, 0)
        fm = np.zeros(m, np.float32)
        fv = np.zeros(m, np.float32)
        flow = np.empty((m, SMALL, SMALL, 2), np.float32) if want_flow else None
        self._check(self._L.avd_farneback_pairs(self._h, ptr, mem, n, fm.ctypes.data, fv.ctypes.data,
                                                flow.ctypes.data if want_flow else None))
        return (fm, fv, flow) if want_flow else (fm, fv)

    def analyze_frames(self, frames) -> np.ndarray:
        """-> structured array (RECORD_DTYPE) with one record per frame."""
        ptr, mem, n, h, w, rs, fs, keep = self._frames_ptr(frames)
        rec = np.zeros(n, RECORD_DTYPE)
        self._check(self._L.avd_analyze_frames(self._h, ptr, mem, n, h, w, rs, fs, rec.ctypes.data))
        return rec

    # -- NV12 (decoder surfaces): y uint8[N,H,W], uv uint8[N,H/2,W] with U,V interleaved ----------------------
    def _nv12_ptrs(self, y, uv):
        """-> (yptr, uvptr, mem, n, h, w, y_row, uv_row, y_frame, uv_frame, keepalive)"""
        if _is_torch_tensor(y) != _is_torch_tensor(uv):
            raise ValueError("both planes must be numpy arrays or both torch tensors")
        if _is_torch_tensor(y):
            if y.dim() != 3 or uv.dim() != 3 or str(y.dtype) != "torch.uint8" or str(uv.dtype) != "torch.uint8" or y.is_cuda != uv.is_cuda:
                raise ValueError("planes must be uint8[N,H,W] and uint8[N,H/2,W] on the same device")
            ok = lambda t: t.stride(2) == 1 and t.stride(1) >= t.shape[2] and (t.shape[0] == 1 or t.stride(0) >= t.stride(1) * t.shape[1])
            y, uv = (y if ok(y) else y.contiguous()), (uv if ok(uv) else uv.contiguous())
            if y.is_cuda:
                self._after_torch_stream(y)
            n, h, w = y.shape
            strides = (y.stride(1), uv.stride(1), y.stride(0) if n > 1 else h * y.stride(1), uv.stride(0) if n > 1 else (h // 2) * uv.stride(1))
            ptrs, mem = (y.data_ptr(), uv.data_ptr()), (AVD_MEM_DEVICE if y.is_cuda else AVD_MEM_HOST)
        else:
            y, uv = np.asarray(y), np.asarray(uv)
            if y.ndim != 3 or uv.ndim != 3 or y.dtype != np.uint8 or uv.dtype != np.uint8:
                raise ValueError("planes must be uint8[N,H,W] and uint8[N,H/2,W]")
            ok = lambda a: a.strides[2] == 1 and a.strides[1] >= a.shape[2] and (a.shape[0] == 1 or a.strides[0] >= a.strides[1] * a.shape[1])
            y, uv = (y if ok(y) else np.ascontiguousarray(y)), (uv if ok(uv) else np.ascontiguousarray(uv))
            n, h, w = y.shape
            strides = (y.strides[1], uv.strides[1], y.strides[0] if n > 1 else h * y.strides[1], uv.strides[0] if n > 1 else (h // 2) * uv.strides[1])
            ptrs, mem = (y.ctypes.data, uv.ctypes.data), AVD_MEM_HOST
        if tuple(uv.shape) != (n, h // 2, w):
            raise ValueError(f"chroma plane must be uint8[{n},{h // 2},{w}] (interleaved U,V), got {tuple(uv.shape)}")
        return ptrs + (mem, n, h, w) + strides + ((y, uv),)

    def preprocess_nv12(self, y, uv):
        yp, cp, mem, n, h, w, yr, cr, yf, cf, keep = self._nv12_ptrs(y, uv)
        small = np.empty((n, SMALL, SMALL), np.uint8)
        hsh = np.empty((n, HASH * HASH), np.uint8)
        s = np.empty(n, np.int64)
        q = np.empty(n, np.int64)
        self._check(self._L.avd_preprocess_nv12(self._h, yp, cp, mem, n, h, w, yr, cr, yf, cf, small.ctypes.data,
                                                hsh.ctypes.data, s.ctypes.data, q.ctypes.data))
        return small, hsh, s, q

    def analyze_frames_nv12(self, y, uv) -> np.ndarray:
        yp, cp, mem, n, h, w, yr, cr, yf, cf, keep = self._nv12_ptrs(y, uv)
        rec = np.zeros(n, RECORD_DTYPE)
        self._check(self._L.avd_analyze_frames_nv12(self._h, yp, cp, mem, n, h, w, yr, cr, yf, cf, rec.ctypes.data))
        return rec

    def analyze_frames_nv12_async(self, y, uv, rec: np.ndarray):
        yp, cp, mem, n, h, w, yr, cr, yf, cf, keep = self._nv12_ptrs(y, uv)
        assert rec.dtype == RECORD_DTYPE and rec.size >= n and rec.flags.c_contiguous
        self._check(self._L.avd_analyze_frames_nv12_async(self._h, yp, cp, mem, n, h, w, yr, cr, yf, cf, rec.ctypes.data))
        return keep

    # -- LayerNorm / softmax (extensions) -------------------------------------------------------------------------
    def layernorm(self, x, gamma, beta, eps: float = 1e-5, timing_reps: int = 0, out=None):
        """x: [rows, cols] float32 numpy array, or a contiguous float32 / bfloat16 torch-ROCm tensor (then ``out`` of the same
        kind receives the result in HBM, default: a new tensor).  -> (y, kernel_ms or None)"""
        g = np.ascontiguousarray(gamma, np.float32)
        b = np.ascontiguousarray(beta, np.float32)
        ms = C.c_float(0.0)
        if _is_torch_tensor(x):
            import torch
            if not (x.is_cuda and x.is_contiguous() and x.dim() == 2):
                raise ValueError("x must be a contiguous 2-D cuda tensor")
            bf = {"torch.float32": 0, "torch.bfloat16": 1}[str(x.dtype)]
            y = out if out is not None else torch.empty_like(x)
            self._after_torch_stream(x)
            self._check(self._L.avd_layernorm(self._h, x.data_ptr(), AVD_MEM_DEVICE, bf, x.shape[0], x.shape[1], g.ctypes.data, b.ctypes.data,
                                              eps, y.data_ptr(), timing_reps, C.byref(ms)))
            return y, (ms.value if timing_reps > 0 else None)
        a = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(a)
        self._check(self._L.avd_layernorm(self._h, a.ctypes.data, AVD_MEM_HOST, 0, a.shape[0], a.shape[1], g.ctypes.data, b.ctypes.data, eps,
                                          y.ctypes.data, timing_reps, C.byref(ms)))
        return y, (ms.value if timing_reps > 0 else None)

    def softmax(self, x, timing_reps: int = 0):
        """x: [rows, cols] float32 (numpy or contiguous cuda tensor) -> (probabilities of the same kind, kernel_ms or None)"""
        ms = C.c_float(0.0)
        if _is_torch_tensor(x):
            import torch
            if not (x.is_cuda and x.is_contiguous() and x.dim() == 2 and str(x.dtype) == "torch.float32"):
                raise ValueError("x must be a contiguous 2-D float32 cuda tensor")
            y = torch.empty_like(x)
            self._after_torch_stream(x)
            self._check(self._L.avd_softmax(self._h, x.data_ptr(), AVD_MEM_DEVICE, x.shape[0], x.shape[1], y.data_ptr(), timing_reps, C.byref(ms)))
            return y, (ms.value if timing_reps > 0 else None)
        a = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(a)
        self._check(self._L.avd_softmax(self._h, a.ctypes.data, AVD_MEM_HOST, a.shape[0], a.shape[1], y.ctypes.data, timing_reps, C.byref(ms)))
        return y, (ms.value if timing_reps > 0 else None)

    # -- ViT-B/16 patch embedding (extension, never part of ai_score) ------------------------------------------
    def vit_set_weights(self, weight: np.ndarray, bias=None):
        """weight float32[768, 768] ([out][c*256 + py*16 + px]) -> rounded to bf16 (nearest even); bias float32[768]."""
        wbits = f32_to_bf16_bits(np.ascontiguousarray(weight, np.float32).reshape(768, 768))
        b = None if bias is None else np.ascontiguousarray(bias, np.float32).reshape(768)
        self._check(self._L.avd_vit_set_weights(self._h, wbits.ctypes.data, None if b is None else b.ctypes.data))

    def vit_patch_embed(self, frames, timing_reps: int = 0, out=None, bf16: bool = False):
        """-> (tokens [N,196,768], gemm_ms or None).  Tokens are float32, or with ``bf16=True`` bf16 (returned to the host
        as float32 after widening).  ``out``: optional contiguous torch-ROCm tensor [N,196,768] (float32, or bfloat16 with
        ``bf16=True``) that receives the tokens in HBM (nothing is copied to the host then)."""
        ptr, mem, n, h, w, rs, fs, keep = self._frames_ptr(frames)
        ms = C.c_float(0.0)
        if out is not None:
            want = "torch.bfloat16" if bf16 else "torch.float32"
            if not (_is_torch_tensor(out) and out.is_cuda and out.is_contiguous() and tuple(out.shape) == (n, 196, 768)
                    and str(out.dtype) == want):
                raise ValueError(f"out must be a contiguous {want} cuda tensor [N,196,768]")
            self._after_torch_stream(out)
            tptr, tmem, tokens = out.data_ptr(), AVD_MEM_DEVICE, out
        else:
            tokens = np.empty((n, 196, 768), np.uint16 if bf16 else np.float32)
            tptr, tmem = tokens.ctypes.data, AVD_MEM_HOST
        self._check(self._L.avd_vit_patch_embed(self._h, ptr, mem, n, h, w, rs, fs, tptr, tmem, int(bool(bf16)), int(timing_reps), C.byref(ms)))
        if out is None and bf16:
            tokens = bf16_bits_to_f32(tokens)
        return tokens, (float(ms.value) if timing_reps > 0 else None)

    # -- CNN extension (ResNet-50-style forward on the matrix cores; never part of ai_score) -------------------------
    @staticmethod
    def cnn_param_counts():
        """-> (weights, biases): elements of the flat parameter arrays of avd_cnn_set_weights."""
        nw, nb = C.c_size_t(0), C.c_size_t(0)
        if load().avd_cnn_param_counts(C.byref(nw), C.byref(nb)) != 0:
            raise RuntimeError("avd_cnn_param_counts failed")
        return int(nw.value), int(nb.value)

    def cnn_set_weights(self, weights: np.ndarray, biases: np.ndarray):
        """weights: float32 flat (rounded to bf16, nearest even) or uint16 bf16 bits, in the order documented in avd.h."""
        w = np.ascontiguousarray(weights).reshape(-1)
        wbits = w if w.dtype == np.uint16 else f32_to_bf16_bits(w.astype(np.float32, copy=False))
        b = np.ascontiguousarray(biases, np.float32).reshape(-1)
        self._check(self._L.avd_cnn_set_weights(self._h, wbits.ctypes.data, wbits.size, b.ctypes.data, b.size))

    def cnn_forward(self, frames, timing_reps: int = 0):
        """-> (logits float32 [N,1000], forward_ms or None)."""
        ptr, mem, n, h, w, rs, fs, keep = self._frames_ptr(frames)
        logits = np.empty((n, 1000), np.float32)
        ms = C.c_float(0.0)
        self._check(self._L.avd_cnn_forward(self._h, ptr, mem, n, h, w, rs, fs, logits.ctypes.data, int(timing_reps), C.byref(ms)))
        return logits, (float(ms.value) if timing_reps > 0 else None)

    def cnn_conv(self, x: np.ndarray, w: np.ndarray, bias: np.ndarray, stride: int = 1, relu: bool = True, residual=None) -> np.ndarray:
        """One convolution layer (test entry).  x float32 NHWC [n,h,w,cin] and w float32 [cout,k,k,cin] are rounded to bf16;
        -> float32 NHWC (the widened bf16 output)."""
        n, hin, win, cin = x.shape
        cout, k, k2, cin2 = w.shape
        assert k == k2 and cin == cin2
        pad = k // 2
        hout, wout = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
        xb = f32_to_bf16_bits(np.ascontiguousarray(x, np.float32))
        wb = f32_to_bf16_bits(np.ascontiguousarray(w, np.float32))
        b = np.ascontiguousarray(bias, np.float32)
        rb = None if residual is None else f32_to_bf16_bits(np.ascontiguousarray(residual, np.float32).reshape(n, hout, wout, cout))
        y = np.empty((n, hout, wout, cout), np.uint16)
        self._check(self._L.avd_cnn_conv(self._h, xb.ctypes.data, n, hin, win, cin, wb.ctypes.data, b.ctypes.data, cout, k, stride,
                                         int(bool(relu)), None if rb is None else rb.ctypes.data, y.ctypes.data))
        return bf16_bits_to_f32(y)

    # -- audio analyzer (reference app/analyzers/audio.py:40-61 for all windows at once) -----------------------
    def audio_features(self, wav, win: int) -> np.ndarray:
        """wav: mono float32 samples (numpy or torch-ROCm tensor) -> structured array (AUDIO_WINDOW_DTYPE) per window."""
        if _is_torch_tensor(wav):
            t = wav.contiguous()
            if str(t.dtype) != "torch.float32" or t.dim() != 1:
                raise ValueError("wav must be float32[n]")
            if t.is_cuda:
                self._after_torch_stream(t)
            ptr, mem, n, keep = t.data_ptr(), (AVD_MEM_DEVICE if t.is_cuda else AVD_MEM_HOST), int(t.numel()), t
        else:
            a = np.ascontiguousarray(wav, dtype=np.float32)
            if a.ndim != 1:
                raise ValueError("wav must be float32[n]")
            ptr, mem, n, keep = a.ctypes.data, AVD_MEM_HOST, int(a.size), a
        nwin = (n + win - 1) // win if n else 0
        out = np.zeros(nwin, AUDIO_WINDOW_DTYPE)
        self._check(self._L.avd_audio_features(self._h, ptr, mem, n, int(win), out.ctypes.data, nwin))
        return out

    # -- record exchange across ranks (RCCL, bound at run time) --------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = load().avd_comm_unique_id(buf)
        if rc != 0:
            raise AvdError(f"avd_comm_unique_id failed with status {rc} (is librccl.so available?)")
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == 128
        self._check(self._L.avd_comm_init(self._h, int(rank), int(world), C.create_string_buffer(unique_id, 128)))
        self._comm_world = int(world)

    def allgather_records(self, local: np.ndarray) -> np.ndarray:
        local = np.ascontiguousarray(local)
        assert local.dtype == RECORD_DTYPE
        out = np.zeros(len(local) * getattr(self, "_comm_world", 1), RECORD_DTYPE)
        self._check(self._L.avd_allgather_records(self._h, local.ctypes.data, len(local), out.ctypes.data))
        return out

    def allgather_last_records(self, count: int) -> np.ndarray:
        """Gather the first ``count`` records of this context's last analysis call from every rank, straight from HBM
        (enqueued behind the analysis on the context's stream; drains an outstanding asynchronous call)."""
        out = np.zeros(count * getattr(self, "_comm_world", 1), RECORD_DTYPE)
        self._check(self._L.avd_allgather_last_records(self._h, count, out.ctypes.data))
        return out

    def analyze_frames_async(self, frames, rec: np.ndarray):
        ptr, mem, n, h, w, rs, fs, keep = self._frames_ptr(frames)
        assert rec.dtype == RECORD_DTYPE and rec.size >= n and rec.flags.c_contiguous
        self._check(self._L.avd_analyze_frames_async(self._h, ptr, mem, n, h, w, rs, fs, rec.ctypes.data))
        return keep          # the buffer the kernels read: the caller holds it until synchronize()

    def synchronize(self):
        self._check(self._L.avd_synchronize(self._h))

    # -- a batch of clips in one call (include/avd.h: avd_analyze_batch) -----------------------------------------
    def _clip_array(self, clips):
        """clips: sequence of BGR frame stacks uint8[N,H,W,3] or NV12 plane pairs (y, uv); numpy or torch, any mix of
        geometries -> (AvdClip array, frame counts, keepalive)"""
        arr = (AvdClip * len(clips))()
        keep, counts = [], []
        for i, c in enumerate(clips):
            if isinstance(c, tuple):
                yp, cp, mem, n, h, w, yr, cr, yf, cf, k = self._nv12_ptrs(*c)
                arr[i] = AvdClip(yp, cp, mem, n, h, w, yr, yf, cr, cf)
            else:
                ptr, mem, n, h, w, rs, fs, k = self._frames_ptr(c)
                arr[i] = AvdClip(ptr, None, mem, n, h, w, rs, fs, 0, 0)
            keep.append(k)
            counts.append(n)
        return arr, counts, keep

    def analyze_batch(self, clips):
        """-> list of record arrays, one per clip (identical to analyze_frames / analyze_frames_nv12 per clip)."""
        arr, counts, keep = self._clip_array(clips)
        rec = np.zeros(sum(counts), RECORD_DTYPE)
        self._check(self._L.avd_analyze_batch(self._h, arr, len(clips), rec.ctypes.data))
        return list(np.split(rec, np.cumsum(counts)[:-1])) if counts else []

    def analyze_batch_async(self, clips, rec: np.ndarray):
        """Enqueue only; rec (RECORD_DTYPE, sum of the clips' frames) is filled by synchronize().  Returns what the caller
        must keep alive until then."""
        arr, counts, keep = self._clip_array(clips)
        assert rec.dtype == RECORD_DTYPE and rec.size >= sum(counts) and rec.flags.c_contiguous
        self._check(self._L.avd_analyze_batch_async(self._h, arr, len(clips), rec.ctypes.data))
        return keep, counts

    def wait_stream(self, stream_handle: int = 0):
        """Order this context's stream behind everything already enqueued on another HIP stream (raw handle)."""
        self._check(self._L.avd_wait_stream(self._h, C.c_void_p(stream_handle)))

    def release_workspace(self):
        self._check(self._L.avd_release_workspace(self._h))

    def timer_start(self):
        self._check(self._L.avd_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self._L.avd_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    def set_option(self, name: str, value: int):
        """Tuning / test switches, e.g. ``set_option("fb_fused", 0)`` selects the two-kernel Farneback path."""
        self._check(self._L.avd_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        """Current value of an option (environment defaults included), or a read-only counter such as ``rerun_pairs``."""
        v = C.c_int()
        self._check(self._L.avd_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def set_profiling(self, on: bool):
        self._check(self._L.avd_set_profiling(self._h, int(bool(on))))

    def stage_ms(self):
        out = []
        for i in range(6):
            ms = C.c_float()
            self._check(self._L.avd_stage_ms(self._h, i, C.byref(ms)))
            out.append(float(ms.value))
        return out

    KERNEL_IDS = ("preprocess", "hash", "pyramid", "polyexp", "level40", "flow_up80", "level80", "flow_up160", "level160",
                  "flow_up320", "level320", "rerun", "stats", "records", "other")      # enum avd_kernel_id

    def kernel_ms(self) -> dict:
        """Per-kernel device time (ms) of the last drained call with profiling on (avd_kernel_ms)."""
        out = {}
        for i, name in enumerate(self.KERNEL_IDS):
            ms = C.c_float()
            self._check(self._L.avd_kernel_ms(self._h, i, C.byref(ms)))
            out[name] = float(ms.value)
        return out

    def debug_fetch(self, name: str, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype)
        got = self._L.avd_debug_fetch(self._h, name.encode(), out.ctypes.data, out.nbytes)
        if got < 0:
            self._check(int(got))
        if got != out.nbytes:
            raise AvdError(f"debug buffer {name}: expected {out.nbytes} bytes, got {got}")
        return out
