"""CPU ORACLE bindings -- test infrastructure, NOT the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  It wraps oracle/avd_oracle.c (a plain-C restatement of the OpenCV 4.10 /
numpy arithmetic used by reference app/analyzers/video.py) and restates the
control flow of reference ``video.analyze`` (video.py:10-83) over an in-memory
stack of already-decoded BGR frames.

PARITY UNPINNED at the OpenCV boundary -- see oracle/README.md.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libavd_oracle.so")
_lib = None

SMALL = 320          # video.py:43
HASH = 32            # video.py:36


def build(force: bool = False) -> str:
    """Compile oracle/avd_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "avd_oracle.c")
    hdr = os.path.join(_HERE, "avd_oracle.h")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True, stdout=sys.stderr)
    return _SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, f32p, f64p, i64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float),
                                 C.POINTER(C.c_double), C.POINTER(C.c_int64))
        L.avdo_bgr2gray.argtypes = [u8p, C.c_int, C.c_int, C.c_int64, u8p]
        L.avdo_resize_area_u8.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        L.avdo_resize_linear_u8.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        L.avdo_hash_bits.argtypes = [u8p, C.c_int, u8p]
        L.avdo_laplacian_f64.argtypes = [u8p, C.c_int, C.c_int, f64p]
        L.avdo_laplacian_sums.argtypes = [u8p, C.c_int, C.c_int, i64p, i64p]
        L.avdo_farneback.argtypes = [u8p, u8p, C.c_int, C.c_int, f32p, C.c_double, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_double]
        L.avdo_flow_stats.argtypes = [f32p, C.c_int64, f32p, f32p, f32p]
        L.avdo_np_sum_f32.argtypes = [f32p, C.c_int64]
        L.avdo_np_sum_f32.restype = C.c_float
        L.avdo_gaussian_kernel_f32.argtypes = [C.c_int, C.c_double, f32p]
        L.avdo_gaussian_blur_f32.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_double, f32p]
        L.avdo_resize_linear_f32.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int]
        L.avdo_poly_prepare.argtypes = [C.c_int, C.c_double, f32p, f32p, f32p, f64p]
        L.avdo_poly_exp.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_double, f32p]
        L.avdo_update_matrices.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.avdo_update_flow_blur.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.avdo_preprocess_bgr.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                          u8p, u8p, i64p, i64p]
        L.avdo_farneback_pairs.argtypes = [u8p, C.c_int, f32p, f32p]
        L.avdo_nv12_to_bgr24.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int64, C.c_int64, u8p, C.c_int64]
        L.avdo_nv12_to_bgr24.restype = C.c_int
        L.avdo_yuv2rgb_consts.argtypes = [i64p]
        L.avdo_yuv2rgb_consts.restype = None
        L.avdo_ops_reset.restype = None
        L.avdo_ops_get.argtypes = [C.POINTER(C.c_uint64)]
        L.avdo_ops_get.restype = None
        L.avdo_set_model.argtypes = [C.c_int]
        L.avdo_set_model.restype = None
        L.avdo_get_model.restype = C.c_int
        for name in ("avdo_resize_area_u8", "avdo_resize_linear_u8", "avdo_farneback",
                     "avdo_gaussian_kernel_f32", "avdo_resize_linear_f32",
                     "avdo_preprocess_bgr", "avdo_farneback_pairs"):
            getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


# model switches of avd_oracle.h (sensitivity analysis of the open questions, oracle/README.md)
MODEL_GAUSS_MULADD, MODEL_JITTER_PYRAMID, MODEL_JITTER_FLOW, MODEL_THREE_SCALES, MODEL_RESIZE_LERP = 1, 2, 4, 8, 16


class model:
    """``with oracle.model(flags): ...`` runs the oracle with the given modelled choices flipped."""

    def __init__(self, flags: int):
        self.flags = int(flags)

    def __enter__(self):
        self.saved = lib().avdo_get_model()
        lib().avdo_set_model(self.flags)
        return self

    def __exit__(self, *a):
        lib().avdo_set_model(self.saved)


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# ---- stage wrappers --------------------------------------------------------------
def bgr2gray(bgr: np.ndarray) -> np.ndarray:
    bgr = _c(bgr, np.uint8)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().avdo_bgr2gray(_p(bgr, C.c_uint8), h, w, w * 3, _p(out, C.c_uint8))
    return out


def resize_area(gray: np.ndarray, dh: int, dw: int) -> np.ndarray:
    gray = _c(gray, np.uint8)
    out = np.empty((dh, dw), np.uint8)
    rc = lib().avdo_resize_area_u8(_p(gray, C.c_uint8), gray.shape[0], gray.shape[1],
                                   _p(out, C.c_uint8), dh, dw)
    if rc:
        raise ValueError("INTER_AREA upscaling is outside the restated path")
    return out


def resize_linear(gray: np.ndarray, dh: int, dw: int) -> np.ndarray:
    gray = _c(gray, np.uint8)
    out = np.empty((dh, dw), np.uint8)
    rc = lib().avdo_resize_linear_u8(_p(gray, C.c_uint8), gray.shape[0], gray.shape[1],
                                     _p(out, C.c_uint8), dh, dw)
    if rc:
        raise ValueError("resize failed")
    return out


def hash_bits(area: np.ndarray) -> np.ndarray:
    area = _c(area, np.uint8).reshape(-1)
    out = np.empty(area.size, np.uint8)
    lib().avdo_hash_bits(_p(area, C.c_uint8), area.size, _p(out, C.c_uint8))
    return out


def average_hash(bgr: np.ndarray, size: int = HASH) -> np.ndarray:
    """reference video.py:4-8"""
    return hash_bits(resize_area(bgr2gray(bgr), size, size))


def laplacian_f64(gray: np.ndarray) -> np.ndarray:
    gray = _c(gray, np.uint8)
    out = np.empty(gray.shape, np.float64)
    lib().avdo_laplacian_f64(_p(gray, C.c_uint8), gray.shape[0], gray.shape[1], _p(out, C.c_double))
    return out


def laplacian_sums(gray: np.ndarray):
    gray = _c(gray, np.uint8)
    s, q = C.c_int64(), C.c_int64()
    lib().avdo_laplacian_sums(_p(gray, C.c_uint8), gray.shape[0], gray.shape[1], C.byref(s), C.byref(q))
    return int(s.value), int(q.value)


def farneback(prev: np.ndarray, nxt: np.ndarray, pyr_scale=0.5, levels=3, winsize=15,
              iterations=3, poly_n=5, poly_sigma=1.2) -> np.ndarray:
    prev, nxt = _c(prev, np.uint8), _c(nxt, np.uint8)
    h, w = prev.shape
    flow = np.empty((h, w, 2), np.float32)
    rc = lib().avdo_farneback(_p(prev, C.c_uint8), _p(nxt, C.c_uint8), h, w, _p(flow, C.c_float),
                              pyr_scale, levels, winsize, iterations, poly_n, poly_sigma)
    if rc:
        raise ValueError("farneback failed")
    return flow


def flow_stats(flow: np.ndarray):
    flow = _c(flow, np.float32)
    n = flow.size // 2
    m, v = C.c_float(), C.c_float()
    lib().avdo_flow_stats(_p(flow, C.c_float), n, C.byref(m), C.byref(v), None)
    return np.float32(m.value), np.float32(v.value)


def count_farneback_ops(prev: np.ndarray, nxt: np.ndarray) -> dict:
    """Arithmetic operations the oracle executes for ONE frame pair: calcOpticalFlowFarneback (pyramid blur + resize,
    polynomial expansion of both frames at four scales, matrices, three blur iterations per scale) + the flow statistics
    (video.py:45-48).  -> {"f32": n, "f64": n}, counted in avd_oracle.c where the operations happen."""
    L = lib()
    L.avdo_ops_reset()
    fl = farneback(prev, nxt)
    flow_stats(fl)
    out = (C.c_uint64 * 2)()
    L.avdo_ops_get(out)
    return {"f32": int(out[0]), "f64": int(out[1])}


def count_level_ops(R0: np.ndarray, R1: np.ndarray, flow: np.ndarray, iterations: int = 3) -> dict:
    """Operations of the blur iterations of ONE pyramid level of one pair (FarnebackUpdateMatrices + `iterations` x
    FarnebackUpdateFlow_Blur, as avdo_farneback sequences them): what one level kernel of the HIP path computes."""
    L = lib()
    L.avdo_ops_reset()
    fl = np.array(flow, np.float32, copy=True)
    M = update_matrices(R0, R1, fl)
    for i in range(iterations):
        fl, M = update_flow_blur(R0, R1, fl, M, 15, i < iterations - 1)
    out = (C.c_uint64 * 2)()
    L.avdo_ops_get(out)
    return {"f32": int(out[0]), "f64": int(out[1])}


def np_sum_f32(a: np.ndarray) -> np.float32:
    a = _c(a, np.float32).reshape(-1)
    return np.float32(lib().avdo_np_sum_f32(_p(a, C.c_float), a.size))


def gaussian_kernel(ksize: int, sigma: float) -> np.ndarray:
    k = np.empty(ksize, np.float32)
    lib().avdo_gaussian_kernel_f32(ksize, sigma, _p(k, C.c_float))
    return k


def gaussian_blur(img: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    img = _c(img, np.float32)
    out = np.empty_like(img)
    lib().avdo_gaussian_blur_f32(_p(img, C.c_float), img.shape[0], img.shape[1], ksize, sigma, _p(out, C.c_float))
    return out


def resize_linear_f32(img: np.ndarray, dh: int, dw: int) -> np.ndarray:
    img = _c(img, np.float32)
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.float32)
    rc = lib().avdo_resize_linear_f32(_p(img, C.c_float), img.shape[0], img.shape[1], cn, _p(out, C.c_float), dh, dw)
    if rc:
        raise ValueError("resize failed")
    return out


def poly_prepare(n: int = 5, sigma: float = 1.2):
    g = np.zeros(2 * n + 1, np.float32)
    xg = np.zeros_like(g)
    xxg = np.zeros_like(g)
    ig = np.zeros(4, np.float64)
    f32p = C.POINTER(C.c_float)
    off = n * 4
    lib().avdo_poly_prepare(n, sigma, C.cast(g.ctypes.data + off, f32p), C.cast(xg.ctypes.data + off, f32p),
                            C.cast(xxg.ctypes.data + off, f32p), _p(ig, C.c_double))
    return g, xg, xxg, ig


def poly_exp(img: np.ndarray, n: int = 5, sigma: float = 1.2) -> np.ndarray:
    img = _c(img, np.float32)
    out = np.empty(img.shape + (5,), np.float32)
    lib().avdo_poly_exp(_p(img, C.c_float), img.shape[0], img.shape[1], n, sigma, _p(out, C.c_float))
    return out


def update_matrices(R0, R1, flow) -> np.ndarray:
    R0, R1, flow = _c(R0, np.float32), _c(R1, np.float32), _c(flow, np.float32)
    h, w = flow.shape[:2]
    M = np.empty((h, w, 5), np.float32)
    lib().avdo_update_matrices(_p(R0, C.c_float), _p(R1, C.c_float), _p(flow, C.c_float), _p(M, C.c_float), h, w, 0, h)
    return M


def update_flow_blur(R0, R1, flow, M, block_size=15, update=True):
    R0, R1 = _c(R0, np.float32), _c(R1, np.float32)
    flow, M = _c(flow, np.float32).copy(), _c(M, np.float32).copy()
    h, w = flow.shape[:2]
    lib().avdo_update_flow_blur(_p(R0, C.c_float), _p(R1, C.c_float), _p(flow, C.c_float), _p(M, C.c_float),
                                h, w, block_size, int(update))
    return flow, M


# ---- frame-level twins of the product C-ABI (include/avd.h) -------------------------
def nv12_to_bgr(y: np.ndarray, uv: np.ndarray) -> np.ndarray:
    """libswscale's C yuv420 -> BGR24 (BT.601 limited range, nearest chroma) for NV12 planes:
    y uint8[..., H, W], uv uint8[..., H/2, W] (U, V interleaved) -> uint8[..., H, W, 3]."""
    y = np.ascontiguousarray(y, np.uint8)
    uv = np.ascontiguousarray(uv, np.uint8)
    h, w = y.shape[-2:]
    assert uv.shape[-2:] == (h // 2, w) and h % 2 == 0 and w % 2 == 0
    out = np.empty(y.shape + (3,), np.uint8)
    yy, cc, oo = y.reshape(-1, h, w), uv.reshape(-1, h // 2, w), out.reshape(-1, h, w, 3)
    for i in range(len(yy)):
        rc = lib().avdo_nv12_to_bgr24(_p(yy[i], C.c_uint8), _p(cc[i], C.c_uint8), h, w, w, w, _p(oo[i], C.c_uint8), 3 * w)
        if rc:
            raise ValueError("avdo_nv12_to_bgr24 failed")
    return out


def yuv2rgb_consts():
    out = np.zeros(6, np.int64)
    lib().avdo_yuv2rgb_consts(_p(out, C.c_int64))
    return {k: int(v) for k, v in zip(("cy", "crv", "cbu", "cgu", "cgv", "c0"), out)}


def preprocess_bgr(frames: np.ndarray):
    """frames uint8[N,H,W,3] -> (small u8[N,320,320], hash u8[N,1024], lap_sum i64[N], lap_sumsq i64[N])"""
    frames = _c(frames, np.uint8)
    n, h, w, _ = frames.shape
    small = np.empty((n, SMALL, SMALL), np.uint8)
    hsh = np.empty((n, HASH * HASH), np.uint8)
    s = np.empty(n, np.int64)
    q = np.empty(n, np.int64)
    rc = lib().avdo_preprocess_bgr(_p(frames, C.c_uint8), n, h, w, w * 3, h * w * 3,
                                   _p(small, C.c_uint8), _p(hsh, C.c_uint8), _p(s, C.c_int64), _p(q, C.c_int64))
    if rc:
        raise ValueError("preprocess failed (frame smaller than 32x32?)")
    return small, hsh, s, q


def farneback_pairs(small: np.ndarray):
    small = _c(small, np.uint8)
    n = small.shape[0]
    fm = np.empty(max(n - 1, 0), np.float32)
    fv = np.empty(max(n - 1, 0), np.float32)
    if n > 1:
        rc = lib().avdo_farneback_pairs(_p(small, C.c_uint8), n, _p(fm, C.c_float), _p(fv, C.c_float))
        if rc:
            raise ValueError("farneback failed")
    return fm, fv


def texture_var(lap_sum: int, lap_sumsq: int, npix: int) -> float:
    """float(cv2.Laplacian(gray, CV_64F).var()) from exact integer moments (video.py:52).
    var = (n*Sxx - Sx^2)/n^2, one correctly-rounded division of exact integers."""
    num = int(npix) * int(lap_sumsq) - int(lap_sum) * int(lap_sum)
    return num / (int(npix) * int(npix))


# ---- video.analyze restated over decoded frames (video.py:10-83) ---------------------
def sample_step(fps) -> int:
    """video.py:19"""
    return max(1, int(round((fps or 30) / 2)))


def analyze_sampled_frames(frames: np.ndarray, meta: dict, exact_numpy_var: bool = False) -> dict:
    """Restates video.py:20-83 given the stack of SAMPLED frames (what cap.retrieve()
    returned at index % step == 0).  ``meta`` plays the role of the ffprobe dict
    (video.py:14-17); capture-property fallbacks are the frame stack's own shape."""
    frames = np.asarray(frames)
    fps = meta.get("fps") or 0.0
    w = meta.get("width") or (int(frames.shape[2]) if frames.ndim == 4 else 0)
    h = meta.get("height") or (int(frames.shape[1]) if frames.ndim == 4 else 0)
    duration = meta.get("duration") or 0.0

    prev_hash = None
    dup = 0
    total = 0
    flow_means, flow_vars, textures, timeline_ai = [], [], [], []
    prev_small = None
    for frame in frames:
        total += 1
        gray = bgr2gray(frame)
        hsh = hash_bits(resize_area(gray, HASH, HASH))
        if prev_hash is not None:
            ham = int(np.sum(hsh ^ prev_hash))
            if ham == 0:
                dup += 1
        prev_hash = hsh

        small = resize_linear(gray, SMALL, SMALL)
        if prev_small is not None:
            flow = farneback(prev_small, small)
            m, v = flow_stats(flow)
            flow_means.append(float(m))
            flow_vars.append(float(v))
        prev_small = small

        if exact_numpy_var:
            textures.append(float(laplacian_f64(gray).var()))
        else:
            s, q = laplacian_sums(gray)
            textures.append(texture_var(s, q, gray.size))

        tex = textures[-1]
        mot = flow_means[-1] if flow_means else 0.0
        ai_susp = float(np.clip(1.0 - (tex / (tex + 1000.0)) * (1.0 + mot), 0.0, 1.0))
        timeline_ai.append(ai_susp)

    dup_density = float(dup / max(1, total - 1))
    sc_rate = float(np.mean(np.array(flow_vars) > 0.5)) if flow_vars else 0.0
    summary = {
        "dup_density": dup_density,
        "scene_change_rate": sc_rate,
        "flow_mean": float(np.mean(flow_means)) if flow_means else 0.0,
        "flow_var": float(np.var(flow_means)) if flow_means else 0.0,
        "texture_var": float(np.var(textures)) if textures else 0.0,
        "w": int(w), "h": int(h), "fps": float(fps),
    }
    tlen = int(max(1, round(duration)))
    if len(timeline_ai) < tlen:
        if timeline_ai:
            timeline_ai += [timeline_ai[-1]] * (tlen - len(timeline_ai))
        else:
            timeline_ai = [0.5] * tlen
    else:
        timeline_ai = timeline_ai[:tlen]
    return {"timeline": timeline_ai, "summary": summary, "timeline_ai": timeline_ai}
