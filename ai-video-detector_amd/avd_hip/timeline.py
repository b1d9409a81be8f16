"""Scalar tail of the video analyzer: per-frame records -> the reference's result dict.

Everything here is O(N) float64 work on at most a few hundred numbers; it stays on the
host in numpy so that the reductions are the very same numpy calls the reference makes
(reference app/analyzers/video.py:54-83).  Inputs are the avd_frame_record fields that
the HIP kernels produce from pixels.
"""
from __future__ import annotations

import numpy as np


def sample_step(fps, per_second: float = 2) -> int:
    """Frame sampling stride, ~2 analysed frames per second (video.py:19).
    Python's round() is half-to-even: 25 fps -> 12, 29.97 -> 15, 60 -> 30.
    ``per_second`` is a build extension (the reference hard-codes 2): BASELINE.json configs[3] asks for
    dense 8 fps sampling, i.e. step = round(fps / 8); the default reproduces the reference."""
    return max(1, int(round((fps or 30) / per_second)))


def timeline_length(duration) -> int:
    """video.py:73"""
    return int(max(1, round(duration)))


def texture_variance(lap_sum, lap_sumsq, npix) -> np.ndarray:
    """Population variance of the (integer-valued) Laplacian image from its exact moments:
    var = (n*Sxx - Sx^2) / n^2, evaluated in exact integer arithmetic and rounded once.
    Agrees with ``cv2.Laplacian(gray, cv2.CV_64F).var()`` (video.py:52) to ~1e-15 relative."""
    n = int(npix)
    nn = n * n
    return np.array([(n * q - s * s) / nn for s, q in zip(lap_sum.tolist(), lap_sumsq.tolist())], np.float64)


def suspicion(tex: np.ndarray, mot: np.ndarray) -> np.ndarray:
    """ai_susp = clip(1 - tex/(tex+1000) * (1+mot), 0, 1) per sampled frame (video.py:56)."""
    return np.clip(1.0 - (tex / (tex + 1000.0)) * (1.0 + mot), 0.0, 1.0)


def records_to_result(rec: np.ndarray, npix: int, w, h, fps, duration) -> dict:
    """Assemble ``{"timeline", "summary", "timeline_ai"}`` exactly as video.py:54-83 does.

    rec: structured array (avd_hip.RECORD_DTYPE), one entry per sampled frame in order;
    rec[0] has no flow (flow_mean/var are ignored there, ham == -1).
    """
    total = int(len(rec))
    if total:
        tex = texture_variance(rec["lap_sum"], rec["lap_sumsq"], npix)
        flow_means = rec["flow_mean"][1:].astype(np.float64)     # float(np.float32) is exact
        flow_vars = rec["flow_var"][1:].astype(np.float64)
        mot = np.concatenate(([0.0], flow_means))                # first frame: no motion term yet
        timeline = suspicion(tex, mot).tolist()                  # Python floats, as float(v) per element gives
        dup = int(np.count_nonzero(rec["ham"][1:] == 0))
    else:
        tex = flow_means = flow_vars = np.empty(0, np.float64)
        timeline, dup = [], 0

    summary = {
        "dup_density": float(dup / max(1, total - 1)),
        "scene_change_rate": float(np.mean(flow_vars > 0.5)) if flow_vars.size else 0.0,
        "flow_mean": float(np.mean(flow_means)) if flow_means.size else 0.0,
        "flow_var": float(np.var(flow_means)) if flow_means.size else 0.0,
        "texture_var": float(np.var(tex)) if tex.size else 0.0,
        "w": int(w), "h": int(h), "fps": float(fps),
    }

    tlen = timeline_length(duration)
    if len(timeline) >= tlen:
        timeline = timeline[:tlen]            # entries are per sampled frame (~0.5 s): quirk kept
    elif timeline:
        timeline = timeline + [timeline[-1]] * (tlen - len(timeline))
    else:
        timeline = [0.5] * tlen
    # the reference returns ONE list object under both keys (fuse() later extends it in place)
    return {"timeline": timeline, "summary": summary, "timeline_ai": timeline}
