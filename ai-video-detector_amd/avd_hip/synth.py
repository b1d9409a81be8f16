"""Seeded synthetic clips (SURVEY.md section 8d recipe) for parity tests and bench.py.

Frame k of a clip is a smooth random colour field translated by k*(dx,dy) pixels plus i.i.d.
integer noise in {-2..2}; every 10th sampled frame duplicates its predecessor (exercises the
aHash duplicate counter) and there is one hard scene cut in the middle (exercises the
flow-variance scene-change threshold).  Only SAMPLED frames are materialised.
"""
from __future__ import annotations

import numpy as np


def _smooth_field(rng, h, w, sigma=8.0):
    """Gaussian-filtered white noise scaled to 0..255, 3 slightly different channels."""
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.rfftfreq(w)[None, :]
    tf = np.exp(-2.0 * (np.pi * sigma) ** 2 * (fx * fx + fy * fy))
    out = np.empty((h, w, 3), np.float32)
    shared = rng.standard_normal((h, w)).astype(np.float32)
    for c in range(3):
        noise = shared + 0.35 * rng.standard_normal((h, w)).astype(np.float32)
        f = np.fft.irfft2(np.fft.rfft2(noise) * tf, s=(h, w))
        f = (f - f.min()) / max(float(f.max() - f.min()), 1e-9)
        out[..., c] = (f * 255.0).astype(np.float32)
    return out


def make_clip(n: int, h: int, w: int, seed: int = 0, dup_every: int = 10, scene_cut: bool = True) -> np.ndarray:
    """-> uint8[n, h, w, 3] (BGR), the sampled frames of one synthetic clip."""
    rng = np.random.default_rng(seed)
    margin = 3 * max(n, 1) + 4
    fields = [_smooth_field(rng, h + 2 * margin, w + 2 * margin)]
    if scene_cut:
        fields.append(_smooth_field(rng, h + 2 * margin, w + 2 * margin))
    dx, dy = rng.uniform(-3.0, 3.0, size=2)
    frames = np.empty((n, h, w, 3), np.uint8)
    for k in range(n):
        if dup_every and k > 0 and k % dup_every == 0:
            frames[k] = frames[k - 1]
            continue
        fld = fields[1] if (scene_cut and k >= n // 2) else fields[0]
        oy = margin + int(round(k * dy))
        ox = margin + int(round(k * dx))
        crop = fld[oy:oy + h, ox:ox + w]
        noise = rng.integers(-2, 3, size=(h, w, 3), dtype=np.int16)
        frames[k] = np.clip(np.rint(crop).astype(np.int16) + noise, 0, 255).astype(np.uint8)
    return frames


def random_frames(n: int, h: int, w: int, seed: int = 0) -> np.ndarray:
    """White-noise frames: worst case for every integer path (all byte values, no smoothness)."""
    return np.random.default_rng(seed).integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)


def bgr_to_nv12(frames: np.ndarray):
    """Synthetic decoder surfaces for the NV12 ingest path: BT.601 limited-range Y plane uint8[n,h,w] and an
    interleaved U,V plane uint8[n,h/2,w] (2x2 chroma average) of BGR frames uint8[n,h,w,3] (h, w even).
    This is only a data generator (a forward transform in float); the inverse the product implements is
    libswscale's integer conversion."""
    f = frames.astype(np.float32)
    b, g, r = f[..., 0], f[..., 1], f[..., 2]
    y = 16.0 + (65.481 * r + 128.553 * g + 24.966 * b) / 255.0
    u = 128.0 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255.0
    v = 128.0 + (112.0 * r - 93.786 * g - 18.214 * b) / 255.0
    n, h, w = y.shape
    sub = lambda c: c.reshape(n, h // 2, 2, w // 2, 2).mean(axis=(2, 4))
    uv = np.empty((n, h // 2, w), np.uint8)
    uv[..., 0::2] = np.clip(np.rint(sub(u)), 0, 255).astype(np.uint8)
    uv[..., 1::2] = np.clip(np.rint(sub(v)), 0, 255).astype(np.uint8)
    return np.clip(np.rint(y), 0, 255).astype(np.uint8), uv
