"""Compression hints from container metadata -- restates reference app/analyzers/heuristics_v2.py.

Pure scalar host code (an input of ``fusion.fuse``); pinned bit-for-bit against vectors produced
by the reference's own module (tests/golden/hints_golden.json).
"""
from __future__ import annotations

# upper bounds (inclusive) of bits-per-pixel classes, heuristics_v2.py:9-12
_BPP_CLASSES = ((0.04, "very_heavy"), (0.08, "heavy"), (0.15, "normal"))


def _compression_class(bpp: float) -> str:
    for bound, name in _BPP_CLASSES:
        if bpp <= bound:
            return name
    return "light"


def compute_hints(meta: dict, path: str) -> dict:
    w = meta.get("width") or 0
    h = meta.get("height") or 0
    fps = meta.get("fps") or 0.0
    bit_rate = meta.get("bit_rate") or 0
    pixel_rate = (w * h * fps) if (w and h and fps) else 0.0
    bpp = float(bit_rate) / max(1.0, pixel_rate)
    return {
        "w": w, "h": h, "fps": fps, "br": bit_rate,
        "bpp": round(bpp, 5),
        "compression": _compression_class(bpp),
        "video_has_signal": (w * h) > 0 and fps > 0,
        "dup_avg": 0.0,       # the reference hard-codes this (heuristics_v2.py:18)
    }
