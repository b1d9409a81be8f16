"""Host side of the CNN extension (csrc/avd_cnn.hip): the topology the library is built for,
its parameter order, seeded synthetic parameters (there are no trained weights: the reference has no learned model) and
the arithmetic count used by bench.py.  Never part of ai_score."""
from __future__ import annotations

import numpy as np

DEPTH = (3, 4, 6, 3)
CLASSES = 1000


def layers():
    """[(role, cin, cout, ksize, stride, out_side)] in the order of avd_cnn_set_weights (the linear layer excluded)."""
    out = [("stem", 3, 64, 7, 2, 112)]
    cin, side = 64, 56
    for stage, depth in enumerate(DEPTH):
        mid, wide = 64 << stage, (64 << stage) * 4
        for block in range(depth):
            stride = 2 if (block == 0 and stage > 0) else 1
            out.append(("reduce", cin, mid, 1, 1, side))
            side //= stride
            out.append(("spatial", mid, mid, 3, stride, side))
            out.append(("expand", mid, wide, 1, 1, side))
            if block == 0:
                out.append(("shortcut", cin, wide, 1, stride, side))
            cin = wide
    return out


def macs_per_frame() -> int:
    """multiply-accumulates of one 224 x 224 frame (convolutions + the linear layer)."""
    return sum(cout * k * k * cin * side * side for _, cin, cout, k, _, side in layers()) + 2048 * CLASSES


def seeded_parameters(seed: int = 0):
    """(weights float32 flat, biases float32 flat): He-style normal weights, the expanding convolutions and shortcuts
    damped by 0.5 so that the residual sums stay bounded without batch norm."""
    rng = np.random.default_rng(seed)
    ws, bs = [], []
    for role, cin, cout, k, _, _ in layers():
        std = np.sqrt(2.0 / (cin * k * k)) * (0.5 if role in ("expand", "shortcut") else 1.0)
        ws.append((rng.standard_normal((cout, k, k, cin), dtype=np.float32) * np.float32(std)).ravel())
        bs.append(rng.standard_normal(cout, dtype=np.float32) * np.float32(0.05))
    ws.append((rng.standard_normal((CLASSES, 2048), dtype=np.float32) * np.float32(np.sqrt(1.0 / 2048))).ravel())
    bs.append(rng.standard_normal(CLASSES, dtype=np.float32) * np.float32(0.05))
    return np.concatenate(ws), np.concatenate(bs)
