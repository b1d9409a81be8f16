"""Round 5: latency of one resident 120 x 1080p clip with k of its pairs flagged (stripe frames), per choice of the re-run's level mask
(fb_rerun_fused) -- and, with --trace N, N plain calls of the 1-pair case for rocprofv3 --kernel-trace.
Usage: python tools/r05_rerun_cost.py [--trace N]
"""
import math
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import avd_hip  # noqa: E402
from avd_hip import synth  # noqa: E402


def main():
    n, h, w = 120, 1080, 1920
    clip = synth.make_clip(n, h, w, seed=0, dup_every=10)
    dev = torch.device("cuda", 0)
    base = torch.from_numpy(clip).to(dev)
    xs = torch.arange(w, device=dev, dtype=torch.float32)

    def stripe(phase):
        row = (127.0 + 120.0 * torch.sin((xs + phase) * (2.0 * math.pi / 60.0))).round().clamp(0, 255).to(torch.uint8)
        return row[None, :, None].expand(h, w, 3)

    def variant(pairs):
        v = base.clone()
        if pairs >= n - 1:
            idx = list(range(n))
        else:
            stride = max(3, (n - 2) // max(pairs, 1))
            idx = [f for j in range(pairs) for f in (1 + j * stride, 2 + j * stride)]
        for f in idx:
            v[f] = stripe(7.0 * f)
        return v

    rec = np.zeros(n, avd_hip.RECORD_DTYPE)
    with avd_hip.Context(0) as c:
        def lat(v, reps=9):
            for _ in range(2):
                c.analyze_frames_async(v, rec); c.synchronize()
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                c.analyze_frames_async(v, rec); c.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            return statistics.median(ts), c.get_option("rerun_pairs")
        if "--trace" in sys.argv:
            v = variant(1)
            for _ in range(int(sys.argv[sys.argv.index("--trace") + 1])):
                c.analyze_frames_async(v, rec); c.synchronize()
            return
        t0, _ = lat(base)
        print(f"unflagged clip: {t0:.3f} ms")
        for pairs in (1, 4, 12, 30, 119):
            v = variant(pairs)
            row = []
            for mask in (0x8, 0xC, 0xE, 0xF):
                c.set_option("fb_rerun_fused", mask)
                t, m = lat(v)
                row.append(f"mask {mask:#x}: {t:.3f} ms (+{t - t0:.3f})")
            print(f"{pairs:3d} pairs replaced, {m:3d} flagged | " + " | ".join(row))
            c.set_option("fb_rerun_fused", 0xC)


if __name__ == "__main__":
    main()
