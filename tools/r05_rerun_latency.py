"""Round 5 experiment: what does an EXACT Farneback pass cost for 1 .. 119 pairs, per path?
(fused one-workgroup-per-pair kernels, the two-kernel path, and the fast kernels for comparison.)
Resident frames, median of `reps` calls of avd_farneback_pairs (which drains the stream and copies the statistics out).
Usage: python tools/r05_rerun_latency.py [reps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import avd_hip  # noqa: E402
from tests.content_families import families  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    fam = families()
    rng = np.random.default_rng(5)
    frames = np.empty((120, 320, 320), np.uint8)
    for k in range(60):
        frames[2 * k], frames[2 * k + 1] = fam["smooth_shift"](rng)
    dev = torch.from_numpy(frames).cuda()
    modes = [("fast", dict(fb_mode=1, fb_rerun=0)), ("exact_fused", dict(fb_mode=0, fb_fused=0xF)), ("exact_two_kernel", dict(fb_mode=0, fb_fused=0))]
    print(f"{'path':18s}" + "".join(f"{n:>9d}" for n in (1, 2, 4, 12, 32, 119)) + "   pairs -> ms per call")
    for name, opts in modes:
        with avd_hip.Context(0) as c:
            for k, v in opts.items():
                c.set_option(k, v)
            row = []
            for npairs in (1, 2, 4, 12, 32, 119):
                t = dev[: npairs + 1]
                for _ in range(3):
                    c.farneback_pairs(t)
                ts = []
                for _ in range(reps):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    c.farneback_pairs(t)
                    ts.append((time.perf_counter() - t0) * 1e3)
                row.append(float(np.median(ts)))
            print(f"{name:18s}" + "".join(f"{x:9.3f}" for x in row))


if __name__ == "__main__":
    main()
