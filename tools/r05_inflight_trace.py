"""Round 5: the bench's in-flight loop (one host thread, m contexts) over a resident 120 x 1080p clip with `pairs` stripe pairs, for
rocprofv3 --kernel-trace (tools/r04_idle.py, tools/timeline.py read the trace).  Usage: python tools/r05_inflight_trace.py PAIRS M STEPS"""
import math
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
from avd_hip import synth  # noqa: E402


def main():
    pairs, m, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    n, h, w = 120, 1080, 1920
    dev = torch.device("cuda", 0)
    v = torch.from_numpy(synth.make_clip(n, h, w, seed=0, dup_every=10)).to(dev)
    xs = torch.arange(w, device=dev, dtype=torch.float32)
    stride = max(3, (n - 2) // max(pairs, 1))
    for f in (list(range(n)) if pairs >= n - 1 else [f for j in range(pairs) for f in (1 + j * stride, 2 + j * stride)]):
        row = (127.0 + 120.0 * torch.sin((xs + 7.0 * f) * (2.0 * math.pi / 60.0))).round().clamp(0, 255).to(torch.uint8)
        v[f] = row[None, :, None].expand(h, w, 3)
    ctxs = [avd_hip.Context(0) for _ in range(m)]
    recs = [np.zeros(n, avd_hip.RECORD_DTYPE) for _ in range(m)]
    for j in range(m):
        for _ in range(2):
            ctxs[j].analyze_frames_async(v, recs[j]); ctxs[j].synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    q = []
    log = []
    for i in range(steps):
        if len(q) == m:
            a = time.perf_counter(); k = q.pop(0); ctxs[k].synchronize(); log.append(("sync", k, a, time.perf_counter()))
        j = i % m
        a = time.perf_counter(); ctxs[j].analyze_frames_async(v, recs[j]); q.append(j); log.append(("submit", j, a, time.perf_counter()))
    while q:
        a = time.perf_counter(); k = q.pop(0); ctxs[k].synchronize(); log.append(("sync", k, a, time.perf_counter()))
    if "--host" in sys.argv:
        for what, k, a, b in log:
            print(f"{(a - t1) * 1e6:9.1f} {(b - t1) * 1e6:9.1f} {(b - a) * 1e6:8.1f} us  ctx {k} {what}")
    print(f"pairs {pairs} m {m}: {steps * n / (time.perf_counter() - t1) / 1e3:.1f} k frames/s, flagged {ctxs[0].get_option('rerun_pairs')}")
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
