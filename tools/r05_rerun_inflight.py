"""Round 5: is the throughput of clips with flagged pairs limited by how many clips are in flight?  120 x 1080p resident clips with 0 / 1 / 12 / 119
stripe pairs, m = 3 / 4 / 6 contexts driven by one host thread (the bench's loop), plus which levels raised the flags.
Usage: python tools/r05_rerun_inflight.py
"""
import math
import os
import statistics
import sys
import threading
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
    steps = 24
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

    ctxs = [avd_hip.Context(0) for _ in range(6)]
    recs = [np.zeros(n, avd_hip.RECORD_DTYPE) for _ in range(6)]
    try:
        for pairs, helping in ((0, 1), (1, 1), (12, 1), (n - 1, 1)):     # 2: re-run off (what the content alone costs the fast pass)
            v = variant(pairs) if pairs else base
            for c in ctxs:
                c.set_option("tail_help", helping & 1)
                c.set_option("fb_rerun", 0 if helping == 2 else 1)
            for j in range(6):
                ctxs[j].analyze_frames_async(v, recs[j]); ctxs[j].synchronize()
            bits = {}
            for r in recs[0]["reserved"]:
                if r:
                    bits[int(r)] = bits.get(int(r), 0) + 1
            out = []
            for m in (2, 3, 4, 5, 6):
                thr = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    q = []
                    for i in range(steps):
                        if len(q) == m:
                            ctxs[q.pop(0)].synchronize()
                        j = i % m
                        ctxs[j].analyze_frames_async(v, recs[j]); q.append(j)
                    while q:
                        ctxs[q.pop(0)].synchronize()
                    thr.append(steps * n / (time.perf_counter() - t1))
                out.append(f"m={m}: {statistics.median(thr) / 1e3:6.1f} k")
            for m in (3, 4, 6):                       # one host thread per context (what a pool of borrowers does): each loops submit + drain on its own
                thr = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    per = steps // m

                    def work(j):
                        for _ in range(per):
                            ctxs[j].analyze_frames_async(v, recs[j]); ctxs[j].synchronize()
                    ths = [threading.Thread(target=work, args=(j,)) for j in range(m)]
                    t1 = time.perf_counter()
                    for t in ths:
                        t.start()
                    for t in ths:
                        t.join()
                    thr.append(per * m * n / (time.perf_counter() - t1))
                out.append(f"threads={m}: {statistics.median(thr) / 1e3:6.1f} k")
            print(f"pairs replaced {pairs:3d} tail_help {helping}: flagged {ctxs[0].get_option('rerun_pairs'):3d}  flag words {bits}  frames/s " + "  ".join(out), flush=True)
    finally:
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
