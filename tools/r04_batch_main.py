#!/usr/bin/env python3
"""Experiment: the bench clip (120 x 1080p) handed over k clips per call (avd_analyze_batch: one launch sequence over all pairs of the
batch), m calls in flight -- against one clip per call.  Prints frames/s per (k, m)."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    sys.path.insert(0, p)
import numpy as np
import avd_hip
from avd_hip import synth
avd_hip.load()
import torch
dev = torch.device("cuda", 0)
clip = synth.make_clip(120, 1080, 1920, seed=1)
copies = [torch.from_numpy(clip).to(dev) for _ in range(6)]
for k, m in ((1, 3), (2, 2), (2, 3), (3, 2), (3, 3), (4, 2), (6, 1), (6, 2)):
    ctxs = [avd_hip.Context(0) for _ in range(m)]
    recs = [np.zeros(120 * k, avd_hip.RECORD_DTYPE) for _ in range(m)]
    def call(j):
        if k == 1:
            ctxs[j].analyze_frames_async(copies[j], recs[j])
        else:
            ctxs[j].analyze_batch_async([copies[(j * k + i) % 6] for i in range(k)], recs[j])
    for j in range(m):
        call(j); ctxs[j].synchronize()
    vals = []
    rounds = max(6, 36 // k)
    for _ in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        q = []
        for i in range(rounds):
            if len(q) == m:
                ctxs[q.pop(0)].synchronize()
            call(i % m); q.append(i % m)
        while q:
            ctxs[q.pop(0)].synchronize()
        vals.append(rounds * k * 120 / (time.perf_counter() - t))
    print("clips per call %d, calls in flight %d: %.0f frames/s (%.0f..%.0f)" % (k, m, statistics.median(vals), min(vals), max(vals)), flush=True)
    for c in ctxs:
        c.close()
