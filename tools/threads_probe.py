#!/usr/bin/env python3
"""Aggregate throughput of T host threads, each with its own avd context, analysing the same resident clip in a
loop (the reference's serving model: api.py runs the analyzer on worker threads).  python tools/threads_probe.py T [steps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    sys.path.insert(0, p)
import numpy as np
import avd_hip
from avd_hip import synth
avd_hip.load()
import torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
clip = synth.make_clip(120, 1080, 1920, seed=0)
frames = torch.from_numpy(clip).to("cuda:0")
ctxs = [avd_hip.Context(0) for _ in range(T)]
for c in ctxs:
    c.analyze_frames(frames)
bar = threading.Barrier(T + 1)
def work(c):
    bar.wait()
    for _ in range(steps):
        c.analyze_frames(frames)
ths = [threading.Thread(target=work, args=(c,)) for c in ctxs]
for t in ths: t.start()
torch.cuda.synchronize(); bar.wait(); t0 = time.perf_counter()
for t in ths: t.join()
dt = time.perf_counter() - t0
print(f"threads {T}: {T * steps * 120 / dt:.0f} frames/s, {dt / (T * steps) * 1e3:.3f} ms per clip")
