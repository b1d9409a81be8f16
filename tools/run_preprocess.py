#!/usr/bin/env python3
"""Run only the fused preprocess kernel on a resident 1080p clip (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    sys.path.insert(0, p)
import numpy as np
import avd_hip
from avd_hip import synth
avd_hip.load()
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
clip = synth.random_frames(4, h, w, seed=1)
frames = torch.from_numpy(np.concatenate([clip] * (n // 4))).to("cuda:0")
ctx = avd_hip.Context(0)
for _ in range(reps):
    ctx.preprocess_bgr(frames)
print("done", frames.shape)
