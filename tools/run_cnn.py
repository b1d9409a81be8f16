#!/usr/bin/env python3
"""Run the CNN extension's forward pass on resident frames (for rocprofv3 kernel traces / --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    sys.path.insert(0, p)
import numpy as np
import avd_hip
from avd_hip import synth, cnn
avd_hip.load()
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
clip = synth.random_frames(4, 360, 640, seed=1)
frames = torch.from_numpy(np.concatenate([clip] * (n // 4))).to("cuda:0")
ctx = avd_hip.Context(0)
ctx.set_option("cnn_chunk", int(os.environ.get("AVD_CNN_CHUNK", "128")))     # frames per forward pass
if os.environ.get("AVD_CNN_FUSE"):
    ctx.set_option("cnn_fuse", int(os.environ["AVD_CNN_FUSE"]))
ctx.cnn_set_weights(*cnn.seeded_parameters(0))
logits, ms = ctx.cnn_forward(frames, timing_reps=reps)
print("forward %.3f ms for %d frames" % (ms, n))
