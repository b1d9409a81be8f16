#!/usr/bin/env python3
"""Run the ViT patch-embed extension on resident frames (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    sys.path.insert(0, p)
import numpy as np
import avd_hip
from avd_hip import synth
avd_hip.load()
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 960
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
clip = synth.random_frames(4, 360, 640, seed=1)
frames = torch.from_numpy(np.concatenate([clip] * (n // 4))).to("cuda:0")
ctx = avd_hip.Context(0)
rng = np.random.default_rng(7)
ctx.vit_set_weights((rng.standard_normal((768, 768)) * 0.02).astype(np.float32), (rng.standard_normal(768) * 0.1).astype(np.float32))
tok = torch.empty((n, 196, 768), dtype=torch.bfloat16, device="cuda:0")
_, ms = ctx.vit_patch_embed(frames, timing_reps=reps, out=tok, bf16=True)
print("gemm %.4f ms for %d frames" % (ms, n))
