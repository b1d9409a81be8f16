import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    sys.path.insert(0, p)
import numpy as np, avd_hip
from avd_hip import synth
ctx = avd_hip.Context(0)
clip = synth.make_clip(120, 256, 256, seed=0)
ctx.analyze_frames(clip); ctx.analyze_frames(clip)
v = ctx.debug_fetch("vs0", (128 * 5 * 320 * 8,), np.float64)
off = 127 * 5 * 320 * 8
print("consumer: body %.0f barrier %.0f phases %.0f" % tuple(v[off:off + 3]))
print("producer: wait-gathers %.0f finish %.0f issue %.0f barrier %.0f" % tuple(v[off + 8:off + 12]))
