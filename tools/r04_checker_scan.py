#!/usr/bin/env python3
"""One-off: every "checker" pair of the 1 440-pair soak (AVD_SOAK_PER_FAMILY=60) -- cell size, shift, oracle flow_mean / flow_var, the default
mode's deviation in both shapes of the 160-px level, the oracle's own +-1-ulp sensitivity, whether the pair was re-run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    sys.path.insert(0, p)
import numpy as np
import avd_hip
from oracle import oracle
from tests.content_families import families
fam = families()
names = list(fam)
j = names.index("checker")
lib = oracle.lib()
rows = []
for i in range(60):
    seed = 9000 + 7919 * i + 17 * j
    rng = np.random.default_rng(seed)
    c = int(np.random.default_rng(seed).choice([2, 4, 8, 16, 32]))
    a, b = fam["checker"](np.random.default_rng(seed))
    frames = np.stack([a, b])
    m, v = oracle.flow_stats(oracle.farneback(a, b))
    out = []
    for wide in (1, 0):
        with avd_hip.Context(0) as ctx:
            ctx.set_option("fb_wide160", wide)
            fm, fv = ctx.farneback_pairs(frames)
            out.append((float(fm[0]), float(fv[0]), ctx.get_option("rerun_pairs")))
    sens_m = sens_v = 0.0
    try:
        for model in (2, 4):
            lib.avdo_set_model(model)
            mm, vv = oracle.flow_stats(oracle.farneback(a, b))
            sens_m = max(sens_m, abs(float(mm) - float(m))); sens_v = max(sens_v, abs(float(vv) - float(v)))
    finally:
        lib.avdo_set_model(0)
    rel = lambda x, y: abs(x - y) / max(abs(y), 1e-30)
    flag = "  <--" if max(rel(out[0][0], m), rel(out[0][1], v), rel(out[1][0], m), rel(out[1][1], v)) > 1e-6 else ""
    print("seed %7d cell %2d  oracle mean %.6g var %.6g | wide: dmean %.2e dvar %.2e rerun %d | narrow: dmean %.2e dvar %.2e rerun %d | oracle +-1ulp: mean %.2e var %.2e%s"
          % (seed, c, m, v, rel(out[0][0], m), rel(out[0][1], v), out[0][2], rel(out[1][0], m), rel(out[1][1], v), out[1][2], sens_m / max(abs(m), 1e-30), sens_v / max(abs(v), 1e-30), flag), flush=True)
