#!/usr/bin/env python3
"""One-off randomized parity soak on the GPU box: random geometries / contents / clip lengths, HIP path vs
oracle, everything bit-exact.  python tools/soak_parity.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    sys.path.insert(0, p)
import numpy as np
import avd_hip
from avd_hip import synth
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
ctx = avd_hip.Context(0)
t0 = time.time()
for i in range(cases):
    h = int(rng.integers(32, 600)); w = int(rng.integers(32, 900))
    if rng.random() < 0.4:
        w = (w // 16 + 1) * 16                      # aligned fast path
    n = int(rng.integers(1, 7)) if rng.random() < 0.7 else int(rng.integers(8, 42))     # some longer clips: pair counts around the XCD run length
    if n > 7:
        h, w = min(h, 160), min(w, 208)
    kind = rng.integers(0, 3)
    if kind == 0:
        clip = synth.random_frames(n, h, w, seed=int(rng.integers(1 << 30)))
    elif kind == 1:
        clip = synth.make_clip(n, h, w, seed=int(rng.integers(1 << 30)), dup_every=int(rng.integers(2, 5)))
    else:
        base = synth.make_clip(1, h, w, seed=int(rng.integers(1 << 30)))[0]
        clip = np.stack([np.roll(base, int(rng.integers(-20, 20)) * k, axis=int(rng.integers(0, 2))) for k in range(n)])
    rec = ctx.analyze_frames(clip)
    small, hsh, s, q = O.preprocess_bgr(clip)
    fm, fv = O.farneback_pairs(small)
    ok = np.array_equal(rec["lap_sum"], s) and np.array_equal(rec["lap_sumsq"], q)
    ok &= np.array_equal(rec["flow_mean"][1:], fm) and np.array_equal(rec["flow_var"][1:], fv)
    ham = np.array([-1] + [int(np.sum(hsh[k] ^ hsh[k - 1])) for k in range(1, n)])
    ok &= np.array_equal(rec["ham"], ham)
    print(f"case {i:3d} n={n} {h}x{w} kind={kind}: {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
print(f"soak ok: {cases} cases in {time.time() - t0:.1f} s")
