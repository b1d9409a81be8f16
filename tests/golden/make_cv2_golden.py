#!/usr/bin/env python3
"""Capture stage-level golden vectors of the hot path from a REAL OpenCV (NOT run in the build image).

The reference pins ``opencv-python-headless>=4.10,<4.11`` (requirements.txt:8); that wheel is neither installed
nor installable in the build container, so parity at the OpenCV boundary is UNPINNED (oracle/README.md).  On any
machine that has the pinned wheel and numpy:

    pip install 'opencv-python-headless>=4.10,<4.11' 'numpy>=1.26,<2.0'
    python tests/golden/make_cv2_golden.py            # writes tests/golden/cv2_stage_golden.npz (~2.5 MB)

and commit the .npz.  ``tests/test_cv2_golden.py`` then compares the oracle against it stage by stage (and on a
GPU box the HIP path through the same vectors), which turns "HIP == oracle" into "HIP == cv2".  Each stage is the
exact call ``app/analyzers/video.py`` makes (line cited), on seeded synthetic frames that need nothing but numpy
to regenerate (avd_hip.synth is deterministic), so only OUTPUTS are stored, plus tiny inputs for self-checking:

  gray        cv2.cvtColor(frame, COLOR_BGR2GRAY)                              video.py:5,43,51
  area32      cv2.resize(gray, (32, 32), interpolation=INTER_AREA)            video.py:6
  hash        (area32 >= area32.mean())                                       video.py:7-8
  small320    cv2.resize(gray, (320, 320))                                    video.py:43
  lap_var     cv2.Laplacian(gray, CV_64F).var()                               video.py:52
  lap_sum/sq  exact integer moments of the same CV_64F image
  flow        cv2.calcOpticalFlowFarneback(prev, cur, None, .5, 3, 15, 3, 5, 1.2, 0)   video.py:45  (two pairs, full field)
  flow_mean/var  np.mean / np.var of sqrt(fx^2 + fy^2)                        video.py:46-48
  pyr_probe   the 4-vs-3 pyramid-scale question: Farneback with levels=2 and levels=4 on one pair (means only)

Geometries: 720p and 1080p (non-integer INTER_AREA scales, the deployed case), 640x640 (the exact-2x INTER_AREA
reroute of INTER_LINEAR), 96x128 (integer-scale area-fast), 67x101 (odd sizes), 240x426 (y up-scaling to 320).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "ai-video-detector_amd"))

GEOMS = [(720, 1280), (1080, 1920), (640, 640), (96, 128), (67, 101), (240, 426)]
FRAMES = 3          # per geometry: two Farneback pairs, one of them across a scene cut at 1080p


def clips():
    from avd_hip import synth          # numpy only
    for (h, w) in GEOMS:
        smooth = synth.make_clip(FRAMES, h, w, seed=7000 + h, dup_every=0, scene_cut=(h == 1080))
        noise = synth.random_frames(1, h, w, seed=h * 7 + w)
        yield (h, w), np.concatenate([smooth, noise])      # 3 smooth frames + 1 white-noise frame (worst case for the integer paths)


def main():
    import cv2
    out = {}
    info = {"cv2": cv2.__version__, "numpy": np.__version__, "geoms": GEOMS, "frames_per_geom": FRAMES + 1,
            "ipp": bool(getattr(cv2, "ipp", None) and cv2.ipp.useIPP()),
            "cpu_features": cv2.getCPUFeaturesLine() if hasattr(cv2, "getCPUFeaturesLine") else "",
            "generator": "tests/golden/make_cv2_golden.py", "reference_sites": "app/analyzers/video.py:4-8,36-52"}
    for (h, w), frames in clips():
        key = f"{h}x{w}"
        gray = np.stack([cv2.cvtColor(f, cv2.COLOR_BGR2GRAY) for f in frames])
        area = np.stack([cv2.resize(g, (32, 32), interpolation=cv2.INTER_AREA) for g in gray])
        small = np.stack([cv2.resize(g, (320, 320)) for g in gray])
        lap = [cv2.Laplacian(g, cv2.CV_64F) for g in gray]
        out[key + "/gray_crc"] = np.array([int(np.sum(g.astype(np.uint64) * (np.arange(g.size, dtype=np.uint64).reshape(g.shape) % 251 + 1)))
                                           for g in gray], np.uint64)          # position-weighted checksum (full gray is big)
        out[key + "/gray_rows"] = gray[:, ::max(1, h // 8), :]                  # a few full rows, exact
        out[key + "/area32"] = area
        out[key + "/hash"] = np.stack([(a >= a.mean()).astype(np.uint8).flatten() for a in area])
        out[key + "/small320"] = small
        out[key + "/lap_var"] = np.array([l.var() for l in lap], np.float64)
        out[key + "/lap_sum"] = np.array([int(l.sum()) for l in lap], np.int64)
        out[key + "/lap_sumsq"] = np.array([int((l * l).sum()) for l in lap], np.int64)
        means, vars_, flows = [], [], []
        for i in range(1, len(small)):
            flow = cv2.calcOpticalFlowFarneback(small[i - 1], small[i], None, 0.5, 3, 15, 3, 5, 1.2, 0)
            mag = np.sqrt(flow[..., 0] ** 2 + flow[..., 1] ** 2)
            means.append(np.mean(mag))
            vars_.append(np.var(mag))
            if (h, w) in ((720, 1280), (1080, 1920)) and i <= 2:
                flows.append(flow.astype(np.float32))
        out[key + "/flow_mean"] = np.array(means, np.float32)
        out[key + "/flow_var"] = np.array(vars_, np.float32)
        if flows:
            out[key + "/flow"] = np.stack(flows)
        if (h, w) == (720, 1280):
            probe = {}
            for lv in (2, 3, 4):
                fl = cv2.calcOpticalFlowFarneback(small[0], small[1], None, 0.5, lv, 15, 3, 5, 1.2, 0)
                probe[lv] = float(np.mean(np.sqrt(fl[..., 0] ** 2 + fl[..., 1] ** 2)))
            info["pyr_probe_flow_mean_by_levels"] = probe
    out["info"] = np.frombuffer(json.dumps(info).encode(), np.uint8)
    path = os.path.join(HERE, "cv2_stage_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", json.dumps(info))


if __name__ == "__main__":
    main()
