#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own pure-numpy modules.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_reference_golden.py

Loads reference app/analyzers/fusion.py and heuristics_v2.py BY FILE PATH (importing the
package would pull cv2 through app/analyzers/__init__.py), calls ``fuse`` / ``_bin_timeline`` /
``compute_hints`` over a grid of inputs that straddles every threshold in those files, and
writes inputs + outputs as JSON fixtures (data only; no reference source is copied):
    tests/golden/fusion_golden.json   tests/golden/hints_golden.json
Python's JSON float repr round-trips exactly, so comparisons in tests are bit-exact.
"""
import copy
import importlib.util
import itertools
import json
import os
import random

REF = os.environ.get("AVD_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, "app", "analyzers", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    fusion = load("fusion")
    hx = load("heuristics_v2")
    rng = random.Random(1234)

    def tl(n, lo, hi):
        return [rng.uniform(lo, hi) for _ in range(n)]

    cases = []

    def add(audio, video, hints, note):
        a, v, h = copy.deepcopy(audio), copy.deepcopy(video), copy.deepcopy(hints)
        out = fusion.fuse(a, v, h)
        cases.append({"note": note, "audio": audio, "video": video, "hints": hints, "out": out,
                      "video_timeline_after": v.get("timeline"), "audio_timeline_after": a.get("timeline")})

    # 1. the always-present case in this deployment: audio analyzer failed (audio.py:112-118 / api.py:124)
    def audio_err(n):
        return {"scores": {}, "flags_audio": {"error": "RuntimeError"}, "timeline": [0.5] * n}

    summaries = []
    for flow_mean, texture_var, sc, dup in itertools.product(
            (0.0, 4.99, 5.01, 8.01, 20.0), (0.0, 199.0, 201.0, 299.0, 301.0),
            (0.0, 0.69, 0.71, 0.9, 1.0), (0.0, 0.019, 0.021, 0.049, 0.051, 0.21, 0.26)):
        summaries.append({"dup_density": dup, "scene_change_rate": sc, "flow_mean": flow_mean,
                          "flow_var": 1.0, "texture_var": texture_var, "w": 1920, "h": 1080, "fps": 30.0})
    rng.shuffle(summaries)
    hint_variants = [
        {"compression": "normal", "bpp": 0.1286, "dup_avg": 0.0, "video_has_signal": True},
        {"compression": "heavy", "bpp": 0.06, "dup_avg": 0.0, "video_has_signal": True},
        {"compression": "very_heavy", "bpp": 0.01, "dup_avg": 0.0, "video_has_signal": False},
        {"compression": "light", "bpp": 0.3, "dup_avg": 0.25, "video_has_signal": True},
        {},
    ]
    for i, vs in enumerate(summaries[:90]):
        n = rng.choice((1, 2, 3, 10, 24))
        lo, hi = rng.choice(((0.0, 0.2), (0.3, 0.7), (0.8, 1.0), (0.0, 1.0)))
        vt = tl(n, lo, hi)
        add(audio_err(n), {"timeline": vt, "summary": vs, "timeline_ai": vt}, hint_variants[i % len(hint_variants)],
            "audio-error path")

    # 2. real audio flags: speech ratio / tts_like straddling 0.25, 0.6, 0.95
    for sr, tts in itertools.product((0.0, 0.2, 0.25, 0.3, 1.0), (0.0, 0.6, 0.61, 0.7, 0.95, 0.96)):
        for vs in summaries[90:92]:
            na, nv = rng.choice(((10, 10), (7, 12), (12, 7), (1, 5), (20, 20)))
            lo, hi = rng.choice(((0.0, 0.3), (0.4, 0.6), (0.7, 1.0)))
            audio = {"scores": {"tts_like": tts}, "flags_audio": {"speech_ratio": sr, "tts_like": tts},
                     "timeline": tl(na, lo, hi)}
            vt = tl(nv, *rng.choice(((0.0, 0.3), (0.4, 0.6), (0.7, 1.0))))
            add(audio, {"timeline": vt, "summary": vs, "timeline_ai": vt}, rng.choice(hint_variants), "audio flags")

    # 3. degenerate shapes: empty lists, missing keys, exact 0.5 means (sign(0) == sign(0) agreement)
    add({}, {}, {}, "all empty")
    add({"timeline": []}, {"timeline": [], "timeline_ai": [0.9, 0.8], "summary": {}}, {}, "video falls back to timeline_ai")
    add({"timeline": [0.5, 0.5]}, {"timeline": [0.5], "summary": None}, {}, "neutral both, summary None")
    add(audio_err(3), {"timeline": [0.5] * 3, "summary": {"error": "AvdError"}, "timeline_ai": [0.5] * 3},
        {"compression": "normal", "bpp": 0.1, "video_error": "AvdError"}, "video-error neutral path (api.py:136)")
    add({"timeline": [1.0] * 5, "flags_audio": {"speech_ratio": 1.0, "tts_like": 0.99}},
        {"timeline": [1.0] * 5, "summary": {"dup_density": 0.3}}, {"video_has_signal": False}, "saturated ai")
    add({"timeline": [0.0] * 4, "flags_audio": {"speech_ratio": 1.0}},
        {"timeline": [0.0] * 4, "summary": {"dup_density": 0.3}}, {"compression": "heavy", "bpp": 0.05}, "saturated real")

    bins = []
    for ts in ([], [0.3], [0.2, 0.9], [0.1, 0.5, 0.9], [1.2, -0.3, 0.5, 0.5], tl(17, 0, 1), tl(60, 0.4, 0.6)):
        bins.append({"in": ts, "out": fusion._bin_timeline(list(ts))})

    with open(os.path.join(HERE, "fusion_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_reference_golden.py", "reference": "app/analyzers/fusion.py",
                   "thresholds": {"THRESH_REAL_MAX": fusion.THRESH_REAL_MAX, "THRESH_AI_MIN": fusion.THRESH_AI_MIN},
                   "fuse": cases, "bin_timeline": bins}, f)

    hints = []
    metas = [{}, {"width": 1920, "height": 1080, "fps": 30.0, "bit_rate": 8_000_000},
             {"width": 1280, "height": 720, "fps": 30.0, "bit_rate": 0},
             {"width": 0, "height": 0, "fps": 0.0, "bit_rate": 5_000_000},
             {"width": 3840, "height": 2160, "fps": 29.97, "bit_rate": 45_000_000}]
    for w, h, fps in ((1920, 1080, 30.0), (1280, 720, 25.0), (640, 360, 23.976)):
        for bpp in (0.0, 0.0399, 0.04, 0.0401, 0.0799, 0.08, 0.0801, 0.1499, 0.15, 0.1501, 1.0):
            metas.append({"width": w, "height": h, "fps": fps, "bit_rate": int(bpp * w * h * fps), "duration": 10.0,
                          "vcodec": "h264", "acodec": None, "format_name": "mov,mp4"})
    for m in metas:
        hints.append({"meta": m, "out": hx.compute_hints(dict(m), "/tmp/x.mp4")})
    with open(os.path.join(HERE, "hints_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_reference_golden.py", "reference": "app/analyzers/heuristics_v2.py",
                   "cases": hints}, f)
    print(f"wrote {len(cases)} fuse cases, {len(bins)} bin cases, {len(hints)} hint cases")


if __name__ == "__main__":
    main()
