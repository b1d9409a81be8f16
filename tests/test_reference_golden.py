"""Host tail of the hot path (fusion / hints) against golden vectors produced by the
REFERENCE's own modules (tests/golden/make_reference_golden.py).  Bit-exact."""
import copy
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def fusion_mod():
    from app.analyzers import fusion
    return fusion


def test_thresholds_default(fusion_mod):
    g = _load("fusion_golden.json")
    assert fusion_mod.THRESH_REAL_MAX == g["thresholds"]["THRESH_REAL_MAX"]
    assert fusion_mod.THRESH_AI_MIN == g["thresholds"]["THRESH_AI_MIN"]


def test_fuse_matches_reference_bit_exact(fusion_mod):
    g = _load("fusion_golden.json")
    assert len(g["fuse"]) > 100
    labels = set()
    for case in g["fuse"]:
        audio, video, hints = (copy.deepcopy(case[k]) for k in ("audio", "video", "hints"))
        out = fusion_mod.fuse(audio, video, hints)
        assert out == case["out"], case["note"]
        # in-place padding side effect of the reference (fusion.py:19-21)
        assert video.get("timeline") == case["video_timeline_after"], case["note"]
        assert audio.get("timeline") == case["audio_timeline_after"], case["note"]
        labels.add(out["result"]["label"])
    assert labels == {"real", "ai", "uncertain"}


def test_fuse_keeps_list_aliasing(fusion_mod):
    """video.analyze returns ONE list under both keys; fuse extends it in place."""
    tl = [0.2, 0.4]
    video = {"timeline": tl, "summary": {}, "timeline_ai": tl}
    audio = {"timeline": [0.5] * 5, "flags_audio": {"error": "X"}}
    fusion_mod.fuse(audio, video, {})
    assert video["timeline"] is video["timeline_ai"] and len(tl) == 5 and tl[2:] == [0.4] * 3


def test_bin_timeline_matches_reference(fusion_mod):
    for case in _load("fusion_golden.json")["bin_timeline"]:
        assert fusion_mod._bin_timeline(list(case["in"])) == case["out"]


def test_compute_hints_matches_reference():
    from app.analyzers import heuristics_v2
    g = _load("hints_golden.json")
    classes = set()
    for case in g["cases"]:
        out = heuristics_v2.compute_hints(dict(case["meta"]), "/tmp/x.mp4")
        assert out == case["out"]
        assert list(out.keys()) == list(case["out"].keys())
        classes.add(out["compression"])
    assert classes == {"very_heavy", "heavy", "normal", "light"}
