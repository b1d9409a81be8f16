"""bench.py pieces that do not need a GPU: the roofline bookkeeping (SURVEY.md 8d byte counts) and
the command-line contract of the driver (`python bench.py --gpus N --steps K --warmup W`, defaults that finish in
minutes)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_algorithmic_bytes_match_the_survey():
    b = _bench()
    assert b.preprocess_bytes_per_frame(1080, 1920) == 6_324_240
    assert b.preprocess_bytes_per_frame(720, 1280) == 2_868_240
    assert b.preprocess_bytes_per_frame(2160, 3840) == 24_986_640


def test_roofline_objects():
    """The dominant kernel (fused Farneback level 0) is `roofline`; its algorithmic bytes count every frame's
    polynomial expansion ONCE per iteration (a frame is R0 of one pair and R1 of the next) plus the flow read and
    written once -- never more than the kernel can have moved."""
    b = _bench()
    per_iter = 120 * 320 * 320 * 20 + 119 * 320 * 320 * 16
    assert b.fused_level_bytes(120) == 3 * per_iter
    assert per_iter < 119 * 320 * 320 * (20 + 20 + 8 + 8)              # R once per FRAME, not once per pair
    stage = [0.158, 0.027, 1.9, 0.006, 1.0, 0.0]
    dom, pre, fb = b.roofline_objects(120, 1080, 1920, stage, latency_ms=2.2)
    for k in (dom, pre):
        assert k["bound"] == "hbm" and k["unit"] == "GB/s" and k["peak"] == 8000.0
        assert abs(k["frac"] - k["achieved"] / k["peak"]) < 1e-3
        assert k["traffic"] is None or k["traffic"] >= 0.9 * k["algorithmic_bytes_per_launch"]   # algorithmic <= measured
    assert abs(dom["achieved"] - 3 * per_iter / 1.0e-3 / 1e9) < 1.0 and dom["launches_per_step"] == 1
    assert abs(pre["achieved"] - 120 * 6_324_240 / 0.158e-3 / 1e9) < 1.0
    assert "k_fb_level<320>" in dom["kernel"] and 0 < dom["share_of_step"] < 1
    assert fb["algorithmic_bytes"] == sum(b.fused_level_bytes(120, 3, w) for w in (320, 160, 80, 40))


def test_cli_defaults_are_the_driver_contract():
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag, default in (("--gpus", "1"), ("--steps", "20"), ("--warmup", "3"), ("--inflight", "3"), ("--repeats", "25")):
        assert re.search(r'add_argument\("%s", type=int, default=%s' % (re.escape(flag), default), src), flag
    # one JSON line on stdout, printed by rank 0 only
    assert src.count("print(json.dumps(out))") == 1
