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
    assert b.algorithmic_bytes_per_frame(1080, 1920) == 6_324_240
    assert b.algorithmic_bytes_per_frame(720, 1280) == 2_868_240
    assert b.algorithmic_bytes_per_frame(2160, 3840) == 24_986_640


def test_farneback_kernel_roofline_objects():
    b = _bench()
    m = b.farneback_model(120, stage_ms=2.26, uv_ms=0.30, hscan_ms=0.14, ms_per_step=2.6)
    uv, hs = m["k_uv_320"], m["k_hscan_320"]
    assert uv["algorithmic_bytes_per_launch"] == 119 * 320 * 320 * 88       # R0 20 + R1 20 + flow 8 read, D 40 written
    assert hs["algorithmic_bytes_per_launch"] == 119 * 320 * 320 * 48       # D 40 read, flow 8 written
    for k in (uv, hs):
        assert k["bound"] == "hbm" and k["unit"] == "GB/s" and k["peak"] == 8000.0
        assert abs(k["frac"] - k["achieved"] / k["peak"]) < 1e-3
        assert k["launches_per_step"] == 3
    assert abs(uv["achieved"] - uv["algorithmic_bytes_per_launch"] / 0.30e-3 / 1e9) < 1.0
    assert m["design_traffic_bytes"] == 119 * (320 * 320 + 160 * 160 + 80 * 80 + 40 * 40) * 3 * 136


def test_cli_defaults_are_the_driver_contract():
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag, default in (("--gpus", "1"), ("--steps", "20"), ("--warmup", "3"), ("--inflight", "3")):
        assert re.search(r'add_argument\("%s", type=int, default=%s' % (re.escape(flag), default), src), flag
    # one JSON line on stdout, printed by rank 0 only
    assert src.count("print(json.dumps(out))") == 1
