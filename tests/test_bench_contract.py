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
    """The dominant kernel (Farneback level 0) is `roofline`; its algorithmic bytes count every frame's polynomial
    expansion ONCE per iteration (a frame is R0 of one pair and R1 of the next) plus the flow read and written once --
    never more than the kernel can have moved.  fb_mode fast = one iteration per launch (three launches in stage 4),
    exact = all three iterations in one launch."""
    b = _bench()
    per_iter = 120 * 320 * 320 * 20 + 119 * 320 * 320 * 16
    assert b.fused_level_bytes(120) == 3 * per_iter and b.fused_level_bytes(120, 1) == per_iter
    assert per_iter < 119 * 320 * 320 * (20 + 20 + 8 + 8)              # R once per FRAME, not once per pair
    stage = [0.158, 0.027, 1.9, 0.006, 0.45, 0.0]
    ops = {"level0_three_iterations_per_pair": {"f32": 30_000_000, "f64": 12_000_000}, "per_pair_as_cv2": {"f32": 80_000_000, "f64": 32_000_000}}
    dom, pre, fb = b.roofline_objects(120, 1080, 1920, stage, latency_ms=1.3, mode="fast", ops=ops)
    for k in (dom, pre):
        assert k["bound"] == "hbm" and k["unit"] == "GB/s" and k["peak"] == 8000.0
        assert abs(k["frac"] - k["achieved"] / k["peak"]) < 1e-3
        assert k["traffic"] is None or k["traffic"] >= 0.9 * k["algorithmic_bytes_per_launch"]   # algorithmic <= measured
    assert dom["launches_per_step"] == 3 and abs(dom["avg_launch_ms"] - 0.15) < 1e-9
    # the first launch reads the 160-px level's flow (and resizes it itself) instead of a 320-px one: mean of the three launches
    mean_launch = (3 * per_iter - 119 * (320 * 320 - 160 * 160) * 8) // 3
    assert dom["algorithmic_bytes_per_launch"] == mean_launch < per_iter
    assert abs(dom["achieved"] - mean_launch / 0.15e-3 / 1e9) < 1.0
    assert abs(pre["achieved"] - 120 * 6_324_240 / 0.158e-3 / 1e9) < 1.0
    assert "k_fb_fast<320>" in dom["kernel"] and 0 < dom["share_of_step"] < 1
    assert dom["valu"]["f32_ops_per_level"] == 119 * 30_000_000 and 0 < dom["valu"]["frac_of_vector_peak"] < 1
    assert fb["algorithmic_bytes"] == sum(b.fused_level_bytes(120, 3, w) for w in (320, 160, 80, 40))
    edom, _, _ = b.roofline_objects(120, 1080, 1920, [0.158, 0.027, 1.6, 0.006, 0.84, 0.0], latency_ms=1.84, mode="exact", ops=ops)
    assert edom["bound"] == "valu" and "k_fb_level<320>" in edom["kernel"] and edom["launches_per_step"] == 1
    assert abs(edom["achieved"] - 3 * per_iter / 0.84e-3 / 1e9) < 1.0


def test_oracle_operation_counts_are_exact_and_stable():
    """the VALU roofline uses an operation count taken from the oracle itself (avdo_ops_*), not an estimate.
    The per-pixel figures are fixed by the code: the level count scales with the pixel count, and the warped branch of
    UpdateMatrices (47 operations) against the out-of-range one (1) is the only data dependence."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2, 320, 320), dtype=np.uint8)
    whole = O.count_farneback_ops(img[0], img[1])
    again = O.count_farneback_ops(img[0], img[1])
    assert whole == again and whole["f32"] > 5e7 and whole["f64"] > 2e7
    R = [O.poly_exp(i.astype(np.float32)) for i in img]
    zero = np.zeros((320, 320, 2), np.float32)
    lvl = O.count_level_ops(R[0], R[1], zero, 3)
    # three blur iterations: 28 double operations per pixel and iteration + 5 vertical adds, + the row / column initialisation
    px = 320 * 320
    assert lvl["f64"] == 3 * (px * (5 + 28) + 5 * 320 * 6 + 320 * 5 * 7)
    far = np.full((320, 320, 2), 1000.0, np.float32)                       # every warp leaves the image: the short branch
    assert O.count_level_ops(R[0], R[1], far, 3)["f32"] < lvl["f32"]


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` outside a launcher starts two ranks itself (a fresh torch.distributed.run child, before
    any GPU call) and rank 0 reports n_gpus = 2.  --launch-check exercises exactly that plumbing on gloo without a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    rec = json.loads(line[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2


def test_cpu_baseline_runs_one_process_per_core_started_together():
    import numpy as np
    from avd_hip import synth
    b = _bench()
    clip = synth.make_clip(4, 96, 128, seed=3, dup_every=0)
    res = b.cpu_baseline(clip, {"width": 128, "height": 96, "fps": 30.0, "duration": 2.0}, 4, 2)
    assert res["kind"] == "port" and res["cores"] == 2 and "multi_process_error" not in res
    assert res["value"] > 0 and res["single_thread_value"] > 0 and res["host_cores_available"] == os.cpu_count()


def test_cli_defaults_are_the_driver_contract():
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag, default in (("--gpus", "1"), ("--steps", "20"), ("--warmup", "3"), ("--inflight", "3"), ("--repeats", "25"),
                          ("--cpu-procs", "0")):
        assert re.search(r'add_argument\("%s", type=int, default=%s' % (re.escape(flag), default), src), flag
    # one JSON line on stdout, printed by rank 0 only
    assert src.count("print(json.dumps(out))") == 1
    # the CPU baseline and the self-launch come before the first GPU-related import
    assert src.index("sys.exit(self_launch(args))") < src.index("cpu_base = cpu_baseline(") < src.index("    import avd_hip\n")
