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
    # one JSON line on stdout, written by rank 0 only, to the descriptor saved before fd 1 was pointed at stderr
    assert src.count("os.write(real_stdout, (json.dumps(compact_line(out, args.details))") == 1
    assert src.index("os.dup2(2, 1)") < src.index("    import avd_hip\n")
    # the CPU baseline and the self-launch come before the first GPU-related import
    assert src.index("sys.exit(self_launch(args))") < src.index("cpu_base = cpu_baseline(") < src.index("    import avd_hip\n")


def _verbose_sample(b):
    """A verbose record with every key bench.py emits, long notes included (what --details FILE receives)."""
    stage = [0.158, 0.005, 0.93, 0.006, 0.435, 0.0]
    ops = {"level0_three_iterations_per_pair": {"f32": 25_200_000, "f64": 10_200_000}, "per_pair_as_cv2": {"f32": 80_000_000, "f64": 32_000_000}}
    dom, pre, fb = b.roofline_objects(120, 1080, 1920, stage, latency_ms=1.15, mode="fast", ops=ops)
    kms = {"preprocess": 0.158, "hash": 0.008, "pyramid": 0.052, "polyexp": 0.12, "level40": 0.033, "flow_up80": 0.005, "level80": 0.074,
           "flow_up160": 0.012, "level160": 0.141, "flow_up320": 0.0, "level320": 0.435, "rerun": 0.003, "stats": 0.019, "records": 0.004, "other": 0.01}
    dom.update({"traffic_from_profiles": "profiles/r04_pmc.json", "limiter": "x" * 300, "valu_issue_frac_from_profiles": 0.55,
                "ta_busy_frac_from_profiles": 0.53, "frac_of_vector_peak": 0.16, "preprocess_frac": pre["frac"], "whole_step_frac": 0.42,
                "kernels": b.kernel_table(120, 1080, 1920, kms, 0.26)})
    note = "n" * 400
    full = {"metric": "sampled frames/sec analysed (1080p30 60 s clip, 2 fps sampling)", "value": 126000.0, "unit": "frames/s", "n_gpus": 1,
            "steps": 20, "warmup": 3, "ms_per_step": 0.95, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 pixels; f32/f64 Farneback (cv2's own types)", "data": "synthetic", "fb_mode_used": "fast+rerun",
            "repeats": {"n": 25, "statistic": "median", "value_min": 1.0, "value_max": 2.0, "ms_per_step_min": 1.0, "ms_per_step_max": 2.0},
            "config": {"workload": "BASELINE.json configs[1]: 1080p30 60 s clip, 2 fps sampling, one clip per GPU per step", "frames_per_clip": 120,
                       "height": 1080, "width": 1920, "clips_per_step": 1, "clips_in_flight_per_gpu": 3, "fb_mode": "fast+rerun", "sec_per_video": 0.0147,
                       "sec_per_video_note": note, "sec_per_video_nv12": 0.0083, "sec_per_video_resident": 0.00115, "parallelism": "single GPU"},
            "roofline": dom, "roofline_preprocess": pre, "roofline_farneback_stage": fb, "stages_ms": {"note": note}, "ops": ops,
            "cpu_baseline": {"value": 453.0, "unit": "frames/s", "cores": 16, "kind": "port", "sample": "s" * 200, "single_thread_value": 30.7,
                             "process_ladder": [{"procs": 16, "frames_per_s": 453.0}] * 4, "cores_usable_how": "cgroup CPU quota 16"},
            "fb_modes": {"value_uses": "fast+rerun", "rerun_pairs": 0, "guarantee": "g" * 200, "exact_frames_per_s": 110000.0, "exact_level320_ms": 0.84,
                         "fast": {"what": note}, "exact": {"what": note, "roofline": dict(dom)},
                         "rerun_cost": {"content": note, "how": note, "unflagged_sec_per_video_resident": 0.00111, "exact_mode_sec_per_video_resident": 0.0023,
                                        "exact_mode_frames_per_s": 110000.0,
                                        "cases": [{"pairs_replaced": n, "pairs_flagged": n, "sec_per_video_resident": 0.0018, "added_ms": 0.68,
                                                   "frames_per_s": 104000.0, "note": note} for n in (1, 12, 119)]}},
            "mfma_patch_embed": {"kernel": note, "bound": "mfma", "achieved": 880.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.35, "avg_launch_ms": 0.25},
            "mfma_cnn_forward": {"kernel": note, "bound": "mfma", "achieved": 407.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.163, "forward_ms": 2.4,
                                 "hbm_traffic_bytes_per_forward": 7_610_000_000, "hbm_note": note},
            "layernorm_tokens": {"kernel": note, "bound": "hbm", "achieved": 4740.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.59, "avg_launch_ms": 0.122},
            "roofline_nv12_ingest": {"kernel": note, "bound": "hbm", "achieved": 1600.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.2, "avg_launch_ms": 0.26},
            "audio_analyzer": {"what": note, "gpu_call_ms": 0.58, "windows": 120}, "pcie_inclusive_fps": 9000.0, "pcie_inclusive_fps_one_clip_at_a_time": 8000.0,
            "short_clips_fps": {"workload": note, "value": 124000.0, "unit": "frames/s", "one_clip_per_call": 73000.0},
            "mixed_stream_fps": {"workload": note, "value": 108000.0, "unit": "frames/s", "one_clip_per_call": 67000.0},
            "result_check": {"ai_timeline_head": [0.1, 0.2, 0.3], "dup_density": 0.09, "label": "uncertain", "ai_score": 0.45, "confidence": 0.1, "reason": "r" * 80}}
    return full


def test_the_printed_line_is_compact_and_keeps_what_the_driver_reads():
    """the driver's record cut a 15-KB line.  The printed line stays below 8 KB whatever the verbose record
    holds, keeps the contract keys, `roofline` with the per-kernel table and PMC-derived fractions named *_from_profiles, and
    `cpu_baseline`; everything else goes to --details FILE."""
    import json
    b = _bench()
    full = _verbose_sample(b)
    line = b.compact_line(full, "gpurun_out/details.json")
    text = json.dumps(line)
    assert len(text) < 8192, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert r["traffic_from_profiles"] and r["valu_issue_frac_from_profiles"] == 0.55 and r["ta_busy_frac_from_profiles"] == 0.53
    names = [k["name"] for k in r["kernels"]]
    for want in ("preprocess", "pyramid", "polyexp", "level40", "level80", "level160", "level320", "stats"):
        assert want in names, want
    assert any(nm.startswith("nv12") for nm in names)
    for k in r["kernels"]:
        assert set(k) == {"name", "bound", "us", "launches", "frac"} and k["us"] > 0
    lvl = next(k for k in r["kernels"] if k["name"] == "level320")
    assert abs(lvl["frac"] - r["frac"]) < 0.01                       # the dominant kernel's row agrees with the roofline object
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 16 and "process_ladder" not in line["cpu_baseline"]
    assert line["fb_modes"]["rerun_pairs"] == 0 and line["config"]["workload"].startswith("BASELINE.json configs[1]")
    rc = line["fb_modes"]["rerun_cost"]                              # what a flagged pair costs is in the driver's line
    assert [c["pairs_flagged"] for c in rc["cases"]] == [1, 12, 119] and all(set(c) == {"pairs_flagged", "sec_per_video_resident", "added_ms", "frames_per_s"}
                                                                              for c in rc["cases"])
    assert rc["unflagged_sec_per_video_resident"] == 0.00111 and rc["exact_mode_frames_per_s"] == 110000.0 and "content" not in rc
    assert line["extensions"]["patch_embed"]["frac"] == 0.35 and "kernel" not in line["extensions"]["patch_embed"]
    # N > 1: no cpu_baseline (rank 0 at N = 1 only), the roofline stays
    full["cpu_baseline"] = None
    full["n_gpus"] = 8
    l8 = b.compact_line(full)
    assert l8["cpu_baseline"] is None and l8["roofline"]["kernels"] and "details" not in l8
