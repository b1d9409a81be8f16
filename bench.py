#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-frame video-analysis hot path on MI355X.

A "step" = one pass of the hot path (fused preprocess -> aHash/Hamming -> Farneback -> flow
statistics -> per-frame records -> scalar timeline tail -> fusion) over ONE synthetic clip per GPU:
BASELINE.json configs[1], a 1080p30 60 s clip sampled at 2 fps = 120 BGR frames
(uint8[120,1080,1920,3], 746 MB) already resident in HBM when the timed region starts.
With N > 1 ranks every rank analyses its own clip (whole clips per GPU) and one
RCCL all-gather of the 32-byte per-frame records reassembles all timelines: weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself -- a fresh
`python -m torch.distributed.run` child, before this process has touched the GPU -- and relays rank 0's line.)

Prints ONE JSON line on rank 0.  What is measured, and how:
  value / ms_per_step   W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize, max over ranks;
                        the K-step region is repeated --repeats times in the same process and the MEDIAN repeat is
                        reported (min / max beside it: boxes and DVFS move a single 25 ms region by several percent).
                        --inflight clips are in flight per GPU, each on its own avd context AND its own copy of the
                        input in HBM (a service never analyses the same buffer three times at once).
  roofline              the kernel with the largest share of the step: the Farneback level kernel at 320x320 (fb_mode
                        fast, the default and what `value` uses: one launch per blur iteration), timed with HIP events
                        around its three launches on the library's own stream while clips run ALONE (an exclusive pass
                        before the timed region; with clips in flight an event-to-event time includes whatever shares
                        the GPU).  achieved = ALGORITHMIC bytes per launch / average launch duration; traffic = HBM
                        bytes per launch from the committed rocprofv3 --pmc passes (profiles/).  `fb_modes` reports the
                        exact level kernel (fb_mode exact: bit-identical to the oracle, one workgroup per pair) beside it.
  short_clips_fps /     BASELINE.json configs[0]-sized clips (20 sampled 720p frames) and a configs[4] mixed-resolution
  mixed_stream_fps      stream (720p / 1080p / 4K), handed over as BATCHES (avd_analyze_batch: one Farneback launch
                        sequence over all pairs of a batch) and, for comparison, one clip per call.
  roofline_preprocess   the HBM-bound fused full-resolution kernel (north_star's ">= 60 % of HBM peak" target).
  config.sec_per_video  BASELINE.json's second metric: latency of one clip from decoded frames in PINNED HOST memory to
                        the final result (PCIe-inclusive); the HBM-resident latency is beside it.
  pcie_inclusive_fps    whole-job rate when every clip is handed over as a pinned host buffer (never `value`).
  cpu_baseline          the CPU oracle (kind "port"), clip-parallel on ALL host cores (one process per core, each the whole
                        bounded sample, started together), with the one-thread figure beside it; measured BEFORE this
                        process makes its first GPU call.
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
FP32_VALU_PEAK_TF = 157.3
PMC_FILE = next((f for f in (os.path.join(ROOT, "profiles", "r05_pmc.json"), os.path.join(ROOT, "profiles", "r04_pmc.json"), os.path.join(ROOT, "profiles", "r03_pmc.json"))
                 if os.path.exists(f)), os.path.join(ROOT, "profiles", "r03_pmc.json"))      # rocprofv3 --pmc passes of the fast mode
PMC_FILE_EXACT = os.path.join(ROOT, "profiles", "r02_pmc.json")
FP64_VALU_PEAK_TF = 78.6


def preprocess_bytes_per_frame(h, w):
    """one read of the BGR frame + the small outputs (320x320 gray, 1024 hash bits, two int64 moments)."""
    return h * w * 3 + 320 * 320 + 1024 + 16


def fused_level_bytes(n_frames, iterations=3, w=320):
    """Algorithmic bytes of ONE launch of the fused Farneback level kernel: per iteration every
    frame's polynomial expansion is read once (5 floats per pixel; a frame is R0 of one pair and R1 of the next)
    and every pair's flow is read and written once (2 floats each way).  The on-chip working set of a pair (4 MB of
    R + 0.8 MB of flow) exceeds LDS, so each iteration streams it again."""
    pairs = max(n_frames - 1, 0)
    per_iter = n_frames * w * w * 20 + pairs * w * w * 16
    return iterations * per_iter


def load_pmc(path=None):
    path = path or PMC_FILE
    if os.path.exists(path):
        with open(path) as fh:
            return json.load(fh)
    return {}


def oracle_op_counts(small_pair):
    """Exact arithmetic-operation counts from the instrumented oracle (oracle/avd_oracle.c, avdo_ops_*): (a) everything
    cv2 executes for one pair (both frames' pyramids and polynomial expansions included), (b) the blur iterations of
    the 320 x 320 level of one pair = what the three launches of the level-0 kernel compute."""
    from oracle import oracle as O
    a, b = small_pair
    whole = O.count_farneback_ops(a, b)
    R0, R1 = O.poly_exp(a.astype(np.float32)), O.poly_exp(b.astype(np.float32))
    flow = np.zeros((320, 320, 2), np.float32)
    flow[..., 0] = 1.5
    flow[..., 1] = -0.75                                       # a non-zero flow: every pixel takes the warped branch
    lvl0 = O.count_level_ops(R0, R1, flow, 3)
    return {"per_pair_as_cv2": whole, "level0_three_iterations_per_pair": lvl0,
            "counted_in": "oracle/avd_oracle.c (OPS32 / OPS64 at the statements that execute; an fma counts 2)"}


def roofline_objects(n, h, w, stage, latency_ms, mode="fast", ops=None):
    """stage: avd_stage_ms of clips run alone: [preprocess, hash.., farneback+stats, copy-out, level 0 (all iterations), -].
    mode "fast": three launches of k_fb_fast<320> (one per iteration); "exact": one launch of k_fb_level<320>."""
    exact = mode == "exact"
    pmc = load_pmc(PMC_FILE_EXACT if exact else PMC_FILE) if (n, h, w) == (120, 1080, 1920) else {}
    pairs = max(n - 1, 0)
    lvl_ms, pre_ms, stage_ms = float(stage[4]), float(stage[0]), float(stage[2])
    launches = 1 if exact else 3
    fb_ms = lvl_ms / launches                                   # average duration of ONE launch
    alg = fused_level_bytes(n, 3 if exact else 1)
    if not exact:
        # the first of the three launches reads the 160-px level's flow (a quarter of the bytes) and resizes it itself
        # (no k_flow_up<320> launch): mean over the three launches, which is what avg_launch_ms and traffic are
        alg = (3 * alg - pairs * (320 * 320 - 160 * 160) * 8) // 3
    ach = alg / (fb_ms * 1e-3) / 1e9 if fb_ms > 0 else 0.0
    lvl_ops = (ops or {}).get("level0_three_iterations_per_pair")
    valu = None
    if lvl_ops:
        f32, f64 = pairs * lvl_ops["f32"], pairs * lvl_ops["f64"]
        t = lvl_ms * 1e-3
        valu = {"f32_ops_per_level": f32, "f64_ops_per_level": f64,
                "f32_tops": round(f32 / t / 1e12, 2) if t > 0 else 0.0, "f64_tops": round(f64 / t / 1e12, 2) if t > 0 else 0.0,
                # time the vector units need at their peak rates (an add or a multiply is half an fma) over the time taken
                "frac_of_vector_peak": round((f32 / (FP32_VALU_PEAK_TF * 0.5e12) + f64 / (FP64_VALU_PEAK_TF * 0.5e12)) / t, 4) if t > 0 else 0.0,
                "fp32_vector_peak_tflops": FP32_VALU_PEAK_TF, "fp64_vector_peak_tflops": FP64_VALU_PEAK_TF,
                "note": "exact operation counts of the oracle for the same level (no fma contraction: cv2's arithmetic is mul + add); "
                        "the peaks count an fma as two, so an add / mul stream can reach half of them"}
    if exact:
        dominant = {
            "kernel": "k_fb_level<320> (fb_mode exact: normal equations + vertical / horizontal double RUNNING sums + 2x2 solve, 3 iterations in "
                      "one launch, one workgroup per pair, bit-identical to the oracle)",
            "bound": "valu", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "bound_note": "VALU issue of the vertical waves + the dependent double-add chain of the horizontal scan on 119 of 256 CUs "
                          "(profiles/r02_experiments.md); the HBM figures are reported for comparison with the fast kernel"}
    else:
        dominant = {
            "kernel": "k_fb_fast<320> (fb_mode fast: one blur iteration per launch, a pair spread over two column strips = 238 workgroups of "
                      "12 waves; literal vertical running sums, horizontal 15-column windows summed directly in double, 2x2 solve)",
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "bound_note": "per iteration every frame's polynomial expansion (20 B/px) and every pair's flow in + out (16 B/px) cross HBM: "
                          "441 MB per launch at 119 pairs (the first launch reads the coarser level's flow instead: 368 MB; mean 417 MB); "
                          "PMC traffic ~1.1x that (halo columns, partial lines, |flow| written by the last launch).  Below the achievable "
                          "~6.3 TB/s the remaining limiter is the texture-addresser / L1 path of the bilinear gathers and the per-step "
                          "workgroup barrier (profiles/r03_experiments.md)"}
    dominant.update({
        "traffic": pmc.get("k_fb_level<320>" if exact else "k_fb_fast<320>", {}).get("hbm_bytes"),
        "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(fb_ms, 4), "launches_per_step": launches,
        "share_of_step": round(lvl_ms / latency_ms, 4) if latency_ms > 0 else 0.0,
        "valu": valu,
        "timed": "HIP events around the level's launches on the library's stream, clips run alone before the timed region"})
    alg_pre = preprocess_bytes_per_frame(h, w) * n
    ach_pre = alg_pre / (pre_ms * 1e-3) / 1e9 if pre_ms > 0 else 0.0
    pre = {"kernel": "k_preprocess_vec (fused BGR->gray, INTER_AREA partials, INTER_LINEAR 320x320, Laplacian moments)",
           "bound": "hbm", "achieved": round(ach_pre, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach_pre / HBM_PEAK_GBS, 4), "traffic": pmc.get("k_preprocess_vec", {}).get("hbm_bytes"),
           "algorithmic_bytes_per_launch": alg_pre, "avg_launch_ms": round(pre_ms, 4),
           "share_of_step": round(pre_ms / latency_ms, 4) if latency_ms > 0 else 0.0}
    px = 320 * 320 + 160 * 160 + 80 * 80 + 40 * 40
    fb_alg = sum(fused_level_bytes(n, 3, ww) for ww in (320, 160, 80, 40))
    t = stage_ms * 1e-3
    whole = (ops or {}).get("per_pair_as_cv2")
    fb = {"stage": "Farneback (pyramid, polynomial expansion, level kernels of 4 scales) + flow statistics", "avg_ms": round(stage_ms, 4),
          "algorithmic_bytes": fb_alg, "achieved": round(fb_alg / t / 1e9, 1) if t > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": round(fb_alg / t / 1e9 / HBM_PEAK_GBS, 4) if t > 0 else 0.0,
          "ops_per_pair_as_cv2": whole,
          "tops_cv2_equivalent": round(pairs * (whole["f32"] + whole["f64"]) / t / 1e12, 2) if (whole and t > 0) else None,
          "pixels_per_pair_all_levels": px,
          "traffic": pmc.get("farneback_stage", {}).get("hbm_bytes")}
    return dominant, pre, fb


def kernel_table(n, h, w, kms, nv12_ms=None):
    """roofline.kernels: every kernel (group) of one clip with its ALGORITHMIC bytes priced against the HBM peak and its
    measured duration (avd_kernel_ms: HIP events on the library's stream, clips run alone, mean over the exclusive pass).
    A level's row is the sum of its launches.  `bound` says what the kernel is limited by, which for everything but the
    full-resolution pass is NOT the HBM bandwidth (the fraction shows how far from it): the level kernels by VALU issue + the
    texture-addresser path of the bilinear gathers (320 px) or by the latency of 16 .. 44 barrier steps (smaller levels),
    the polynomial expansion by VALU issue (cv2's double accumulators)."""
    np_ = max(n - 1, 0)
    px = {320: 320 * 320, 160: 160 * 160, 80: 80 * 80, 40: 40 * 40}
    allpx = sum(px.values())
    lvl = lambda ww, first_quarter: 3 * (n * px[ww] * 20 + np_ * px[ww] * 16) - (np_ * (px[ww] - px[ww] // 4) * 8 if first_quarter else 0)  # noqa: E731
    rows = [
        ("preprocess", "hbm", n * preprocess_bytes_per_frame(h, w), kms.get("preprocess"), 1),
        ("hash", "latency", n * (h * 32 * 4 + 2048), kms.get("hash"), 1),
        ("pyramid", "latency+lds", n * (px[320] + 4 * allpx), kms.get("pyramid"), 1),
        ("polyexp", "valu", n * allpx * 24, kms.get("polyexp"), 1),
        ("level40", "latency", lvl(40, False), kms.get("level40"), 3),
        ("flow_up80", "latency", np_ * (px[40] + px[80]) * 8, kms.get("flow_up80"), 1),
        ("level80", "latency", lvl(80, False), kms.get("level80"), 3),
        ("flow_up160", "latency", np_ * (px[80] + px[160]) * 8, kms.get("flow_up160"), 1),
        ("level160", "latency", lvl(160, False), kms.get("level160"), 3),
        ("flow_up320", "hbm", np_ * (px[160] + px[320]) * 8, kms.get("flow_up320"), 1),
        ("level320", "valu+ta", lvl(320, not kms.get("flow_up320")), kms.get("level320"), 3),
        ("rerun", "launch", 0, kms.get("rerun"), 1),
        ("stats", "latency", np_ * px[320] * 4, kms.get("stats"), 1),
        ("records", "launch", n * 2080, kms.get("records"), 1),
    ]
    if nv12_ms:
        rows.append(("nv12_ingest (instead of preprocess)", "valu", n * (h * w * 3 // 2 + 320 * 320 + 1024 + 16), nv12_ms, 1))
    out = []
    for name, bound, alg, ms, launches in rows:
        if not ms or ms <= 0:
            continue
        out.append({"name": name, "bound": bound, "us": round(ms * 1e3, 1), "launches": launches,
                    "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if alg else None})
    return out


def compact_line(full, details_path=None):
    """The ONE JSON line rank 0 prints: the driver's contract keys, the roofline of the dominant kernel with the per-kernel
    table, cpu_baseline, and one figure per extra; everything else (notes, sub-objects) goes to --details FILE."""
    pick = lambda d, keys: {k: d[k] for k in keys if d is not None and k in d}  # noqa: E731
    line = pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    line["config"] = pick(full["config"], ("workload", "frames_per_clip", "height", "width", "clips_per_step", "clips_in_flight_per_gpu",
                                           "fb_mode", "fb_wide160", "sec_per_video", "sec_per_video_resident", "sec_per_video_nv12", "parallelism", "rccl_ranks_seen",
                                           "collective_backend"))
    line["repeats"] = pick(full.get("repeats"), ("n", "statistic", "value_min", "value_max"))
    r = full["roofline"]
    line["roofline"] = pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "hbm", "traffic", "traffic_from_profiles",
                                "algorithmic_bytes_per_launch", "avg_launch_ms", "launches_per_step", "share_of_step", "limiter",
                                "frac_of_vector_peak", "valu_issue_frac_from_profiles", "ta_busy_frac_from_profiles", "kernels",
                                "whole_step_frac", "preprocess_frac", "timed"))
    if full.get("cpu_baseline"):
        line["cpu_baseline"] = pick(full["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "single_thread_value", "cores_usable_how"))
    else:
        line["cpu_baseline"] = None
    if full.get("fb_modes"):
        line["fb_modes"] = pick(full["fb_modes"], ("value_uses", "rerun_pairs", "guarantee", "exact_frames_per_s", "fast_without_rerun_frames_per_s",
                                                   "exact_level320_ms"))
        rc = full["fb_modes"].get("rerun_cost")
        if rc:
            line["fb_modes"]["rerun_cost"] = {"unflagged_sec_per_video_resident": rc.get("unflagged_sec_per_video_resident"),
                                              "exact_mode_sec_per_video_resident": rc.get("exact_mode_sec_per_video_resident"),
                                              "exact_mode_frames_per_s": rc.get("exact_mode_frames_per_s"),
                                              "cases": [pick(c, ("pairs_flagged", "sec_per_video_resident", "added_ms", "frames_per_s")) for c in rc["cases"]]}
    ext = {}
    for key, short in (("mfma_patch_embed", "patch_embed"), ("mfma_cnn_forward", "cnn_forward"), ("layernorm_tokens", "layernorm"),
                       ("roofline_nv12_ingest", "nv12_ingest")):
        if full.get(key):
            ext[short] = pick(full[key], ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "forward_ms", "hbm_traffic_bytes_per_forward"))
            if key == "mfma_cnn_forward" and full[key].get("four_clips_per_pass"):
                ext[short]["frac_four_clips_per_pass"] = full[key]["four_clips_per_pass"]["frac"]
    if full.get("audio_analyzer"):
        ext["audio"] = pick(full["audio_analyzer"], ("gpu_call_ms", "windows"))
    if ext:
        line["extensions"] = ext
    for key in ("pcie_inclusive_fps", "pcie_inclusive_fps_one_clip_at_a_time"):
        if key in full:
            line[key] = full[key]
    for key in ("short_clips_fps", "mixed_stream_fps"):
        if full.get(key):
            line[key] = pick(full[key], ("value", "one_clip_per_call"))
    line["result_check"] = full.get("result_check")
    if details_path:
        line["details"] = details_path
    return line


_CPU_CHILD = r"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
O.lib()
clip = np.load(sys.argv[2], mmap_mode="r")            # shared page cache: no per-process copy of the sample
meta = json.loads(sys.argv[3])
print("ready", flush=True)
while not os.path.exists(sys.argv[4]):                # all processes start together
    time.sleep(0.002)
t0 = time.perf_counter()
O.analyze_sampled_frames(clip, meta)
print(time.perf_counter() - t0, flush=True)
"""


def usable_cores():
    """Cores this process may actually use: the scheduler affinity, capped by the container's CPU quota (cgroup v2
    cpu.max / v1 cfs quota) -- os.cpu_count() reports the HOST's logical CPUs, which a GPU box's container only gets a
    share of (starting one process per host CPU there only oversubscribes the share).  -> (cores, how it was found)"""
    import math
    try:
        cores, how = len(os.sched_getaffinity(0)), "scheduler affinity"
    except (AttributeError, OSError):
        cores, how = os.cpu_count() or 1, "os.cpu_count()"
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
                if q > 0 and per > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < cores:
        cores, how = max(1, int(math.floor(quota + 1e-9))), f"cgroup CPU quota {quota:g}"
    return cores, how


def cpu_baseline(clip, meta, max_frames, procs):
    """The CPU oracle (a port of the reference's cv2/numpy arithmetic; cv2's Farneback is single-threaded) timed on a
    bounded sample of the same clip: (a) one thread, one clip -- the reference's per-request path; (b) one process per
    host core, every one the whole sample, started together once all of them have loaded -- how a CPU box is loaded for
    throughput (clip-parallel).  Fresh interpreters that only import numpy and the oracle; called before this process
    makes its first GPU call."""
    import subprocess
    import tempfile
    from oracle import oracle as O
    O.lib()
    sample = np.ascontiguousarray(clip[:max_frames])
    meta = dict(meta, duration=len(sample) / 2.0)
    t0 = time.perf_counter()
    O.analyze_sampled_frames(sample, meta)
    dt1 = time.perf_counter() - t0
    single = len(sample) / dt1
    out = {"value": round(single, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"first {len(sample)} sampled frames of the same clip, oracle/avd_oracle.c single thread, {dt1:.1f} s wall",
           "single_thread_value": round(single, 3), "host_cores_available": os.cpu_count()}
    if procs > 1:
        kids = []
        try:
            shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            with tempfile.TemporaryDirectory(dir=shm) as td:
                path, go = os.path.join(td, "sample.npy"), os.path.join(td, "go")
                np.save(path, sample)
                kids = [subprocess.Popen([sys.executable, "-c", _CPU_CHILD, ROOT, path, json.dumps(meta), go],
                                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(procs)]
                for k in kids:
                    if k.stdout.readline().strip() != "ready":
                        raise RuntimeError("a baseline process did not start")
                t0 = time.perf_counter()
                open(go, "w").close()
                times = [float(k.stdout.readline()) for k in kids]
                dtp = time.perf_counter() - t0
                for k in kids:
                    k.wait(timeout=60)
            out = {"value": round(procs * len(sample) / dtp, 2), "unit": "frames/s", "cores": procs, "kind": "port",
                   "sample": f"{procs} processes (one per host core) x the first {len(sample)} sampled frames of the same clip, started "
                             f"together (clip-parallel, oracle/avd_oracle.c, one thread each): {dtp:.1f} s wall, slowest process {max(times):.1f} s",
                   "single_thread_value": round(single, 3), "host_cores_available": os.cpu_count()}
        except Exception as exc:                       # the single-thread figure stands; say why the other is missing
            out["multi_process_error"] = repr(exc)
            for k in kids:
                if k.poll() is None:
                    k.kill()
    return out


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as a fresh child
    (`python -m torch.distributed.run`, one rank per GPU) and relay its output.  This process has not touched the GPU
    (nothing GPU-related is imported before this point), so nothing is re-executed from an initialised process."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.run(cmd, env=env).returncode


def launch_check(args, rank, world):
    """--launch-check: the multi-rank plumbing without a GPU (CPU test of the launch contract): the ranks form the gloo
    group, agree on the world size with one all-reduce, rank 0 prints a JSON line.  Not a measurement."""
    import torch
    import torch.distributed as tdist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tdist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        tdist.all_reduce(t)
        seen = int(t.item())
        tdist.destroy_process_group()
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": seen, "gpus_flag": args.gpus}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=25, help="the K-step timed region is repeated this many times in-process; "
                                                            "the median repeat is reported (1 = a single region)")
    ap.add_argument("--frames", type=int, default=120, help="sampled frames per clip (120 = 60 s at 2 fps)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--cpu-frames", type=int, default=120, help="frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=0,
                    help="processes of the clip-parallel CPU baseline: 0 (default) = one per host core; 1 = single thread only")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact-mode pass and the short-clip / mixed-stream batches")
    ap.add_argument("--launch-check", action="store_true", help="multi-rank plumbing check without a GPU (gloo); not a measurement")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--details", default=None, metavar="FILE",
                    help="also write the verbose record (every sub-object and note behind the printed line) to FILE")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurements")
    ap.add_argument("--no-vit", action="store_true", help="skip the extensions reported apart (ViT patch-embed GEMM, CNN forward, audio analyzer)")
    ap.add_argument("--pcie", action="store_true", help="accepted for compatibility (the PCIe-inclusive figures are on by default)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="clips in flight per GPU, each on its own avd context / stream / workspace / input copy.  3 (default) = "
                         "how a service drives the GPU: the next clips are submitted before the previous one is drained, so the "
                         "per-pair workgroups of one clip's Farneback levels (119 of 256 CUs), the records all-gather, the host "
                         "tail and the launch gaps of one clip overlap with the kernels of another; 1 = every step is submitted "
                         "and drained alone.  All K steps complete inside the timed region either way.")
    ap.add_argument("--collective", default="torch", choices=["torch", "native"],
                    help="record all-gather for N>1: torch.distributed (default) or the library's own avd_allgather_records "
                         "(RCCL bound at run time inside libavd_hip.so; the unique id travels through torch's process group)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))                 # before anything GPU-related is imported

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if args.launch_check:
        sys.exit(launch_check(args, rank, world))

    # stdout carries exactly ONE line, rank 0's JSON record: whatever a library (RCCL, gloo, a compiler run by build()) writes to
    # file descriptor 1 in any rank goes to stderr instead; the record is written to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    n, h, w = args.frames, args.height, args.width
    meta = {"width": w, "height": h, "fps": 30.0, "duration": n / 2.0}
    from avd_hip import synth                       # numpy only
    clip = synth.make_clip(n, h, w, seed=args.seed + rank)          # synthetic, SURVEY.md 8(d) recipe

    # CPU legs FIRST, before this process makes any GPU call: the baseline (its children are fresh interpreters) and the
    # oracle's exact operation counts
    cpu_base = ops = None
    if rank == 0 and world == 1:
        from oracle import oracle as O
        two = np.stack([O.resize_linear(O.bgr2gray(f), 320, 320) for f in clip[:2]])
        ops = oracle_op_counts(two)
        if args.cpu_frames > 0:
            cores, how = usable_cores()
            nf = min(args.cpu_frames, n)
            if args.cpu_procs > 0:
                cpu_base = cpu_baseline(clip, meta, nf, max(1, min(args.cpu_procs, os.cpu_count() or 1)))
            else:
                # one process per usable core.  Where nothing on paper restricts a many-CPU host (a container may still only
                # get a share of it), climb 16, 32, 64 ... and stop when doubling the processes no longer pays: the best
                # figure and ITS process count are reported
                p = min(cores, 16)
                cpu_base = cpu_baseline(clip, meta, nf, p)
                ladder = [{"procs": p, "frames_per_s": cpu_base["value"]}]
                while p < cores:
                    p = min(cores, 2 * p)
                    nxt = cpu_baseline(clip, meta, nf, p)
                    ladder.append({"procs": p, "frames_per_s": nxt["value"]})
                    if "multi_process_error" in nxt or nxt["value"] < 1.3 * cpu_base["value"]:
                        if nxt["value"] > cpu_base["value"] and "multi_process_error" not in nxt:
                            cpu_base = nxt
                        break
                    cpu_base = nxt
                cpu_base["process_ladder"] = ladder
            cpu_base["cores_usable"] = cores
            cpu_base["cores_usable_how"] = how

    import avd_hip
    from avd_hip import dist as avd_dist
    from avd_hip.timeline import records_to_result
    from avd_hip.pipeline import audio_unavailable
    from app.analyzers import fusion, heuristics_v2
    if local_rank == 0:
        avd_hip.build()                             # no-op when the in-tree .so is up to date
    else:                                           # other ranks wait for rank 0's build instead of racing it
        from avd_hip import _lib as _avd_lib
        for _ in range(600):
            if os.path.exists(_avd_lib.SO_PATH):
                break
            time.sleep(0.2)
    avd_hip.load()                                  # fail loudly before anything else if the .so is missing
    import torch
    dev_index = int(os.environ.get("AVD_BENCH_DEVICE", local_rank))      # rehearsal: several ranks on one GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1
    if use_dist:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            tdist.init_process_group("nccl", device_id=dev)
        else:
            tdist.init_process_group("gloo")
    gather_dev = dev if args.backend == "nccl" else None

    m = max(1, args.inflight)
    host = torch.from_numpy(clip)
    frames = [host.to(dev) for _ in range(m)]                        # one resident copy per in-flight slot
    ctxs = [avd_hip.Context(dev_index) for _ in range(m)]
    prof = {"on": False}

    def set_prof(on):
        """Per-stage HIP events (avd_set_profiling) cost a few event packets per clip: they are on for the passes that report
        stage times and OFF wherever a throughput or a latency is measured (value, ms_per_step, sec_per_video*)."""
        for c in ctxs:
            c.set_profiling(bool(on))
        prof["on"] = bool(on)

    set_prof(True)
    recs = [np.zeros(n, avd_hip.RECORD_DTYPE) for _ in range(m)]
    if use_dist and args.collective == "native":
        for c in ctxs:                                      # one communicator per context (each has its own stream)
            box = [avd_hip.Context.comm_unique_id() if rank == 0 else None]
            tdist.broadcast_object_list(box, src=0)
            c.comm_init(rank, world, box[0])
    hints = heuristics_v2.compute_hints({**meta, "bit_rate": 8_000_000}, "")
    stage = np.zeros(6)
    pending = []

    def submit(i, src=None):
        j = i % m
        keep = ctxs[j].analyze_frames_async(frames[j] if src is None else src, recs[j])
        pending.append((j, keep))

    def retire():
        j, _keep = pending.pop(0)
        if use_dist and args.collective == "native":
            # straight from the device records, enqueued behind the clip on its stream; drains the clip as well
            allrec = ctxs[j].allgather_last_records(n)
            rec = recs[j]
            if prof["on"]:
                stage[:] += np.array(ctxs[j].stage_ms())
        else:
            ctxs[j].synchronize()
            if prof["on"]:
                stage[:] += np.array(ctxs[j].stage_ms())
            rec = recs[j]
            allrec = avd_dist.gather_fixed(rec, device=gather_dev) if use_dist else rec
        # scalar tail (video.py:54-83) + fusion (fusion.py:16) for this rank's clip; other clips' records are local too
        video = records_to_result(allrec[rank * n:(rank + 1) * n], h * w, w, h, meta["fps"], meta["duration"])
        fused = fusion.fuse(audio_unavailable("", meta), video, hints)
        return video, fused

    def run(steps, src=None):
        out = None
        for i in range(steps):
            if len(pending) == m:
                out = retire()
            submit(i, src)
        while pending:
            out = retire()
        return out

    def barrier():
        if use_dist:
            tdist.barrier()
        torch.cuda.synchronize()

    # EXCLUSIVE pass, before the timed region: clips submitted and drained alone.  It gives (a) the resident latency of
    # one clip and (b) per-kernel durations from HIP events on the kernels' own stream that are not mixed with another
    # clip's kernels.
    for _ in range(args.warmup):
        submit(0)
        retire()
    lat, excl = [], np.zeros(6)
    n_excl = 10
    kms = {}                                # per-kernel device times of a clip run alone (avd_kernel_ms), mean of the pass
    for _ in range(n_excl):
        stage[:] = 0
        submit(0)
        retire()
        excl += stage
        for kk, vv in ctxs[0].kernel_ms().items():
            kms[kk] = kms.get(kk, 0.0) + vv / n_excl
    excl /= n_excl
    rerun_pairs = ctxs[0].get_option("rerun_pairs")
    fb_mode_used = {0: "exact", 1: "fast+rerun" if ctxs[0].get_option("fb_rerun") else "fast (re-run off)"}[ctxs[0].get_option("fb_mode")]
    set_prof(False)                   # from here on nothing but the work itself is on the streams
    submit(0)
    retire()
    for _ in range(n_excl):
        t1 = time.perf_counter()
        submit(0)
        retire()
        lat.append(time.perf_counter() - t1)
    latency_ms = statistics.median(lat) * 1e3

    # decoded frames in PINNED HOST memory -> result: BASELINE.json's "end-to-end sec/video" (decode excluded)
    host_lat_ms = pcie_fps = pcie_fps_inflight = None
    if not args.no_pcie:
        pinned = host.pin_memory()
        submit(0, pinned)
        retire()
        hl = []
        for _ in range(5):
            t1 = time.perf_counter()
            submit(0, pinned)
            retire()
            hl.append(time.perf_counter() - t1)
        host_lat_ms = statistics.median(hl) * 1e3
        pcie_fps = n / statistics.median(hl)
        if m > 1:
            reps = 4 * m
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run(reps, pinned)
            pcie_fps_inflight = reps * n / (time.perf_counter() - t1)

    # NV12 ingest: the same clip as decoder surfaces (1.5 B per pixel), reported apart -- never `value`
    nv12 = None
    if not args.no_pcie and (h % 2 == 0 and w % 2 == 0):
        k = min(n, 8)
        ys, cs = synth.bgr_to_nv12(clip[:k])                  # a few distinct surfaces, cycled: the kernel time does not depend on content
        idx = np.arange(n) % k
        hy, hc = torch.from_numpy(ys[idx]).pin_memory(), torch.from_numpy(cs[idx]).pin_memory()
        dy, dc = hy.to(dev), hc.to(dev)
        c0 = ctxs[0]
        rec_nv = c0.analyze_frames_nv12(dy, dc)
        pre, pre_stage = [], []
        c0.set_profiling(True)
        for _ in range(10):
            c0.analyze_frames_nv12(dy, dc)
            pre.append(c0.kernel_ms()["preprocess"])            # the ingest kernel's own region (as the BGR kernel's row of roofline.kernels is measured)
            pre_stage.append(c0.stage_ms()[0])                   # rounds 3 / 4 reported this: the stage, which also holds k_hash and the stage events
        c0.set_profiling(False)
        t1 = time.perf_counter()
        for _ in range(3):
            c0.analyze_frames_nv12(hy, hc)
        nv12 = {"pre_ms": statistics.median(pre), "pre_stage_ms": statistics.median(pre_stage), "host_fps": 3 * n / (time.perf_counter() - t1),
                "flow_mean_head": [round(float(v), 6) for v in rec_nv["flow_mean"][1:3]]}
        nl = []
        for _ in range(5):                                     # latency of one clip from pinned NV12 surfaces to its records + host tail
            t1 = time.perf_counter()
            rnv = c0.analyze_frames_nv12(hy, hc)
            records_to_result(rnv, h * w, w, h, meta["fps"], meta["duration"])
            nl.append(time.perf_counter() - t1)
        nv12["host_latency_ms"] = statistics.median(nl) * 1e3
        if m > 1:                                              # pinned surfaces -> records with m clips in flight (copy of one under compute of another)
            nrec = [np.zeros(n, avd_hip.RECORD_DTYPE) for _ in range(m)]
            q, reps = [], 4 * m
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(reps):
                if len(q) == m:
                    ctxs[q.pop(0)[0]].synchronize()
                j = i % m
                q.append((j, ctxs[j].analyze_frames_nv12_async(hy, hc, nrec[j])))
            while q:
                ctxs[q.pop(0)[0]].synchronize()
            nv12["host_fps_inflight"] = reps * n / (time.perf_counter() - t1)
        del dy, dc

    # ViT-B/16 patch-embed MFMA stage (a build-defined extension, NOT part of `value`): the GEMM
    # [M = frames x 196, 768] x [768, 768] on 8 x the clip's frames (a shape that fills the chip several times over)
    vit = None
    if not args.no_vit and world == 1:
        rngv = np.random.default_rng(7)
        ctxs[0].vit_set_weights((rngv.standard_normal((768, 768)) * 0.02).astype(np.float32),
                                (rngv.standard_normal(768) * 0.1).astype(np.float32))
        big = frames[0].repeat(8, 1, 1, 1) if n * 8 * h * w * 3 < 16e9 else frames[0]
        tok = torch.empty((big.shape[0], 196, 768), dtype=torch.bfloat16, device=dev)
        ctxs[0].vit_patch_embed(big, timing_reps=3, out=tok, bf16=True)
        t1 = time.perf_counter()
        _, gemm_ms = ctxs[0].vit_patch_embed(big, timing_reps=20, out=tok, bf16=True)
        whole = (time.perf_counter() - t1) * 1e3 - 20 * gemm_ms
        tok32 = torch.empty((big.shape[0], 196, 768), dtype=torch.float32, device=dev)
        ctxs[0].vit_patch_embed(big, timing_reps=3, out=tok32)
        _, gemm_ms32 = ctxs[0].vit_patch_embed(big, timing_reps=20, out=tok32)
        vit = {"frames": int(big.shape[0]), "gemm_ms": gemm_ms, "gemm_ms_f32_tokens": gemm_ms32, "whole_call_ms": whole,
               "checksum": float(tok32[0, 0, :8].sum().item())}
        del big, tok, tok32

    # CNN extension (build-defined, NOT part of `value`): ResNet-50-style forward of the clip's frames
    cnn = None
    if not args.no_vit and world == 1:
        from avd_hip import cnn as host_cnn
        wts, bss = host_cnn.seeded_parameters(0)
        ctxs[0].cnn_set_weights(wts, bss)
        ctxs[0].cnn_forward(frames[0], timing_reps=1)
        logits, fwd_ms = ctxs[0].cnn_forward(frames[0], timing_reps=5)
        cnn = {"frames": int(frames[0].shape[0]), "forward_ms": fwd_ms, "macs_per_frame": host_cnn.macs_per_frame(),
               "top1_head": [int(v) for v in logits[:3].argmax(axis=1)], "fused": int(ctxs[0].get_option("cnn_fuse"))}
        # the same network layer by layer (A/B of the fused blocks), and four clips per forward pass (the 14x14 / 7x7 stages have
        # too few pixels per 120 frames to fill 256 CUs with large tiles)
        ctxs[0].set_option("cnn_fuse", 0)
        ctxs[0].cnn_forward(frames[0], timing_reps=1)
        _, cnn["forward_ms_layer_by_layer"] = ctxs[0].cnn_forward(frames[0], timing_reps=5)
        ctxs[0].set_option("cnn_fuse", cnn["fused"])
        four = torch.cat([frames[0]] * 4)
        ctxs[0].set_option("cnn_chunk", int(four.shape[0]))
        ctxs[0].cnn_forward(four, timing_reps=1)
        _, cnn["forward_ms_four_clips"] = ctxs[0].cnn_forward(four, timing_reps=3)
        cnn["frames_four_clips"] = int(four.shape[0])
        ctxs[0].set_option("cnn_chunk", 128)
        del wts, bss, four

    # LayerNorm over the patch-embed tokens and softmax over the CNN's logits (north_star's stack; extensions, reported apart)
    norm = None
    if not args.no_vit and world == 1:
        rows = 960 * 196
        tokb = torch.randn((rows, 768), dtype=torch.float32, device=dev).to(torch.bfloat16)
        gam, bet = np.ones(768, np.float32), np.zeros(768, np.float32)
        outb = torch.empty_like(tokb)
        ctxs[0].layernorm(tokb, gam, bet, timing_reps=3, out=outb)
        _, ln_ms = ctxs[0].layernorm(tokb, gam, bet, timing_reps=20, out=outb)
        lg = torch.randn((n, 1000), dtype=torch.float32, device=dev)
        ctxs[0].softmax(lg, timing_reps=3)
        _, sm_ms = ctxs[0].softmax(lg, timing_reps=50)
        norm = {"rows": rows, "ln_ms": ln_ms, "softmax_ms": sm_ms, "softmax_rows": n}
        del tokb, outb, lg

    # audio analyzer, reported apart: the clip's 60 s sound track as 120 half-second windows
    audio = None
    if not args.no_vit and world == 1:
        from avd_hip import audio as host_audio
        rnga = np.random.default_rng(11)
        ta = np.arange(int(16000 * meta["duration"])) / 16000.0
        wave_f32 = (0.3 * np.sin(2 * np.pi * 140.0 * ta) * (np.sin(2 * np.pi * 0.9 * ta) > -0.2) + 0.05 * rnga.standard_normal(ta.size)).astype(np.float32)
        dwave = torch.from_numpy(wave_f32).to(dev)
        ctxs[0].audio_features(dwave, 8000)
        t1 = time.perf_counter()
        for _ in range(5):
            reca = ctxs[0].audio_features(dwave, 8000)
        dt_feat = (time.perf_counter() - t1) / 5
        t1 = time.perf_counter()
        res_a = host_audio.features_to_result(host_audio.window_values(reca), len(wave_f32) / 16000.0)
        audio = {"windows": int(len(reca)), "features_ms": dt_feat * 1e3, "tail_ms": (time.perf_counter() - t1) * 1e3,
                 "speech_ratio": res_a["scores"]["speech_ratio"]}
        del dwave

    # ---- fb_mode exact beside the default (fast): latency of a clip alone, level-0 time, throughput with clips in flight
    exact = None
    if not args.no_extras and world == 1:
        ectx = [avd_hip.Context(dev_index) for _ in range(m)]
        for c in ectx:
            c.set_option("fb_mode", 0)
            c.set_profiling(True)
        erec = [np.zeros(n, avd_hip.RECORD_DTYPE) for _ in range(m)]
        for _ in range(2):
            ectx[0].analyze_frames_async(frames[0], erec[0]); ectx[0].synchronize()
        el, est = [], np.zeros(6)
        for _ in range(5):
            ectx[0].analyze_frames_async(frames[0], erec[0]); ectx[0].synchronize()
            est += np.array(ectx[0].stage_ms())
        est /= 5
        for c in ectx:
            c.set_profiling(False)
        for _ in range(6):                                     # as latency_ms above: records + the host tail, no stage events
            t1 = time.perf_counter()
            ectx[0].analyze_frames_async(frames[0], erec[0]); ectx[0].synchronize()
            ev = records_to_result(erec[0], h * w, w, h, meta["fps"], meta["duration"])
            fusion.fuse(audio_unavailable("", meta), ev, hints)
            el.append(time.perf_counter() - t1)
        el = el[1:]
        thr = []
        for _ in range(5):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            q = []
            for i in range(args.steps):
                if len(q) == m:
                    ectx[q.pop(0)].synchronize()
                j = i % m
                ectx[j].analyze_frames_async(frames[j], erec[j]); q.append(j)
            while q:
                ectx[q.pop(0)].synchronize()
            thr.append(args.steps * n / (time.perf_counter() - t1))
        exact = {"latency_ms": statistics.median(el) * 1e3, "stage": est, "fps": statistics.median(thr),
                 "flow_mean_head": [float(v) for v in erec[0]["flow_mean"][1:4]]}
        for c in ectx:
            c.close()
        del ectx

    # ---- what the exact re-run of flagged pairs costs (default mode): the bench clip with 1 / 12 / all of its pairs replaced by content the level
    # kernels flag (vertical sinusoid stripes: 1-D structure, the normal equations are singular) -- latency of the clip alone and throughput with
    # clips in flight, beside the unflagged clip and fb_mode = exact.  The bench clip itself flags nothing (fb_modes.rerun_pairs).
    rerun_cost = None
    if not args.no_extras and world == 1 and ctxs[0].get_option("fb_mode") == 1 and ctxs[0].get_option("fb_rerun"):
        xs = torch.arange(w, device=dev, dtype=torch.float32)

        def stripe_frame(phase):
            row = (127.0 + 120.0 * torch.sin((xs + phase) * (2.0 * math.pi / 60.0))).round().clamp(0, 255).to(torch.uint8)
            return row[None, :, None].expand(h, w, 3)

        def variant(pairs):
            """frames[0] with `pairs` of its consecutive pairs made of stripe frames"""
            v = frames[0].clone()
            if pairs >= n - 1:
                idx = list(range(n))
            else:
                stride = max(3, (n - 2) // pairs)
                idx = [f for j in range(pairs) for f in (1 + j * stride, 2 + j * stride)]
            for f in idx:
                v[f] = stripe_frame(7.0 * f)
            return v

        rerun_cost = {"content": "vertical sinusoid stripes (period 60 px at 1080p) in place of the frames of 1 / 12 / all pairs of the bench clip",
                      "unflagged_sec_per_video_resident": round(latency_ms / 1e3, 6), "cases": []}
        rrec = [np.zeros(n, avd_hip.RECORD_DTYPE) for _ in range(m)]
        for pairs in (1, 12, n - 1):
            v = variant(pairs)
            for _ in range(2):
                ctxs[0].analyze_frames_async(v, rrec[0]); ctxs[0].synchronize()
            flagged = ctxs[0].get_option("rerun_pairs")
            ll = []
            for _ in range(7):
                t1 = time.perf_counter()
                ctxs[0].analyze_frames_async(v, rrec[0]); ctxs[0].synchronize()
                records_to_result(rrec[0], h * w, w, h, meta["fps"], meta["duration"])
                ll.append(time.perf_counter() - t1)
            thr = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                q = []
                for i in range(args.steps):
                    if len(q) == m:
                        ctxs[q.pop(0)].synchronize()
                    j = i % m
                    ctxs[j].analyze_frames_async(v, rrec[j]); q.append(j)
                while q:
                    ctxs[q.pop(0)].synchronize()
                thr.append(args.steps * n / (time.perf_counter() - t1))
            lm = statistics.median(ll) * 1e3
            rerun_cost["cases"].append({"pairs_replaced": pairs, "pairs_flagged": int(flagged), "sec_per_video_resident": round(lm / 1e3, 6),
                                        "added_ms": round(lm - latency_ms, 4), "frames_per_s": round(statistics.median(thr), 1)})
            del v
        if exact is not None:
            rerun_cost["exact_mode_sec_per_video_resident"] = round(exact["latency_ms"] / 1e3, 6)
            rerun_cost["exact_mode_frames_per_s"] = round(exact["fps"], 1)

    # ---- batches: configs[0]-sized short clips and a configs[4] mixed-resolution stream (avd_analyze_batch)
    batches = None
    if not args.no_extras and world == 1:
        def stream_fps(make_calls, frames_per_round, rounds=6):
            """make_calls(ctx_index) enqueues one round on that context; m rounds in flight"""
            for j in range(m):
                make_calls(j); ctxs[j].synchronize()
            best = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                q = []
                for i in range(rounds):
                    if len(q) == m:
                        ctxs[q.pop(0)].synchronize()
                    make_calls(i % m); q.append(i % m)
                while q:
                    ctxs[q.pop(0)].synchronize()
                best.append(rounds * frames_per_round / (time.perf_counter() - t1))
            return statistics.median(best)

        def one_by_one_fps(clips, rounds=2):
            """the same clips, one avd_analyze_frames_async call per clip, m clips in flight"""
            total = sum(int(c.shape[0]) for c in clips)
            rb = [np.zeros(max(int(c.shape[0]) for c in clips), avd_hip.RECORD_DTYPE) for _ in range(m)]
            best = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                q, i = [], 0
                for _r in range(rounds):
                    for c in clips:
                        if len(q) == m:
                            ctxs[q.pop(0)].synchronize()
                        ctxs[i % m].analyze_frames_async(c, rb[i % m]); q.append(i % m); i += 1
                while q:
                    ctxs[q.pop(0)].synchronize()
                best.append(rounds * total / (time.perf_counter() - t1))
            return statistics.median(best)

        short_src = [torch.from_numpy(synth.make_clip(20, 720, 1280, seed=50 + i)).to(dev) for i in range(3)]
        short = [short_src[i % 3] for i in range(13)]                       # 13 clips x 20 frames = 259 consecutive pairs per batch
        srec = [np.zeros(13 * 20, avd_hip.RECORD_DTYPE) for _ in range(m)]
        short_batched = stream_fps(lambda j: ctxs[j].analyze_batch_async(short, srec[j]), 13 * 20)
        short_single = one_by_one_fps(short)
        mix_geo = [(20, 720, 1280)] * 6 + [(20, 1080, 1920)] * 4 + [(20, 2160, 3840)] * 2
        order = np.random.default_rng(4).permutation(len(mix_geo))
        mix_src = {g: torch.from_numpy(synth.make_clip(*g, seed=70 + g[1])).to(dev) for g in set(mix_geo)}
        mixed = [mix_src[mix_geo[i]] for i in order]                        # shuffled 720p / 1080p / 4K stream, 240 frames
        mrec = [np.zeros(20 * len(mixed), avd_hip.RECORD_DTYPE) for _ in range(m)]
        mixed_batched = stream_fps(lambda j: ctxs[j].analyze_batch_async(mixed, mrec[j]), 20 * len(mixed), rounds=4)
        mixed_single = one_by_one_fps(mixed, rounds=1)
        batches = {"short": (short_batched, short_single), "mixed": (mixed_batched, mixed_single)}
        del short_src, short, mix_src, mixed

    run(args.warmup)                  # W untimed warm-up steps in the timed region's own (pipelined) mode
    elapsed_all, timed_stage = [], np.zeros(6)
    result = fused = None
    for _ in range(max(1, args.repeats)):
        stage[:] = 0
        barrier()
        t0 = time.perf_counter()
        result, fused = run(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            elapsed = float(t.item())
        elapsed_all.append(elapsed)
    elapsed = statistics.median(elapsed_all)
    # the same pipelined mode once more with the stage events on, after the timed repeats: event-to-event stage times under load
    set_prof(True)
    run(min(args.warmup, 2))
    stage[:] = 0
    run(args.steps)
    timed_stage = stage / max(args.steps, 1)
    set_prof(False)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        fps = lambda e: world * n * args.steps / e          # noqa: E731
        dominant, pre, fb = roofline_objects(n, h, w, excl, latency_ms, "exact" if fb_mode_used == "exact" else "fast", ops)
        pmc_k = load_pmc().get("k_fb_level<320>" if fb_mode_used == "exact" else "k_fb_fast<320>", {}) if (n, h, w) == (120, 1080, 1920) else {}
        dominant["traffic_from_profiles"] = os.path.relpath(PMC_FILE, ROOT) if dominant.get("traffic") is not None else None
        dominant["valu_issue_frac_from_profiles"] = pmc_k.get("valu_issue_frac")
        dominant["ta_busy_frac_from_profiles"] = pmc_k.get("ta_busy_frac")
        dominant["frac_of_vector_peak"] = (dominant.get("valu") or {}).get("frac_of_vector_peak")
        if fb_mode_used != "exact":
            # the Farneback yardstick is the vector peak, not HBM.  The headline triple prices the oracle's EXACT operation counts of the
            # level (avdo_ops_*) against the vector peaks of their mix; the bytes view stays beside it (`hbm`).  What sets the step time is in
            # `limiter`: the ISA table of the three role loops (tools/isa_loop_table.py, profiles/r05_isa_k_fb_fast320.txt).
            v = dominant.get("valu") or {}
            dominant["hbm"] = {"achieved": dominant["achieved"], "peak": dominant["peak"], "unit": "GB/s", "frac": dominant["frac"]}
            if v.get("frac_of_vector_peak"):
                t_lvl = float(excl[4]) * 1e-3
                ops_t = (v["f32_ops_per_level"] + v["f64_ops_per_level"]) / t_lvl / 1e12 if t_lvl > 0 else 0.0
                dominant.update({"bound": "valu", "achieved": round(ops_t, 2), "peak": round(ops_t / v["frac_of_vector_peak"], 2),
                                 "unit": "Top/s", "frac": v["frac_of_vector_peak"]})
            dominant["limiter"] = ("two co-limiters that do not overlap, and a workgroup barrier per 4-row step: vector-ALU issue on the three SIMDs that carry a "
                                   "solver wave -- per step X (window sums + 2 x 2 solve in double: 159 f64 + 100 f32 instructions, 849 cycles) + one normal-equation "
                                   "wave (168 f32 + 20 loads, 335) + the chain wave (89 f32 + 35 f64, 307) = ~1 490 of the ~3 200 cycles a step takes (84 steps per launch; "
                                   "PMC: VALU issue 0.54 chip-wide) -- and the texture addresser (129 vector-memory instructions x 16 cycles per step = 0.65; PMC 0.53); "
                                   "the normal-equation waves are both the loads' only issuers and the second-largest VALU consumer "
                                   "(tools/isa_loop_table.py, profiles/r05_isa_k_fb_fast320.txt)")
        else:
            dominant["limiter"] = "the dependent double-add chain of the horizontal scan + VALU issue of the vertical waves, one workgroup per pair on 119 of 256 CUs"
        dominant["kernels"] = kernel_table(n, h, w, kms, None if nv12 is None else nv12["pre_ms"])
        # north_star's ">= 60 % of the HBM peak on preprocessing": the fused kernel alone (its row of the table), not the stage that also
        # holds the aHash kernel and the profiling events
        dominant["preprocess_frac"] = next((k["frac"] for k in dominant["kernels"] if k["name"] == "preprocess"), pre["frac"])
        out = {
            "metric": "sampled frames/sec analysed (1080p30 60 s clip, 2 fps sampling)",
            "value": round(fps(elapsed), 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8 pixels; f32/f64 Farneback (cv2's own types)",
            "data": "synthetic", "fb_mode_used": fb_mode_used,
            "repeats": {"n": len(elapsed_all), "statistic": "median", "value_min": round(fps(max(elapsed_all)), 2),
                        "value_max": round(fps(min(elapsed_all)), 2),
                        "ms_per_step_min": round(min(elapsed_all) / args.steps * 1e3, 4),
                        "ms_per_step_max": round(max(elapsed_all) / args.steps * 1e3, 4)},
            "config": {"workload": "BASELINE.json configs[1]: 1080p30 60 s clip, 2 fps sampling, one clip per GPU per step",
                       "frames_per_clip": n, "height": h, "width": w, "clips_per_step": world, "clips_in_flight_per_gpu": m,
                       "fb_mode": fb_mode_used,
                       "fb_wide160": int(ctxs[0].get_option("fb_wide160")),      # 2 (default) = per call: one strip per pair with clips in flight, two strips for a clip alone
                       "input_copies_in_hbm": m,
                       "sec_per_video": round((host_lat_ms if host_lat_ms is not None else latency_ms) / 1e3, 6),
                       "sec_per_video_note": ("one clip alone, from decoded frames in pinned host memory to the fused result "
                                              "(PCIe-inclusive, median of 5)" if host_lat_ms is not None else
                                              "one clip alone, frames resident in HBM (--no-pcie)"),
                       "sec_per_video_nv12": None if nv12 is None else round(nv12["host_latency_ms"] / 1e3, 6),
                       "sec_per_video_nv12_note": "the same clip handed over as pinned NV12 decoder surfaces (1.5 B per pixel over PCIe, median of 5)",
                       "sec_per_video_resident": round(latency_ms / 1e3, 6),
                       "sec_per_video_resident_note": f"one clip alone, frames already in HBM (median of {n_excl}, before the timed region)",
                       "decoded_frame_equivalent_fps": round(fps(elapsed) * 15, 1),
                       "parallelism": f"clip-parallel x{world}, one all-gather ({args.backend}) of 32 B/frame records" if world > 1 else "single GPU",
                       # what the process group itself reports after init (a SCALE record can be checked for N ranks at a glance)
                       "rccl_ranks_seen": (tdist.get_world_size() if use_dist else 1), "collective_backend": (args.backend if use_dist else None)},
            "roofline": dominant,
            "roofline_preprocess": pre,
            "roofline_farneback_stage": fb,
            "stages_ms": {"preprocess": round(float(excl[0]), 4), "clip_table_upload": round(float(excl[1]), 4),
                          "farneback_and_flow_stats": round(float(excl[2]), 4), "records_copy_out": round(float(excl[3]), 4),
                          "level0_all_iterations": round(float(excl[4]), 4),
                          "note": "HIP events on the library's stream, clips run alone before the timed region; preprocess includes the aHash kernel, "
                                  "farneback_and_flow_stats the record kernel (Hamming distances, record assembly)"},
            "stages_ms_timed_region": {"preprocess": round(float(timed_stage[0]), 4),
                                       "farneback_and_flow_stats": round(float(timed_stage[2]), 4),
                                       "level0_all_iterations": round(float(timed_stage[4]), 4),
                                       "note": "event-to-event times while other clips share the GPU"},
            "result_check": {"ai_timeline_head": [round(v, 6) for v in result["timeline"][:3]],
                             "dup_density": result["summary"]["dup_density"], **fused["result"]},
        }
        # the whole step against the HBM roofline: PMC traffic of one clip (every kernel of the path) over the time the GPU
        # spends per clip at the bench's throughput -- the figure the per-kernel fractions cannot show (the level kernels
        # hold 119 of 256 CUs and leave the bandwidth to the other clips' kernels)
        whole = load_pmc().get("whole_clip", {}).get("hbm_bytes") if (n, h, w) == (120, 1080, 1920) else None
        if whole is not None and world == 1:
            gbs = whole / (ms_per_step * 1e-3) / 1e9
            out["roofline_whole_step"] = {
                "what": "all kernels of one clip (preprocess, hash, pyramid, polynomial expansion, 4 level kernels, flow_up, statistics)",
                "bound": "hbm", "traffic": whole, "algorithmic_bytes_input_only": preprocess_bytes_per_frame(h, w) * n,
                "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "note": "PMC HBM traffic per clip (traffic replayed from " + os.path.relpath(PMC_FILE, ROOT) + ") / ms_per_step with clips in flight"}
            out["roofline"]["whole_step_frac"] = out["roofline_whole_step"]["frac"]
        if pcie_fps is not None:
            out["pcie_inclusive_fps"] = round(pcie_fps_inflight if pcie_fps_inflight is not None else pcie_fps, 1)
            out["pcie_inclusive_fps_one_clip_at_a_time"] = round(pcie_fps, 1)
        if nv12 is not None:
            nv_alg = n * (h * w * 3 // 2 + 320 * 320 + 1024 + 16)
            nv_ach = nv_alg / (nv12["pre_ms"] * 1e-3) / 1e9
            out["roofline_nv12_ingest"] = {
                "kernel": "k_preprocess_nv12 (NV12 surface -> libswscale-style BGR in registers -> gray in LDS -> the fused phases; no BGR in HBM)",
                "bound": "hbm", "achieved": round(nv_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nv_ach / HBM_PEAK_GBS, 4),
                "traffic": None, "algorithmic_bytes_per_launch": nv_alg, "bytes_per_frame_read": h * w * 3 // 2,
                "avg_launch_ms": round(nv12["pre_ms"], 4),
                "stage_ms_with_hash_and_events": round(nv12["pre_stage_ms"], 4),   # what `avg_launch_ms` held in rounds 3 and 4 (0.27-0.28)
                "bound_note": "VALU-bound: ~20 integer operations per pixel for the three clipped table values and the gray",
                "pcie_inclusive_fps_one_clip_at_a_time": round(nv12["host_fps"], 1),
                "pcie_inclusive_fps": None if "host_fps_inflight" not in nv12 else round(nv12["host_fps_inflight"], 1),
                "pcie_note": "pinned NV12 surfaces staged inside the call: 373 MB per clip, the link (~55 GB/s) bounds it at ~17.6 k frames/s"}
        if vit is not None:
            mm = vit["frames"] * 196
            fl = 2.0 * mm * 768 * 768
            tf = fl / (vit["gemm_ms"] * 1e-3) / 1e12
            out["mfma_patch_embed"] = {
                "kernel": "k_gemm_bf16_nt_persistent (ViT-B/16 patch embedding: [frames x 196, 768] x [768, 768], bf16 in, f32 accumulate, "
                          "v_mfma_f32_16x16x32_bf16, 256x256 tiles, LDS-DMA ring that runs on across tiles, one workgroup per CU)",
                "extension": "no reference counterpart (the reference has no learned model); seeded random weights; not part of value / ai_score",
                "bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                "tokens_dtype": "bf16 (f32 accumulate, rounded once)", "achieved_with_f32_tokens": round(fl / (vit["gemm_ms_f32_tokens"] * 1e-3) / 1e12, 1),
                "M": mm, "N": 768, "K": 768, "flops_per_launch": fl, "avg_launch_ms": round(vit["gemm_ms"], 4),
                "frames_per_launch": vit["frames"], "patchify_plus_call_overhead_ms": round(vit["whole_call_ms"], 3),
                "timed": "20 launches between two HIP events on the library's stream, patches resident in HBM"}
        if cnn is not None:
            fl = 2.0 * cnn["macs_per_frame"] * cnn["frames"]
            tf = fl / (cnn["forward_ms"] * 1e-3) / 1e12
            cnn_traffic, cnn_traffic_file = None, None
            try:
                for name in ("r04_pmc_extensions.json", "r02_pmc_extensions.json"):
                    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
                    if os.path.exists(path):
                        ext = json.load(open(path))
                        if cnn["frames"] == 120:
                            cnn_traffic = ext["cnn_forward_120_frames"]["hbm_bytes"]
                            cnn_traffic_file = "profiles/" + name
                        break
            except (OSError, KeyError, ValueError):
                pass
            tf4 = 2.0 * cnn["macs_per_frame"] * cnn["frames_four_clips"] / (cnn["forward_ms_four_clips"] * 1e-3) / 1e12
            out["mfma_cnn_forward"] = {
                "kernel": "k_conv_bf16 + k_conv3_expand (ResNet-50-style forward: every convolution one implicit GEMM, the activation operand gathered by "
                          "LDS-DMA from blocked + swizzled bf16 activations, bias / residual / ReLU fused; in the 56x56 and 28x28 stages a block's 3x3 and its "
                          "expanding 1x1 are ONE launch with the mid activation in LDS, in the stride-1 blocks with the 3x3's input as one slab in LDS (k_slab3_expand); the 7x7 stem gathers pixel pairs from a zero-bordered input image) + input conversion, max / average pooling, linear",
                "extension": "no reference counterpart (the reference has no learned model); seeded random weights; not part of value / ai_score",
                "bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                "frames_per_forward": cnn["frames"], "gmac_per_frame": round(cnn["macs_per_frame"] / 1e9, 3),
                "forward_ms": round(cnn["forward_ms"], 3), "frames_per_s": round(cnn["frames"] / (cnn["forward_ms"] * 1e-3), 1),
                "launches_per_forward": 50 if cnn["fused"] else 57, "top1_head": cnn["top1_head"],
                "forward_ms_layer_by_layer": round(cnn["forward_ms_layer_by_layer"], 3),
                "four_clips_per_pass": {"frames": cnn["frames_four_clips"], "forward_ms": round(cnn["forward_ms_four_clips"], 3),
                                        "achieved": round(tf4, 1), "frac": round(tf4 / 2500.0, 4)},
                "hbm_traffic_bytes_per_forward": cnn_traffic, "hbm_traffic_from_profiles": cnn_traffic_file,
                "hbm_gbps": None if cnn_traffic is None else round(cnn_traffic / (cnn["forward_ms"] * 1e-3) / 1e9, 1),
                "hbm_note": "PMC traffic of 120 frames: bf16 activations written and read once per layer were 63 MB per frame layer by layer (round 2); "
                            "what bounds a layer is the per-CU L2 -> LDS ingest (~40 GB/s per CU, tools/ldsdma_bench.hip), see profiles/r04_experiments.md section 5",
                "timed": "5 whole forward passes (BGR frames in HBM to logits) between two HIP events on the library's stream"}
        if norm is not None:
            lnb = norm["rows"] * 768 * 4                             # bf16 tokens read once and written once
            gbs = lnb / (norm["ln_ms"] * 1e-3) / 1e9
            out["layernorm_tokens"] = {
                "kernel": "k_layernorm<3, bf16> (LayerNorm over the 768 values of a patch-embed token: one wave per row, the row in registers, "
                          "wave-shuffle reductions, float32 statistics; one pass over HBM)",
                "extension": "no reference counterpart; north_star's conv / GEMM / LayerNorm / softmax stack; not part of value / ai_score",
                "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "rows": norm["rows"], "algorithmic_bytes_per_launch": lnb, "avg_launch_ms": round(norm["ln_ms"], 4),
                "softmax_1000_logits": {"rows": norm["softmax_rows"], "avg_launch_ms": round(norm["softmax_ms"], 4),
                                        "note": "120 rows of 4 KB: a launch-latency-sized kernel (one wave per row)"}}
        if audio is not None:
            out["audio_analyzer"] = {
                "what": "avd_audio_features: RMS / zero crossings / Hann + 8000-point real DFT in double (80 x 100 two-step, exact twiddle table) / flatness, roll-off, centroid "
                        "sums for every half-second window of a 60 s 16 kHz sound track in one call (reference audio.py:40-61), reported apart",
                "windows": audio["windows"], "gpu_call_ms": round(audio["features_ms"], 3), "host_tail_ms": round(audio["tail_ms"], 3),
                "windows_per_s": round(audio["windows"] / (audio["features_ms"] * 1e-3), 1), "f64_gflop_per_call_direct_form": round(audio["windows"] * 4001 * 8000 * 4 / 1e9, 2)}
        if exact is not None:
            edom, _, efb = roofline_objects(n, h, w, exact["stage"], exact["latency_ms"], "exact", ops)
            out["fb_modes"] = {
                "value_uses": fb_mode_used, "rerun_pairs": rerun_pairs,
                "guarantee": "pairs flagged by the level kernels (singular normal equations; sign of a residue-sized flow at the top / left border) re-run by the "
                             "exact kernels: bit-identical; others flow <= 1e-5 px, flow_mean/var rel 1e-6, ai_susp 1e-6 -- no content family excepted "
                             "(tests/test_gpu_fbfast.py, tests/test_gpu_soak.py: 28 families)",
                "exact_frames_per_s": round(exact["fps"], 2), "exact_level320_ms": round(float(exact["stage"][4]), 4),
                "rerun_cost": rerun_cost,
                "fast": {"what": "csrc/avd_fbfast.hip + exact re-run of flagged pairs (host-driven, compacted list): flow identical to the oracle on well-posed inputs (<= 1e-5 px), ill-posed pairs re-run exactly",
                         "frames_per_s": out["value"], "sec_per_video_resident": round(latency_ms / 1e3, 6),
                         "level0_all_iterations_ms": round(float(excl[4]), 4), "farneback_and_flow_stats_ms": round(float(excl[2]), 4),
                         "flow_mean_head": [float(v) for v in recs[0]["flow_mean"][1:4]]},
                "exact": {"what": "csrc/avd_fbfused.hip: bit-identical to the oracle, one workgroup per pair",
                          "frames_per_s": round(exact["fps"], 2), "sec_per_video_resident": round(exact["latency_ms"] / 1e3, 6),
                          "level0_all_iterations_ms": round(float(exact["stage"][4]), 4),
                          "farneback_and_flow_stats_ms": round(float(exact["stage"][2]), 4),
                          "flow_mean_head": exact["flow_mean_head"], "roofline": edom}}
        if batches is not None:
            out["short_clips_fps"] = {
                "workload": "BASELINE.json configs[0]-sized clips: 20 sampled 720p frames each, resident in HBM, 13 clips per batch "
                            "(avd_analyze_batch: one Farneback launch sequence over the 259 pairs of a batch), batches in flight as clips are",
                "value": round(batches["short"][0], 1), "unit": "frames/s",
                "one_clip_per_call": round(batches["short"][1], 1)}
            out["mixed_stream_fps"] = {
                "workload": "BASELINE.json configs[4]: shuffled stream of 20-frame clips, 6 x 720p + 4 x 1080p + 2 x 4K per batch of 240 frames "
                            "(cached geometry tables, nothing allocated in steady state; the Farneback stage batches uniformly at 320 x 320)",
                "value": round(batches["mixed"][0], 1), "unit": "frames/s",
                "one_clip_per_call": round(batches["mixed"][1], 1)}
        if "fb_modes" not in out:
            out["fb_modes"] = {"value_uses": fb_mode_used, "rerun_pairs": rerun_pairs}
        out["ops"] = ops
        out["cpu_baseline"] = cpu_base
        if args.details:
            with open(args.details, "w") as fh:
                json.dump(out, fh, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(compact_line(out, args.details)) + "\n").encode())
    if use_dist:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
