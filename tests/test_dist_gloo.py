"""N > 1 path on CPU: two ranks (gloo), frame-parallel shards with a one-frame halo, one
all-gather of the per-frame records, identical timeline on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import avd_hip
    from avd_hip import dist as avd_dist, synth, timeline
    from oracle import oracle as O
    from tests.test_host_and_abi import _records_from_oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clip = synth.make_clip(7, 64, 96, seed=13, dup_every=3)          # same clip on every rank
        local = avd_dist.analyze_shard(lambda fr: _records_from_oracle(O, fr), clip, rank, world)
        allrec = avd_dist.all_gather_records(local)                       # variable-length gather
        fixed = avd_dist.gather_fixed(np.zeros(2, avd_hip.RECORD_DTYPE))   # equal-count gather, one collective
        assert len(fixed) == 2 * world
        res = timeline.records_to_result(allrec, 64 * 96, 96, 64, 30.0, 3.0)
        np.save(os.path.join(out_dir, f"rec{rank}.npy"), allrec)
        np.save(os.path.join(out_dir, f"tl{rank}.npy"), np.array(res["timeline"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_equals_single_process(tmp_path, oracle):
    from avd_hip import synth
    from tests.test_host_and_abi import _records_from_oracle
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    clip = synth.make_clip(7, 64, 96, seed=13, dup_every=3)
    whole = _records_from_oracle(oracle, clip)
    for r in range(world):
        rec = np.load(tmp_path / f"rec{r}.npy")
        rec["ham"][0] = -1
        assert np.array_equal(rec, whole)
    assert np.array_equal(np.load(tmp_path / "tl0.npy"), np.load(tmp_path / "tl1.npy"))


def _cfg2_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import avd_hip
    from avd_hip import dist as avd_dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clips_per_rank, frames = 4, 6
        local = np.zeros(clips_per_rank * frames, avd_hip.RECORD_DTYPE)
        for j in range(clips_per_rank):
            clip_id = rank * clips_per_rank + j                           # BASELINE.json configs[2]: clip c lives on rank c // 4
            sl = slice(j * frames, (j + 1) * frames)
            local["lap_sum"][sl] = clip_id * 1000 + np.arange(frames)
            local["ham"][sl] = np.where(np.arange(frames) == 0, -1, clip_id)
            local["flow_mean"][sl] = clip_id + np.arange(frames) / 16.0
        allrec = avd_dist.gather_fixed(local)                             # ONE collective, equal counts
        np.save(os.path.join(out_dir, f"cfg2_{rank}.npy"), allrec)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_cfg2_layout_eight_ranks_gloo(tmp_path):
    """BASELINE.json configs[2]: 32 clips, whole clips per rank (4 each on 8 ranks), timelines reassembled with ONE
    equal-count all-gather of the 32-byte records -- every rank ends up with all 32 clips in clip order."""
    world, port = 8, _free_port()
    mp.spawn(_cfg2_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    frames = 6
    first = np.load(tmp_path / "cfg2_0.npy")
    assert len(first) == 32 * frames
    for c in range(32):
        sl = slice(c * frames, (c + 1) * frames)
        assert np.array_equal(first["lap_sum"][sl], c * 1000 + np.arange(frames)), c
        assert first["ham"][sl][0] == -1 and np.all(first["ham"][sl][1:] == c)
        assert np.array_equal(first["flow_mean"][sl], (c + np.arange(frames) / 16.0).astype(np.float32))
    for r in range(1, world):
        assert np.array_equal(np.load(tmp_path / f"cfg2_{r}.npy"), first), r


def _gpu_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import avd_hip
    from avd_hip import dist as avd_dist, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clip = synth.make_clip(10, 96, 160, seed=23, dup_every=4)        # same clip on every rank
        with avd_hip.Context(0) as ctx:                                  # both ranks share the one GPU of the test box
            local = avd_dist.analyze_shard(ctx.analyze_frames, clip, rank, world)   # shard + one-frame halo
        allrec = avd_dist.all_gather_records(local)
        np.save(os.path.join(out_dir, f"gpu_rec{rank}.npy"), allrec)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_frame_sharded_hip_path_two_ranks(tmp_path, ctx):
    """One clip sharded over two ranks (each analyses its frame range + a one-frame halo through the
    HIP path), records all-gathered: identical to the single-context result."""
    from avd_hip import synth
    world, port = 2, _free_port()
    mp.spawn(_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    clip = synth.make_clip(10, 96, 160, seed=23, dup_every=4)
    whole = ctx.analyze_frames(clip)
    for r in range(world):
        rec = np.load(tmp_path / f"gpu_rec{r}.npy")
        assert np.array_equal(rec, whole)
