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
