"""BASELINE.json configs[2..4] on ONE GPU, through the product path, against the oracle:

  configs[2]  many concurrent 1080p clips, whole clips per rank, records reassembled with one equal-count
              all-gather (``dist.gather_fixed``)                      -> test_cfg2_whole_clip_sharding_two_ranks
  configs[3]  4K30 120 s clip, dense 8 fps sampling, long enough to cross the 128-pair Farneback chunk
              boundary, through ``app.analyzers.video.analyze``      -> test_cfg3_4k_dense_sampling_through_video_analyze
  configs[4]  mixed-resolution stream (720p / 1080p / 4K) with clips in flight
                                                                     -> test_cfg4_mixed_resolution_stream_against_oracle
(configs[0] / [1] geometries are covered in test_gpu_parity.py.)  Reference sites: app/analyzers/video.py:19
(sampler), :27-58 (per-frame loop), :61-83 (summary / timeline)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_records(oracle, clip):
    """avd_frame_record array of a clip from the ORACLE's stage outputs."""
    from tests.test_host_and_abi import _records_from_oracle
    return _records_from_oracle(oracle, clip)


def _assert_records_equal(got, want, tag=""):
    for key in ("lap_sum", "lap_sumsq", "flow_mean", "flow_var", "ham"):
        assert np.array_equal(got[key], want[key]), (tag, key, np.flatnonzero(got[key] != want[key])[:8])


# ---- configs[3] ---------------------------------------------------------------------------------
class _Synthetic4KSource:
    """A 4K30 'decoder' for ``video.analyze``: frame i of the stream is base[pattern[i // 4]] when i is a
    multiple of the sampling step 4; the frames in between are never materialised (the reference only
    ``grab``s them, video.py:28).  Mirrors avd_hip.sources.FrameSource."""

    def __init__(self, base, pattern, fps=30.0):
        self.base, self.pattern = base, pattern
        self.fps, self.height, self.width = fps, int(base.shape[1]), int(base.shape[2])
        self.frame_count = 4 * len(pattern)
        self.steps_seen, self.closed = [], False

    def sampled(self, step):
        self.steps_seen.append(step)
        for j in self.pattern:
            yield self.base[j]

    def close(self):
        self.closed = True


def _pattern(n, nbase):
    """Index pattern over the base frames: mostly changing, every 10th sampled frame repeats its predecessor."""
    out = []
    for k in range(n):
        out.append(out[-1] if (k and k % 10 == 0) else (k * 5 + k // 7) % nbase)
    return np.array(out)


def test_cfg3_4k_dense_sampling_through_video_analyze(ctx, oracle, monkeypatch):
    """240 sampled 4K frames (4K30, 8 analysed frames per second -> step 4) in ONE HIP call of 241 > 129 frames,
    so the Farneback stage runs in two workspace chunks that overlap by one frame.  The whole result of
    ``video.analyze`` is compared with the oracle run on the same 240 frames, and the device-resident form of
    the same clip (built on the GPU from the six base frames) with the streamed one."""
    torch = pytest.importorskip("torch")
    from app.analyzers import video
    from avd_hip import sources, synth
    n, h, w = 240, 2160, 3840
    base = synth.make_clip(6, h, w, seed=3, dup_every=0)
    pattern = _pattern(n, len(base))
    src = _Synthetic4KSource(base, pattern)
    monkeypatch.setattr(sources, "open_source", lambda path: src)
    monkeypatch.setenv("AVD_SAMPLES_PER_SECOND", "8")
    monkeypatch.setenv("AVD_CHUNK_FRAMES", "256")                 # one HIP call: the clip crosses the 128-pair chunk inside it
    meta = {"width": w, "height": h, "fps": 30.0, "duration": 120.0}
    got = video.analyze("synthetic-4k30.mp4", meta)
    assert src.steps_seen == [4] and src.closed
    assert got["timeline"] is got["timeline_ai"] and len(got["timeline"]) == 120

    # oracle: the per-frame and per-pair values only depend on (frame) and (previous frame, frame); six base frames
    # give at most 36 distinct pairs, so the full 240-frame reference result costs a few seconds
    small, hsh, s, q = oracle.preprocess_bgr(base)
    pair_cache = {}
    want = np.zeros(n, got_dtype())
    want["lap_sum"], want["lap_sumsq"] = s[pattern], q[pattern]
    want["ham"][0] = -1
    for k in range(1, n):
        a, b = int(pattern[k - 1]), int(pattern[k])
        if (a, b) not in pair_cache:
            fm, fv = oracle.farneback_pairs(np.stack([small[a], small[b]]))
            pair_cache[(a, b)] = (fm[0], fv[0], int(np.sum(hsh[a] ^ hsh[b])))
        want["flow_mean"][k], want["flow_var"][k], want["ham"][k] = pair_cache[(a, b)]
    from avd_hip.timeline import records_to_result
    ref = records_to_result(want, h * w, w, h, 30.0, 120.0)
    np.testing.assert_allclose(got["timeline"], ref["timeline"], rtol=0, atol=1e-4)      # north_star tolerance
    assert got["timeline"] == ref["timeline"]                                            # in fact identical
    assert got["summary"] == pytest.approx(ref["summary"], rel=1e-12, abs=1e-12)
    assert got["summary"]["dup_density"] >= 23 / 239 and got["summary"]["w"] == 3840       # every 10th frame repeats
    # and the oracle's own end-to-end restatement (sampler tail included) on a prefix that spans the duplicates
    o_full = oracle.analyze_sampled_frames(base[pattern[:24]], {**meta, "duration": 12.0})
    assert got["timeline"][:12] == o_full["timeline"][:12]

    # device-resident clip built on the GPU from the base frames: same records as the oracle, both Farneback chunks
    dev_base = torch.from_numpy(base).to("cuda:0")
    clip = dev_base[torch.from_numpy(pattern).to("cuda:0")]                              # 240 x 24.9 MB, fresh tensor
    rec = ctx.analyze_frames(clip)
    _assert_records_equal(rec, want, "device-resident 4K clip")
    # properties that do not depend on the size: locality across the chunk boundary, reversal
    sub = ctx.analyze_frames(clip[120:140])
    for key in ("flow_mean", "flow_var", "ham"):
        assert np.array_equal(sub[key][1:], rec[key][121:140]), key
    rev = ctx.analyze_frames(torch.flip(clip, dims=[0]))
    assert np.array_equal(rev["lap_sumsq"], rec["lap_sumsq"][::-1]) and np.array_equal(rev["ham"][1:], rec["ham"][1:][::-1])


def got_dtype():
    import avd_hip
    return avd_hip.RECORD_DTYPE


# ---- configs[4] ---------------------------------------------------------------------------------
def test_cfg4_mixed_resolution_stream_against_oracle(oracle):
    """A shuffled stream of 720p / 1080p / 4K clips, three in flight on their own contexts (each context
    re-plans its tables and workspace when the geometry changes): every clip's records equal the ORACLE's."""
    import avd_hip
    from avd_hip import synth
    geoms = {"720p": (720, 1280), "1080p": (1080, 1920), "4K": (2160, 3840)}
    order = ["1080p", "4K", "720p", "720p", "4K", "1080p", "4K", "720p", "1080p"]
    lengths = [4, 3, 5, 2, 3, 4, 2, 6, 3]
    clips = []
    for i, (g, n) in enumerate(zip(order, lengths)):
        h, w = geoms[g]
        clips.append((f"{i}:{g}", synth.make_clip(n, h, w, seed=300 + i, dup_every=2, scene_cut=False)))
    runner = avd_hip.ClipsInFlight(device=0, depth=3)
    got = list(runner.run(clips))
    assert [t for t, _ in got] == [t for t, _ in clips]
    for (tag, rec), (_, frames) in zip(got, clips):
        want = _oracle_records(oracle, frames)
        _assert_records_equal(rec, want, tag)
        meta = {"width": frames.shape[2], "height": frames.shape[1], "fps": 30.0, "duration": len(frames) / 2.0}
        from avd_hip.timeline import records_to_result
        res = records_to_result(rec, frames.shape[1] * frames.shape[2], meta["width"], meta["height"], 30.0, meta["duration"])
        ref = oracle.analyze_sampled_frames(frames, meta)
        assert res["timeline"] == ref["timeline"], tag


# ---- configs[2] ---------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


CFG2_CLIPS_PER_RANK, CFG2_FRAMES = 4, 5


def _cfg2_clip(rank, j):
    from avd_hip import synth
    return synth.make_clip(CFG2_FRAMES, 1080, 1920, seed=1000 + 10 * rank + j, dup_every=3, scene_cut=(j % 2 == 0))


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
        clips = [(j, _cfg2_clip(rank, j)) for j in range(CFG2_CLIPS_PER_RANK)]          # this rank's whole clips
        runner = avd_hip.ClipsInFlight(device=0, depth=2)                              # both ranks share the test box's one GPU
        local = np.concatenate([rec for _, rec in runner.run(clips)])                    # [4 clips x 5 frames] records
        allrec = avd_dist.gather_fixed(local)                                            # ONE collective, equal counts
        np.save(os.path.join(out_dir, f"cfg2_rank{rank}.npy"), allrec)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_cfg2_whole_clip_sharding_two_ranks(tmp_path, ctx, oracle):
    """2 ranks x 4 whole 1080p clips each on one GPU (HIP path, clips in flight), REAL records through
    ``gather_fixed``: every rank ends up with all 8 clips' records in rank order, equal to a single context
    analysing the 8 clips one after the other, and (two clips) to the oracle."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_cfg2_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    single = np.concatenate([ctx.analyze_frames(_cfg2_clip(r, j)) for r in range(world) for j in range(CFG2_CLIPS_PER_RANK)])
    assert len(single) == world * CFG2_CLIPS_PER_RANK * CFG2_FRAMES
    for r in range(world):
        gathered = np.load(tmp_path / f"cfg2_rank{r}.npy")
        assert gathered.dtype == single.dtype and np.array_equal(gathered, single), r
    for (r, j) in ((0, 1), (1, 2)):
        k = (r * CFG2_CLIPS_PER_RANK + j) * CFG2_FRAMES
        _assert_records_equal(single[k:k + CFG2_FRAMES], _oracle_records(oracle, _cfg2_clip(r, j)), f"rank {r} clip {j}")
    # each clip's first frame starts a new clip: no Hamming / flow carried over from the previous clip
    assert np.all(single["ham"][::CFG2_FRAMES] == -1) and np.all(single["flow_mean"][::CFG2_FRAMES] == 0)


def test_native_rccl_allgather_of_records(ctx):
    """avd_allgather_records: RCCL bound at run time.  A 1-GPU lease can only form a communicator of one
    rank (two ranks on one device are refused by RCCL as duplicate GPUs): the entry points, the staging and the
    collective call itself run; rank order with more ranks is covered by the gloo tests of the same layout."""
    import avd_hip
    from avd_hip import synth
    rec = ctx.analyze_frames(synth.make_clip(5, 96, 160, seed=77, dup_every=2))
    with avd_hip.Context(0) as c:
        with pytest.raises(avd_hip.AvdError):
            c.allgather_records(rec)                                    # no communicator yet
        uid = avd_hip.Context.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        c.comm_init(0, 1, uid)
        got = c.allgather_records(rec)
        assert np.array_equal(got, rec)
        assert np.array_equal(c.allgather_records(rec[:2]), rec[:2])    # another size re-uses the communicator
        # straight from the device records of the context's last call (no host round trip before the collective), behind
        # a blocking call and behind an asynchronous one (which it drains)
        clip = synth.make_clip(5, 96, 160, seed=77, dup_every=2)
        mine = c.analyze_frames(clip)
        assert np.array_equal(c.allgather_last_records(5), mine) and np.array_equal(mine, rec)
        buf = np.zeros(5, avd_hip.RECORD_DTYPE)
        keep = c.analyze_frames_async(clip, buf)
        got = c.allgather_last_records(5)
        assert np.array_equal(got, rec) and np.array_equal(buf, rec)
        assert np.array_equal(c.allgather_last_records(3), rec[:3])
        with pytest.raises(avd_hip.AvdError):
            c.allgather_last_records(6)                                 # more than the last call produced
        with pytest.raises(avd_hip.AvdError):
            c.comm_init(3, 2, uid)                                      # rank out of range
