"""NV12 ingest: decoder surfaces straight into the fused pass.

The conversion restates libswscale's table-driven C converter (yuv2rgb.c; what cv2.VideoCapture.retrieve() runs on a
decoded picture, reference app/analyzers/video.py:28-32).  PARITY UNPINNED: no libswscale exists here and the
constants are from memory (oracle/avd_oracle.c says which); what these tests pin is (CPU) the restatement against an
independent integer formulation and its known answers, and (GPU) the HIP path, which never materialises BGR, against
oracle NV12->BGR followed by the BGR oracle, bit for bit."""
import numpy as np
import pytest

from avd_hip import synth


def _planes(n, h, w, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (n, h, w), dtype=np.uint8), rng.integers(0, 256, (n, h // 2, w), dtype=np.uint8))


# ---- CPU: the restatement itself -------------------------------------------------------------
def test_tables_equal_the_integer_form(oracle):
    """Every table entry is clip8((c0 + (Y + off) * cy) >> 16) with per-chroma index offsets: numpy int64 mirror."""
    c = oracle.yuv2rgb_consts()
    assert c["cy"] == (65536 * 255) // 219 == 76309
    y, uv = _planes(1, 64, 96, seed=1)
    y[0, :2] = np.arange(192, dtype=np.uint8).reshape(2, 96)           # every luma value somewhere
    bgr = oracle.nv12_to_bgr(y, uv)[0]
    Y = y[0].astype(np.int64)
    U = np.repeat(np.repeat(uv[0][:, 0::2], 2, 0), 2, 1).astype(np.int64)      # nearest chroma: 2x2 blocks share U, V
    V = np.repeat(np.repeat(uv[0][:, 1::2], 2, 0), 2, 1).astype(np.int64)
    off = lambda x, inc: ((x * inc) >> 16) - (inc >> 9)                          # arithmetic (floor) shifts
    ch = lambda o: np.clip((c["c0"] + (Y + o) * c["cy"]) >> 16, 0, 255)
    assert np.array_equal(bgr[..., 2], ch(off(V, c["crv"])))
    assert np.array_equal(bgr[..., 0], ch(off(U, c["cbu"])))
    assert np.array_equal(bgr[..., 1], ch(off(U, c["cgu"]) + off(V, c["cgv"])))


def test_known_answers(oracle):
    grey = np.arange(256, dtype=np.uint8).reshape(1, 2, 128)
    neutral = np.full((1, 1, 128), 128, np.uint8)
    out = oracle.nv12_to_bgr(grey, neutral)[0].reshape(256, 3)
    assert np.all(out[:, 0] == out[:, 1]) and np.all(out[:, 1] == out[:, 2])       # U = V = 128: no colour
    assert np.all(np.diff(out[:, 0].astype(int)) >= 0)                               # monotonic in Y
    assert out[16, 0] == 0 and out[:17].max() == 0                                   # black level 16 and below -> 0
    assert out[235, 0] in (253, 254, 255) and out[255, 0] == 255                     # this restatement's white: 253 (README)
    # BT.601: red has V high, blue has U high
    px = lambda Y, U, V: oracle.nv12_to_bgr(np.full((1, 2, 2), Y, np.uint8), np.array([[[U, V]]], np.uint8))[0, 0, 0]
    b, g, r = px(81, 90, 240)
    assert r > 230 and g < 30 and b < 30
    b, g, r = px(41, 240, 110)
    assert b > 230 and r < 30 and g < 30


def test_roundtrip_through_the_synthetic_encoder(oracle):
    """bgr -> (float BT.601 forward) -> NV12 -> oracle inverse stays within a few grey levels on smooth content."""
    clip = synth.make_clip(2, 96, 160, seed=5, dup_every=0)
    y, uv = synth.bgr_to_nv12(clip)
    back = oracle.nv12_to_bgr(y, uv).astype(int)
    assert np.abs(back - clip.astype(int)).mean() < 4.0


# ---- GPU: the ingest kernel --------------------------------------------------------------------
NV12_GEOMS = [(2, 1080, 1920), (2, 720, 1280), (1, 2160, 3840), (3, 66, 102), (2, 360, 640), (2, 34, 48)]


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w", NV12_GEOMS)
def test_preprocess_nv12_bit_exact(ctx, oracle, n, h, w):
    y, uv = _planes(n, h, w, seed=h + w)
    got = ctx.preprocess_nv12(y, uv)
    want = oracle.preprocess_bgr(oracle.nv12_to_bgr(y, uv))
    for name, a, b in zip(("small320", "hash", "lap_sum", "lap_sumsq"), got, want):
        assert np.array_equal(a, b), (name, h, w)


@pytest.mark.gpu
def test_analyze_nv12_equals_bgr_path_and_oracle(ctx, oracle):
    import avd_hip
    clip = synth.make_clip(6, 360, 640, seed=71, dup_every=3)
    y, uv = synth.bgr_to_nv12(clip)
    rec = ctx.analyze_frames_nv12(y, uv)
    bgr = oracle.nv12_to_bgr(y, uv)
    assert np.array_equal(rec, ctx.analyze_frames(bgr))                 # same records as the BGR entry point on swscale's output
    from tests.test_host_and_abi import _records_from_oracle
    assert np.array_equal(rec, _records_from_oracle(oracle, bgr))
    # strided planes (decoder pitch larger than the width) and a frame gap
    yp = np.zeros((6, 368, 704), np.uint8)
    cp = np.zeros((6, 190, 704), np.uint8)
    yp[:, :360, :640], cp[:, :180, :640] = y, uv
    assert np.array_equal(ctx.analyze_frames_nv12(yp[:, :360, :640], cp[:, :180, :640]), rec)
    with pytest.raises(avd_hip.AvdError):
        ctx.preprocess_nv12(np.zeros((1, 65, 64), np.uint8), np.zeros((1, 32, 64), np.uint8))      # odd height


@pytest.mark.gpu
def test_nv12_device_surfaces_and_async(ctx, oracle):
    torch = pytest.importorskip("torch")
    clip = synth.make_clip(5, 720, 1280, seed=72, dup_every=2)
    y, uv = synth.bgr_to_nv12(clip)
    want = ctx.analyze_frames_nv12(y, uv)
    dy, duv = torch.from_numpy(y).to("cuda:0"), torch.from_numpy(uv).to("cuda:0")
    assert np.array_equal(ctx.analyze_frames_nv12(dy, duv), want)
    rec = np.zeros(5, want.dtype)
    keep = ctx.analyze_frames_nv12_async(dy, duv, rec)
    ctx.synchronize()
    del keep
    assert np.array_equal(rec, want)
