"""ViT-B/16 patch embedding on the matrix cores -- a build-defined extension with NO
reference counterpart (the reference has no learned model).  Its oracle is therefore a numpy
restatement: the same float32 bilinear / normalise / bf16 rounding, then the product in FLOAT64 on the bf16-rounded
operands.  Tolerance (stated here, north_star asks for 1e-4 only on the reference's own outputs): the MFMA accumulates
768 products in float32, |error| <= 1e-3 + 1e-3 |ref| is generous by two orders of magnitude."""
import numpy as np
import pytest

from avd_hip import _lib, synth

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
ISTD = (np.float32(1) / np.array([0.229, 0.224, 0.225], np.float32)).astype(np.float32)


def patchify_reference(frames):
    """uint8[N,H,W,3] BGR -> float32 (bf16-rounded) [N*196, 768], k = c*256 + py*16 + px, channels RGB."""
    n, h, w, _ = frames.shape
    f32 = np.float32

    def axis(size, dst=224):
        s = f32(size) / f32(dst)
        f = ((np.arange(dst, dtype=np.float32) + f32(0.5)) * s - f32(0.5)).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        fr = (f - i0.astype(np.float32)).astype(np.float32)
        lo, hi = i0 < 0, i0 >= size - 1
        fr[lo | hi] = 0
        i0[lo] = 0
        i0[hi] = size - 1
        return i0, np.minimum(i0 + 1, size - 1), fr

    x0, x1, fx = axis(w)
    y0, y1, fy = axis(h)
    out = np.empty((n, 196, 768), np.float32)
    for c in range(3):
        src = frames[..., 2 - c].astype(np.float32)
        p00, p01 = src[:, y0][:, :, x0], src[:, y0][:, :, x1]
        p10, p11 = src[:, y1][:, :, x0], src[:, y1][:, :, x1]
        top = p00 + (p01 - p00) * fx[None, None, :]
        bot = p10 + (p11 - p10) * fx[None, None, :]
        v = top + (bot - top) * fy[None, :, None]
        v = ((v * (f32(1) / f32(255)) - MEAN[c]) * ISTD[c]).astype(np.float32)
        v = _lib.bf16_bits_to_f32(_lib.f32_to_bf16_bits(v)).reshape(n, 14, 16, 14, 16)      # [n, gy, py, gx, px]
        out[:, :, c * 256:(c + 1) * 256] = v.transpose(0, 1, 3, 2, 4).reshape(n, 196, 256)
    return out.reshape(n * 196, 768)


def test_bf16_rounding_helper():
    x = np.array([1.0, 1.00390625, 1.005859375, -2.5, 3.1415927, 0.0, 1e-20], np.float32)
    b = _lib.f32_to_bf16_bits(x)
    back = _lib.bf16_bits_to_f32(b)
    assert back[0] == 1.0 and back[3] == -2.5 and back[5] == 0.0
    assert back[1] == 1.0 and back[2] == np.float32(1.0078125)           # ties to even, then up
    assert np.all(np.abs(back - x) <= np.abs(x) * 2.0 ** -8)


def test_patchify_reference_layout():
    fr = np.zeros((1, 224, 224, 3), np.uint8)
    fr[0, 17, 35] = (10, 20, 30)                                         # B, G, R at y=17 (patch row 1, py 1), x=35 (patch col 2, px 3)
    a = patchify_reference(fr).reshape(196, 768)
    patch, off = 1 * 14 + 2, 1 * 16 + 3
    for c, val in enumerate((30, 20, 10)):                              # k order is R, G, B planes
        want = (np.float32(val) * (np.float32(1) / np.float32(255)) - MEAN[c]) * ISTD[c]
        assert abs(a[patch, c * 256 + off] - want) <= abs(want) * 2.0 ** -8 + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w", [(3, 224, 224), (2, 360, 640), (5, 1080, 1920), (1, 135, 240), (1, 2160, 3840)])
def test_patch_embed_against_numpy(ctx, n, h, w):
    rng = np.random.default_rng(n * 1000 + h)
    frames = synth.random_frames(n, h, w, seed=h) if h != 360 else synth.make_clip(n, h, w, seed=3)
    weight = (rng.standard_normal((768, 768)) * 0.02).astype(np.float32)
    bias = (rng.standard_normal(768) * 0.1).astype(np.float32)
    ctx.vit_set_weights(weight, bias)
    tokens, ms = ctx.vit_patch_embed(frames, timing_reps=2)
    assert tokens.shape == (n, 196, 768) and ms > 0
    a = patchify_reference(frames).astype(np.float64)
    wq = _lib.bf16_bits_to_f32(_lib.f32_to_bf16_bits(weight)).astype(np.float64)
    ref = a @ wq.T + bias.astype(np.float64)
    got = tokens.reshape(-1, 768).astype(np.float64)
    err = np.abs(got - ref)
    assert np.all(err <= 1e-3 + 1e-3 * np.abs(ref)), (err.max(), np.abs(ref).max())
    # bf16 tokens: the same product rounded once to bf16 (nearest even): within half a bf16 ulp of the f32 result
    tok16, _ = ctx.vit_patch_embed(frames, bf16=True)
    assert np.all(np.abs(tok16.reshape(-1, 768).astype(np.float64) - got) <= np.abs(got) * 2.0 ** -8 + 1e-30)
    # asymmetric check of the fragment / output maps: a one-hot weight row copies one input column to one output column
    onehot = np.zeros((768, 768), np.float32)
    onehot[np.arange(768), (np.arange(768) * 7 + 3) % 768] = 1.0
    ctx.vit_set_weights(onehot, None)
    tok2, _ = ctx.vit_patch_embed(frames)
    assert np.array_equal(tok2.reshape(-1, 768), patchify_reference(frames)[:, (np.arange(768) * 7 + 3) % 768])


@pytest.mark.gpu
def test_patch_embed_device_output_and_rows_not_multiple_of_tile(ctx):
    torch = pytest.importorskip("torch")
    frames = synth.random_frames(7, 96, 128, seed=9)                      # M = 1372 rows: 5 full 256-row tiles + 92 rows
    rng = np.random.default_rng(5)
    weight = (rng.standard_normal((768, 768)) * 0.02).astype(np.float32)
    ctx.vit_set_weights(weight, None)
    host, _ = ctx.vit_patch_embed(frames)
    out = torch.full((7, 196, 768), float("nan"), dtype=torch.float32, device="cuda:0")
    dev, _ = ctx.vit_patch_embed(torch.from_numpy(frames).to("cuda:0"), out=out)
    assert dev is out and np.array_equal(out.cpu().numpy(), host) and np.isfinite(host).all()
    out16 = torch.zeros((7, 196, 768), dtype=torch.bfloat16, device="cuda:0")
    ctx.vit_patch_embed(torch.from_numpy(frames).to("cuda:0"), out=out16, bf16=True)
    assert np.array_equal(out16.float().cpu().numpy(), ctx.vit_patch_embed(frames, bf16=True)[0])


@pytest.mark.gpu
def test_gemm_wave_shapes_give_the_same_tokens(ctx):
    """Option "gemm_waves": the 256 x 256 tile on eight waves of 8 x 4 MFMA tiles (default) or on sixteen waves of 4 x 4 (round 5's A/B of the
    operand-ingest question, profiles/r05_experiments.md section 3).  Every output is the same chain of MFMAs over k in both: identical bits."""
    frames = synth.random_frames(9, 120, 160, seed=4)                      # M = 1764 rows: tiles with and without padding rows
    rng = np.random.default_rng(6)
    ctx.vit_set_weights((rng.standard_normal((768, 768)) * 0.02).astype(np.float32), (rng.standard_normal(768) * 0.1).astype(np.float32))
    assert ctx.get_option("gemm_waves") == 8
    t8, _ = ctx.vit_patch_embed(frames)
    b8, _ = ctx.vit_patch_embed(frames, bf16=True)
    ctx.set_option("gemm_waves", 16)
    try:
        t16, _ = ctx.vit_patch_embed(frames)
        b16, _ = ctx.vit_patch_embed(frames, bf16=True)
    finally:
        ctx.set_option("gemm_waves", 8)
    assert np.array_equal(t8, t16) and np.array_equal(b8, b16) and np.isfinite(t8).all()
