"""LayerNorm / softmax kernels (csrc/avd_norm.hip) against float32 torch on the CPU.  Build-defined extensions named by
BASELINE.json's north_star ("conv / GEMM / LayerNorm / softmax stack"); the reference has no learned model, so there is no
reference site (its only per-frame "model" is app/analyzers/video.py:54-56).  Tolerances: 2e-6 absolute for float32 (outputs
are O(1); the kernels sum in a different order than torch), one bf16 ulp for bf16 storage."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(5, 768), (196 * 3, 768), (7, 256), (9, 2048), (1, 1024), (7, 33), (3, 1), (4, 1000), (2, 5000)])   # the last four: the general kernel (any row length)
def test_layernorm_f32_against_torch(ctx, rows, cols):
    rng = np.random.default_rng(rows + cols)
    x = (rng.standard_normal((rows, cols)) * 3 + rng.standard_normal((rows, 1)) * 5).astype(np.float32)
    g = rng.standard_normal(cols).astype(np.float32)
    b = rng.standard_normal(cols).astype(np.float32)
    want = torch.nn.functional.layer_norm(torch.from_numpy(x), (cols,), torch.from_numpy(g), torch.from_numpy(b), 1e-5).numpy()
    got, _ = ctx.layernorm(x, g, b, 1e-5)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * max(1.0, float(np.abs(want).max())))
    dev, _ = ctx.layernorm(torch.from_numpy(x).to("cuda:0"), g, b, 1e-5)
    assert np.array_equal(dev.cpu().numpy(), got)               # host-staged and device-resident operands: the same kernel


def test_layernorm_bf16_tokens(ctx):
    """bf16 in / bf16 out (the patch-embed GEMM's token format), float32 statistics: within one bf16 ulp of torch's float32
    result rounded to bf16; a constant row gives exactly beta."""
    rng = np.random.default_rng(3)
    rows, cols = 392, 768
    x = torch.from_numpy((rng.standard_normal((rows, cols)) * 2).astype(np.float32)).to(torch.bfloat16)
    x[5] = 1.25
    g = rng.standard_normal(cols).astype(np.float32)
    b = rng.standard_normal(cols).astype(np.float32)
    want = torch.nn.functional.layer_norm(x.float(), (cols,), torch.from_numpy(g), torch.from_numpy(b), 1e-5)
    got, ms = ctx.layernorm(x.to("cuda:0"), g, b, 1e-5, timing_reps=2)
    got = got.cpu().float()
    ulp = torch.maximum(want.abs(), torch.tensor(2.0 ** -126)) * 2.0 ** -7
    assert torch.all((got - want).abs() <= ulp + 1e-6), float(((got - want).abs() / ulp).max())
    assert torch.equal(got[5], torch.from_numpy(b).to(torch.bfloat16).float()) and ms > 0
    # any other row length goes through the general kernel (round 4 refused it): bf16 tokens of 100 and 770 values
    for c2 in (100, 770):
        x2 = torch.from_numpy((rng.standard_normal((9, c2)) * 2).astype(np.float32)).to(torch.bfloat16)
        g2, b2 = rng.standard_normal(c2).astype(np.float32), rng.standard_normal(c2).astype(np.float32)
        want2 = torch.nn.functional.layer_norm(x2.float(), (c2,), torch.from_numpy(g2), torch.from_numpy(b2), 1e-5)
        got2 = ctx.layernorm(x2.to("cuda:0"), g2, b2, 1e-5)[0].cpu().float()
        ulp2 = torch.maximum(want2.abs(), torch.tensor(2.0 ** -126)) * 2.0 ** -7
        assert torch.all((got2 - want2).abs() <= ulp2 + 1e-6), c2


@pytest.mark.parametrize("rows,cols", [(120, 1000), (3, 4), (5, 4096), (2, 1024), (7, 33), (3, 1), (2, 4100), (5, 1001)])   # the last four: the general kernel
def test_softmax_against_torch(ctx, rows, cols):
    rng = np.random.default_rng(cols)
    x = (rng.standard_normal((rows, cols)) * 6).astype(np.float32)
    x[0, :min(3, cols)] = [80.0, -90.0, 79.5][:cols]                # large logits: the max subtraction matters
    want = torch.softmax(torch.from_numpy(x), dim=1).numpy()
    got, _ = ctx.softmax(x)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(got.sum(axis=1), 1.0, atol=1e-6)
    dev, ms = ctx.softmax(torch.from_numpy(x).to("cuda:0"), timing_reps=2)
    assert np.array_equal(dev.cpu().numpy(), got) and ms > 0


def test_classifier_stack_end_to_end(ctx):
    """The stack north_star names, chained on the device: ViT patch embedding (GEMM) -> LayerNorm of the tokens; CNN forward
    (convolutions / GEMM) -> softmax of the logits: every stage a hand-written kernel, results against torch."""
    from avd_hip import cnn as host_cnn, synth
    frames = synth.make_clip(2, 96, 128, seed=8, dup_every=0)
    rng = np.random.default_rng(0)
    ctx.vit_set_weights((rng.standard_normal((768, 768)) * 0.02).astype(np.float32), None)
    tok, _ = ctx.vit_patch_embed(frames)                             # [2, 196, 768] float32
    g, b = np.ones(768, np.float32), np.zeros(768, np.float32)
    ln, _ = ctx.layernorm(tok.reshape(-1, 768), g, b)
    want = torch.nn.functional.layer_norm(torch.from_numpy(tok.reshape(-1, 768)), (768,)).numpy()
    np.testing.assert_allclose(ln, want, rtol=0, atol=1e-5)
    w, bs = host_cnn.seeded_parameters(0)
    ctx.cnn_set_weights(w, bs)
    logits, _ = ctx.cnn_forward(frames)
    prob, _ = ctx.softmax(logits)
    np.testing.assert_allclose(prob, torch.softmax(torch.from_numpy(logits), dim=1).numpy(), rtol=2e-6, atol=1e-9)
    assert prob.shape == (2, 1000) and np.all(prob.argmax(axis=1) == logits.argmax(axis=1))


@pytest.mark.parametrize("rows,cols", [(3, 4), (2, 1000), (1, 1000), (7, 36), (1, 4), (5, 100)])
def test_host_softmax_on_a_fresh_context(rows, cols):
    """host operands are staged through a device buffer whose size must come from the SAME expressions as its
    layout (gamma | beta | pad to 64 floats | x rounded up to 256 bytes | y).  Round 3 reserved 2 * bytes + 8 * cols + 256 B, less than
    pad + round-up need for these shapes (176 / 128 / 32 bytes short); it only passed because an earlier ViT call had grown the
    buffer.  Here every shape gets a FRESH context (nothing grown before), twice, and a larger call in between."""
    import avd_hip
    rng = np.random.default_rng(rows * 1000 + cols)
    x = (rng.standard_normal((rows, cols)) * 4).astype(np.float32)
    want = torch.softmax(torch.from_numpy(x), dim=1).numpy()
    with avd_hip.Context(0) as c:
        got, _ = c.softmax(x)
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-9)
        big = (rng.standard_normal((64, 512)) * 2).astype(np.float32)
        gb, _ = c.softmax(big)
        np.testing.assert_allclose(gb, torch.softmax(torch.from_numpy(big), dim=1).numpy(), rtol=2e-6, atol=1e-9)
        got2, _ = c.softmax(x)
        assert np.array_equal(got, got2)
        odd = (rng.standard_normal((2, 33)) * 3).astype(np.float32)   # a width the register-resident kernel does not take: the general one does
        np.testing.assert_allclose(c.softmax(odd)[0], torch.softmax(torch.from_numpy(odd), dim=1).numpy(), rtol=2e-6, atol=1e-9)
