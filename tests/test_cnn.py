"""ResNet-50-style CNN forward on the matrix cores -- a build-defined extension with NO
reference counterpart (the reference has no learned model).  Its oracle is a float32 restatement on
the CPU (torch.nn.functional on the host, plumbing only): the same bf16-rounded weights, every activation rounded to bf16
where the HIP path stores it, accumulation in float32.  Tolerances (stated here; north_star's 1e-4 is for the reference's
own outputs): one layer -- the two float32 accumulation orders can land on opposite sides of a bf16 rounding boundary, so
|error| <= 2^-7 |ref| + 2e-3 (one bf16 ulp); the whole network (53 layers of such roundings) -- logits within 0.8 % of the
largest |logit| (budget derived in test_forward_against_float32, with a negative control: a missing residual in the last
block is 5x outside it) and a correlation above 0.9999."""
import numpy as np
import pytest

from avd_hip import _lib, synth

torch = pytest.importorskip("torch")
F = torch.nn.functional

DEPTH = (3, 4, 6, 3)


def topology(roles=False):
    """[(cin, cout, ksize, stride)] in the order of avd_cnn_set_weights, without the final linear layer."""
    convs, role = [(3, 64, 7, 2)], ["stem"]
    cin = 64
    for st, depth in enumerate(DEPTH):
        mid, out = 64 << st, (64 << st) * 4
        for b in range(depth):
            s = 2 if (b == 0 and st > 0) else 1
            convs += [(cin, mid, 1, 1), (mid, mid, 3, s), (mid, out, 1, 1)]
            role += ["reduce", "spatial", "expand"]
            if b == 0:
                convs.append((cin, out, 1, s))
                role.append("shortcut")
            cin = out
    return (convs, role) if roles else convs


def seeded_parameters(seed=0):
    """He-style random weights (the third convolution of a block and the shortcut damped so that the residual sums stay
    bounded without batch norm), small biases; returns flat float32 arrays in the documented order."""
    rng = np.random.default_rng(seed)
    ws, bs = [], []
    convs, role = topology(roles=True)
    for (cin, cout, k, s), what in zip(convs, role):
        fan = cin * k * k
        std = np.sqrt(2.0 / fan) * (0.5 if what in ("expand", "shortcut") else 1.0)
        ws.append((rng.standard_normal((cout, k, k, cin)) * std).astype(np.float32).ravel())
        bs.append((rng.standard_normal(cout) * 0.05).astype(np.float32))
    ws.append((rng.standard_normal((1000, 2048)) * np.sqrt(1.0 / 2048)).astype(np.float32).ravel())
    bs.append((rng.standard_normal(1000) * 0.05).astype(np.float32))
    return np.concatenate(ws), np.concatenate(bs)


def bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)          # round to nearest even, as the kernels do


def conv_reference(x_nhwc, w, bias, stride, relu, residual=None):
    """float32 NHWC in / out; x, w, residual are rounded to bf16 first; output rounded to bf16."""
    x = bf16(torch.from_numpy(np.ascontiguousarray(x_nhwc))).permute(0, 3, 1, 2)
    wt = bf16(torch.from_numpy(np.ascontiguousarray(w))).permute(0, 3, 1, 2)
    y = F.conv2d(x, wt, torch.from_numpy(bias), stride=stride, padding=w.shape[1] // 2)
    if residual is not None:
        y = y + bf16(torch.from_numpy(np.ascontiguousarray(residual))).permute(0, 3, 1, 2)
    if relu:
        y = torch.relu(y)
    return bf16(y).permute(0, 2, 3, 1).contiguous().numpy()


def input_reference(frames):
    """uint8 BGR [N,H,W,3] -> float32 (bf16-rounded) [N,3,224,224] RGB, the arithmetic of the patch-embed input."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("_vit_reference", os.path.join(os.path.dirname(__file__), "test_vit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    patchify_reference = mod.patchify_reference
    n = frames.shape[0]
    a = patchify_reference(frames).reshape(n, 14, 14, 3, 16, 16)              # [n, gy, gx, c, py, px]
    return np.ascontiguousarray(a.transpose(0, 3, 1, 4, 2, 5).reshape(n, 3, 224, 224))


def forward_reference(frames, weights, biases, break_last_residual=False):
    """break_last_residual: a deliberately WRONG network (the last block forgets its residual) -- the negative control of
    the end-to-end test: its logits must differ from the right ones by far more than the test's tolerance."""
    convs = topology()
    wo = bo = 0
    params = []
    for cin, cout, k, s in convs:
        w = torch.from_numpy(weights[wo:wo + cout * k * k * cin].reshape(cout, k, k, cin))
        params.append((bf16(w).permute(0, 3, 1, 2).contiguous(), torch.from_numpy(biases[bo:bo + cout]), k, s))
        wo += cout * k * k * cin
        bo += cout
    wfc = bf16(torch.from_numpy(weights[wo:wo + 1000 * 2048].reshape(1000, 2048)))
    bfc = torch.from_numpy(biases[bo:bo + 1000])

    def conv(x, i, relu=True, res=None):
        w, b, k, s = params[i]
        y = F.conv2d(x, w, b, stride=s, padding=k // 2)
        if res is not None:
            y = y + res
        return bf16(torch.relu(y) if relu else y)

    x = torch.from_numpy(input_reference(frames))
    x = conv(x, 0)
    x = F.max_pool2d(x, 3, 2, 1)
    li = 1
    for st, depth in enumerate(DEPTH):
        for b in range(depth):
            a1 = conv(x, li)
            a2 = conv(a1, li + 1)
            res = conv(x, li + 3, relu=False) if b == 0 else x
            if break_last_residual and st == len(DEPTH) - 1 and b == depth - 1:
                res = None
            x = conv(a2, li + 2, res=res)
            li += 4 if b == 0 else 3
    pooled = x.mean(dim=(2, 3))
    return (pooled @ wfc.t() + bfc).numpy()


def test_parameter_counts_match_the_documented_topology():
    nw, nb = _lib.Context.cnn_param_counts()
    convs = topology()
    assert len(convs) == 53
    assert nw == sum(co * k * k * ci for ci, co, k, s in convs) + 1000 * 2048 == 25_502_912
    assert nb == sum(co for ci, co, k, s in convs) + 1000
    w, b = seeded_parameters(1)
    assert w.size == nw and b.size == nb


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, relu, residual
    (1, 8, 8, 64, 64, 1, 1, True, False),            # K = 64: two half stages, one tile, 64-channel body
    (2, 14, 14, 64, 128, 3, 1, True, False),         # 3x3 with padding, 128-channel body, 392 rows (tail tile)
    (1, 20, 20, 128, 256, 1, 1, False, True),        # residual without ReLU, 256-channel body
    (2, 16, 16, 128, 128, 3, 2, True, False),        # stride 2 on the 3x3
    (1, 14, 14, 256, 512, 1, 2, False, False),       # strided 1x1 (projection shortcut), two column tiles
    (3, 7, 7, 512, 2048, 1, 1, True, True),          # last stage: 147 rows, eight column tiles, residual + ReLU
    (1, 28, 28, 160, 64, 1, 1, True, False),         # the stem's im2col shape: K = 160 (five half stages)
    (1, 9, 11, 96, 192, 3, 1, True, False),          # odd geometry, channel counts that are not powers of two
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CONV_CASES)
def test_one_convolution_against_float32(ctx, case):
    n, h, w, cin, cout, k, stride, relu, with_res = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((cout, k, k, cin)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = rng.standard_normal((n, ho, wo, cout)).astype(np.float32) if with_res else None
    got = ctx.cnn_conv(x, wt, b, stride=stride, relu=relu, residual=res)
    want = conv_reference(x, wt, b, stride, relu, res)
    assert got.shape == want.shape
    err = np.abs(got - want)
    tol = np.abs(want) * 2.0 ** -7 + 2e-3
    assert np.all(err <= tol), (float(err.max()), int((err > tol).sum()))
    assert np.mean(got == want) > 0.98               # all but the rare rounding-boundary cases are bit-identical


@pytest.mark.gpu
def test_convolution_of_a_delta_is_the_flipped_kernel(ctx):
    """Known-answer: a single one at pixel (3, 4) of channel 5, 3x3 weights = small integers, no bias: the output around
    the pixel is the mirrored kernel (cross-correlation), exactly (integers are exact in bf16)."""
    x = np.zeros((1, 8, 8, 32), np.float32)
    x[0, 3, 4, 5] = 1.0
    wt = np.zeros((64, 3, 3, 32), np.float32)
    wt[:, :, :, 5] = np.arange(64 * 9).reshape(64, 3, 3) % 17 - 8
    y = ctx.cnn_conv(x, wt, np.zeros(64, np.float32), stride=1, relu=False)
    for dy in range(3):
        for dx in range(3):
            assert np.array_equal(y[0, 3 + 1 - dy, 4 + 1 - dx], wt[:, dy, dx, 5])
    assert np.count_nonzero(y) == np.count_nonzero(wt[:, :, :, 5])


@pytest.mark.gpu
def test_forward_against_float32(ctx):
    weights, biases = seeded_parameters(0)
    ctx.cnn_set_weights(weights, biases)
    frames = np.concatenate([synth.make_clip(2, 360, 640, seed=5), synth.random_frames(1, 360, 640, seed=6)])
    logits, ms = ctx.cnn_forward(frames, timing_reps=1)
    assert logits.shape == (3, 1000) and np.all(np.isfinite(logits)) and ms > 0
    want = forward_reference(frames, weights, biases)
    scale = float(np.abs(want).max())
    assert scale > 0.1                                # the seeded network does produce a signal
    # Budget: a layer's two float32 accumulation orders flip < 2 % of its bf16 roundings by one ulp (2^-8 relative, asserted
    # per layer above); 53 such layers in sequence, errors adding like a random walk and averaged over 49 pixels by the
    # pooling: sqrt(53 * 0.02) * 2^-8 ~ 0.4 % of an activation's magnitude at worst, i.e. well under 1 % of the largest logit.
    err = float(np.abs(logits - want).max()) / scale
    print(f"[cnn] end-to-end max |logit error| = {err:.5f} of the largest |logit|")
    assert err <= 0.008
    assert np.corrcoef(logits.ravel(), want.ravel())[0, 1] > 0.9999
    # negative control: a network whose LAST block forgets its residual (the smallest structural error there is) lies far
    # outside that tolerance, so this comparison would notice it
    wrong = forward_reference(frames, weights, biases, break_last_residual=True)
    assert float(np.abs(wrong - want).max()) / scale > 5 * 0.008
    # batch independence: a frame alone gives the same logits as inside the batch
    alone, _ = ctx.cnn_forward(frames[1:2])
    assert np.array_equal(alone[0], logits[1])


@pytest.mark.gpu
def test_every_tiling_gives_the_same_bits(ctx):
    """The 256 x 256 / 256 x 128 / 256 x 64 / 128 x 128 bodies accumulate an output's K terms in the same order."""
    outs = {}
    try:
        for policy in (0, 1, 2):
            ctx.set_option("cnn_tiles", policy)
            res = []
            for (n, h, w, cin, cout, k, stride) in [(2, 24, 24, 64, 256, 1, 1), (1, 40, 40, 128, 128, 3, 1), (1, 33, 31, 256, 512, 3, 2),
                                                    (1, 48, 48, 64, 64, 3, 1)]:
                r = np.random.default_rng(n * 1000 + cout + k)
                x = r.standard_normal((n, h, w, cin)).astype(np.float32)
                wt = (r.standard_normal((cout, k, k, cin)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)
                b = (r.standard_normal(cout) * 0.1).astype(np.float32)
                res.append(ctx.cnn_conv(x, wt, b, stride=stride, relu=True))
            outs[policy] = res
    finally:
        ctx.set_option("cnn_tiles", 0)
    for a, b1, b2 in zip(outs[0], outs[1], outs[2]):
        assert np.array_equal(a, b1) and np.array_equal(a, b2)


@pytest.mark.gpu
def test_fused_blocks_give_the_same_bits(ctx):
    """A block's 3x3 + expanding 1x1 in one launch (k_conv3_expand, stages 1 and 2; the default) accumulates every output's K
    terms in the order of the two separate layers and rounds the mid activation to bf16 at the same place: identical logits."""
    weights, biases = seeded_parameters(0)
    ctx.cnn_set_weights(weights, biases)
    frames = synth.random_frames(5, 96, 128, seed=21)
    default = ctx.get_option("cnn_fuse")
    assert default in (1, 2)
    try:
        ctx.set_option("cnn_fuse", 0)
        plain, _ = ctx.cnn_forward(frames)
        ctx.set_option("cnn_fuse", 1)
        fused, _ = ctx.cnn_forward(frames)
        ctx.set_option("cnn_fuse", 2)                  # the 56 x 56 stage with the 3x3's input as one slab in LDS (k_slab3_expand)
        slab, _ = ctx.cnn_forward(frames)
    finally:
        ctx.set_option("cnn_fuse", default)
    assert np.isfinite(fused).all() and float(np.abs(fused).max()) > 0
    assert np.array_equal(fused, plain)
    assert np.array_equal(slab, plain)


@pytest.mark.gpu
def test_forward_in_passes_of_128_frames(ctx):
    """More frames than one pass holds: the second pass reuses every scratch buffer; frames repeat, so must the logits."""
    weights, biases = seeded_parameters(0)
    ctx.cnn_set_weights(weights, biases)
    base = synth.random_frames(3, 96, 128, seed=9)
    frames = np.concatenate([base] * 44)[:131]                       # 131 frames: 128 + 3
    logits, _ = ctx.cnn_forward(frames)
    assert logits.shape == (131, 1000)
    for i in range(131):
        assert np.array_equal(logits[i], logits[i % 3])


@pytest.mark.gpu
def test_forward_needs_weights_and_checks_counts():
    c = _lib.Context(0)
    with pytest.raises(_lib.AvdError):
        c.cnn_forward(np.zeros((1, 64, 64, 3), np.uint8))
    with pytest.raises(_lib.AvdError):
        c.cnn_set_weights(np.zeros(10, np.float32), np.zeros(10, np.float32))
    c.close()
