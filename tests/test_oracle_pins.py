"""Pins for the CPU oracle (oracle/avd_oracle.c).  The reference ships no tests and cv2 is not
installable here, so the OpenCV restatement is pinned by (a) independent numpy formulations of
the integer operators, (b) numpy ITSELF for the float32 reductions the reference calls
(np.mean / np.var, video.py:47-48), (c) hand-derivable known answers, and
(d) semantic checks of the Farneback restatement (it must recover a known translation)."""
import numpy as np
import pytest

from avd_hip import synth


@pytest.fixture(scope="module")
def frame():
    return synth.random_frames(1, 135, 240, seed=7)[0]


# ---- (a) integer operators against independent numpy formulations --------------------------
def test_bgr2gray_fixed_point(oracle, frame):
    f = frame.astype(np.int64)
    want = ((f[..., 0] * 3735 + f[..., 1] * 19235 + f[..., 2] * 9798 + 16384) >> 15).astype(np.uint8)
    assert np.array_equal(oracle.bgr2gray(frame), want)
    grey = np.repeat(np.arange(256, dtype=np.uint8)[:, None, None], 3, axis=2)      # b=g=r=v -> v exactly
    assert np.array_equal(oracle.bgr2gray(grey)[:, 0], np.arange(256))


def test_laplacian_reflect101(oracle, frame):
    g = oracle.bgr2gray(frame)
    gp = np.pad(g.astype(np.int64), 1, mode="reflect")
    want = gp[:-2, 1:-1] + gp[2:, 1:-1] + gp[1:-1, :-2] + gp[1:-1, 2:] - 4 * gp[1:-1, 1:-1]
    lap = oracle.laplacian_f64(g)
    assert np.array_equal(lap, want.astype(np.float64))
    s, q = oracle.laplacian_sums(g)
    assert s == want.sum() and q == (want * want).sum()
    # exact-moment variance vs numpy's two-pass float64 .var() on the CV_64F image
    assert oracle.texture_var(s, q, g.size) == pytest.approx(lap.var(), rel=1e-13)


def test_linear_resize_numpy_mirror(oracle, frame):
    """INTER_LINEAR 8u: numpy mirror of the 11-bit fixed-point formula."""
    g = oracle.bgr2gray(frame)
    sh, sw = g.shape
    dh = dw = 320

    def axis(ssize, dsize, snap):
        scale = 1.0 / (dsize / ssize)
        f = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if snap:
            lo, hi = s < 0, s >= ssize - 1
            f[lo | hi] = 0
            s[lo] = 0
            s[hi] = ssize - 1
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return np.clip(s, 0, ssize - 1), np.clip(s + 1, 0, ssize - 1), w0, w1

    x0, x1, a0, a1 = axis(sw, dw, True)
    y0, y1, b0, b1 = axis(sh, dh, False)
    gi = g.astype(np.int64)
    h0 = gi[y0][:, x0] * a0 + gi[y0][:, x1] * a1
    h1 = gi[y1][:, x0] * a0 + gi[y1][:, x1] * a1
    want = ((((b0[:, None] * (h0 >> 4)) >> 16) + ((b1[:, None] * (h1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    assert np.array_equal(oracle.resize_linear(g, dh, dw), want)


def test_area_resize_against_float64_box_average(oracle, frame):
    """INTER_AREA (float32 accumulation, cv2 order) stays within 1 grey level of the exact
    fractional-coverage box average, and is exact on constant images."""
    g = oracle.bgr2gray(frame)
    got = oracle.resize_area(g, 32, 32).astype(np.float64)
    sh, sw = g.shape

    def cover(ssize, dsize):
        scale = ssize / dsize
        W = np.zeros((dsize, ssize))
        for d in range(dsize):
            lo, hi = d * scale, min((d + 1) * scale, ssize)
            for s in range(int(np.floor(lo)), int(np.ceil(hi))):
                W[d, s] = max(0.0, min(hi, s + 1) - max(lo, s))
            W[d] /= W[d].sum()
        return W

    want = cover(sh, 32) @ g.astype(np.float64) @ cover(sw, 32).T
    assert np.abs(got - want).max() <= 0.5 + 1e-3
    const = np.full((135, 240), 77, np.uint8)
    assert np.all(oracle.resize_area(const, 32, 32) == 77)


# ---- (b) float32 reductions against numpy itself -------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 99999, 102400, 1 << 18])
def test_np_sum_f32_bit_exact(oracle, n):
    rng = np.random.default_rng(n)
    for scale in (1.0, 1e-3, 1e4):
        a = (rng.standard_normal(n) * scale).astype(np.float32)
        assert oracle.np_sum_f32(a) == np.sum(a)


def test_flow_stats_equal_numpy_mean_var(oracle):
    rng = np.random.default_rng(3)
    for k in range(20):
        flow = (rng.standard_normal((320, 320, 2)) * rng.uniform(1e-3, 40)).astype(np.float32)
        m, v = oracle.flow_stats(flow)
        mag = np.sqrt(flow[..., 0] ** 2 + flow[..., 1] ** 2)       # video.py:46
        assert m == np.mean(mag) and v == np.var(mag)               # video.py:47-48


# ---- (c) known answers ----------------------------------------------------------------------
def test_constant_frame_known_answers(oracle):
    frames = np.full((3, 96, 160, 3), 131, np.uint8)
    small, hsh, s, q = oracle.preprocess_bgr(frames)
    assert np.all(small == 131) and np.all(hsh == 1) and np.all(s == 0) and np.all(q == 0)
    res = oracle.analyze_sampled_frames(frames, {"fps": 30.0, "duration": 1.5})
    assert res["timeline"] == [1.0, 1.0]                      # tex = 0 -> ai_susp = 1; tlen = round(1.5) = 2
    assert res["summary"]["dup_density"] == 1.0 and res["summary"]["texture_var"] == 0.0
    assert res["summary"]["flow_mean"] == 0.0                 # identical flat frames: zero flow everywhere


def test_single_bright_pixel_laplacian(oracle):
    g = np.zeros((64, 64), np.uint8)
    g[30, 31] = 200
    s, q = oracle.laplacian_sums(g)
    assert s == 0 and q == 20 * 200 * 200                    # (-4v)^2 + 4 v^2


def test_identical_frames_are_duplicates(oracle):
    clip = synth.make_clip(2, 90, 160, seed=5, dup_every=0, scene_cut=False)
    frames = np.stack([clip[0], clip[0], clip[1]])
    _, hsh, _, _ = oracle.preprocess_bgr(frames)
    assert int(np.sum(hsh[0] ^ hsh[1])) == 0


def test_timeline_shaping_quirks(oracle):
    clip = synth.make_clip(5, 64, 96, seed=9)
    long = oracle.analyze_sampled_frames(clip, {"fps": 30.0, "duration": 8.0})       # pad with last value
    assert len(long["timeline"]) == 8 and long["timeline"][5:] == [long["timeline"][4]] * 3
    short = oracle.analyze_sampled_frames(clip, {"fps": 30.0, "duration": 2.4})      # truncate to round(2.4)
    assert short["timeline"] == long["timeline"][:2]
    assert short["timeline"] is short["timeline_ai"]
    empty = oracle.analyze_sampled_frames(np.zeros((0, 64, 96, 3), np.uint8), {"duration": 2.5})
    assert empty["timeline"] == [0.5, 0.5]                                          # round(2.5) = 2 (half-even)
    assert oracle.sample_step(25) == 12 and oracle.sample_step(29.97) == 15 and oracle.sample_step(0) == 15


# ---- (d) Farneback restatement: semantics ---------------------------------------------------
def _shifted(base, dx, dy):
    from scipy import ndimage
    yy, xx = np.mgrid[0:320, 0:320].astype(np.float64)
    return np.clip(np.rint(ndimage.map_coordinates(base, [yy + 40 - dy, xx + 40 - dx], order=3)), 0, 255).astype(np.uint8)


def test_farneback_recovers_translation(oracle):
    from scipy import ndimage
    rng = np.random.default_rng(0)
    base = ndimage.gaussian_filter(rng.standard_normal((400, 400)), 4.0)
    base = (base - base.min()) / (base.max() - base.min()) * 255
    a = _shifted(base, 0, 0)
    for dx, dy in ((1.5, -0.75), (-4.0, 1.0), (8.0, -6.0)):
        flow = oracle.farneback(a, _shifted(base, dx, dy))
        inner = flow[40:-40, 40:-40]
        assert abs(np.median(inner[..., 0]) - dx) < 0.1 and abs(np.median(inner[..., 1]) - dy) < 0.1
    still = oracle.farneback(a, a)
    # identical frames: zero flow in the interior; the last row/column are 'out of range' for the
    # bilinear warp (x1 < w-1 test), so a small non-zero flow leaks in from the right/bottom border
    assert np.abs(still[40:-40, 40:-40]).max() < 1e-4 and 0 < np.abs(still).max() < 0.5


def test_farneback_constants(oracle):
    """Gaussian taps / polynomial-expansion constants: symmetry, normalisation, closed forms."""
    for ks, sigma in ((3, 0.5), (9, 1.5), (19, 3.5)):
        k = oracle.gaussian_kernel(ks, sigma).astype(np.float64)
        x = np.arange(ks) - ks // 2
        ref = np.exp(-x * x / (2 * sigma * sigma))
        assert np.allclose(k, ref / ref.sum(), rtol=0, atol=1e-7) and np.array_equal(k, k[::-1])
    assert np.array_equal(oracle.gaussian_kernel(3, 0.0), np.array([0.25, 0.5, 0.25], np.float32))
    g, xg, xxg, ig = oracle.poly_prepare(5, 1.2)
    assert abs(g.sum() - 1) < 1e-6 and np.allclose(xg, np.arange(-5, 6) * g) and np.allclose(xxg, np.arange(-5, 6) ** 2 * g)
    # invG entries from the analytic block inverse of the 6x6 moment matrix
    gd = g.astype(np.float64)
    x = np.arange(-5, 6)
    m0, m2, m4 = gd.sum() ** 2, (gd * x * x).sum() * gd.sum(), (gd * x ** 4).sum() * gd.sum()
    m22 = (gd * x * x).sum() ** 2
    G = np.zeros((6, 6))
    G[0, 0] = m0; G[1, 1] = G[2, 2] = m2; G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = m2
    G[3, 3] = G[4, 4] = m4; G[3, 4] = G[4, 3] = m22; G[5, 5] = m22
    inv = np.linalg.inv(G)
    assert np.allclose(ig, [inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]], rtol=1e-6)


def test_polyexp_reproduces_quadratic(oracle):
    """A quadratic image is reproduced exactly by its polynomial expansion (interior)."""
    yy, xx = np.mgrid[0:64, 0:64].astype(np.float64)
    img = (3.0 + 0.5 * xx - 0.25 * yy + 0.02 * xx * xx - 0.01 * yy * yy + 0.015 * xx * yy).astype(np.float32)
    R = oracle.poly_exp(img)[20, 30]            # channels: [d/dy, d/dx, yy/.., xx/.., xy] coefficients
    x, y = 30.0, 20.0
    assert R[1] == pytest.approx(0.5 + 0.04 * x + 0.015 * y, abs=2e-3)       # b_x
    assert R[0] == pytest.approx(-0.25 - 0.02 * y + 0.015 * x, abs=2e-3)     # b_y
    assert R[3] == pytest.approx(0.02, abs=2e-4) and R[2] == pytest.approx(-0.01, abs=2e-4)
    assert R[4] == pytest.approx(0.015, abs=2e-4)
