"""Seeded content families for the Farneback parity soak: name -> generator(rng) -> (prev, next), two uint8[320, 320]
frames as video.py:43 would hand them to cv2.calcOpticalFlowFarneback (video.py:45).  Natural-looking content (smooth and
1/f fields with translation, zoom / rotation, fades, scene cuts, letter- and pillar-boxing, saturation, blockiness, text),
degenerate content (flat, constant, steps, gradients), adversarial content (ramps, stripes, checkerboards of any cell size, with and
without overlays: singular or chaotic normal equations) and static scenes (fresh noise per frame; bit-identical frames but for a small
patch, as a screen recording has them).  Test infrastructure: used by tests/test_gpu_soak.py and tools/experiments/fb_illposed_run.py.
"""
import numpy as np

S = 320


def smooth(rng, sigma, size=S + 160):
    fy = np.fft.fftfreq(size)[:, None]
    fx = np.fft.rfftfreq(size)[None, :]
    tf = np.exp(-2.0 * (np.pi * sigma) ** 2 * (fx * fx + fy * fy))
    f = np.fft.irfft2(np.fft.rfft2(rng.standard_normal((size, size))) * tf, s=(size, size))
    return (f - f.min()) / max(float(f.max() - f.min()), 1e-9) * 255.0


def pink(rng, beta, size=S + 160):
    fy = np.fft.fftfreq(size)[:, None]
    fx = np.fft.rfftfreq(size)[None, :]
    r = np.sqrt(fx * fx + fy * fy)
    r[0, 0] = 1.0
    f = np.fft.irfft2(np.fft.rfft2(rng.standard_normal((size, size))) / r ** beta, s=(size, size))
    f = (f - f.mean()) / max(float(f.std()), 1e-9)
    return np.clip(128 + 45 * f, 0, 255)


def crop(f, oy, ox):
    return f[80 + oy:80 + oy + S, 80 + ox:80 + ox + S]


def u8(a, rng=None, noise=0):
    a = np.asarray(a, np.float64)
    if noise:
        a = a + rng.integers(-noise, noise + 1, size=a.shape)
    return np.clip(np.rint(a), 0, 255).astype(np.uint8)


def families():
    """name -> generator(rng) -> (prev, next) uint8[320,320]"""
    fam = {}

    def smooth_shift(rng):
        f = smooth(rng, rng.choice([3.0, 8.0 / 3.4, 5.0, 12.0]))
        dx, dy = rng.uniform(-6, 6, 2)
        n = int(rng.integers(0, 3))
        return u8(crop(f, 0, 0), rng, n), u8(crop(f, int(round(dy)), int(round(dx))), rng, n)
    fam["smooth_shift"] = smooth_shift

    def smooth_big_shift(rng):
        f = smooth(rng, rng.choice([3.0, 5.0, 12.0]))
        dx, dy = rng.integers(-60, 61, 2)
        return u8(crop(f, 0, 0), rng, 2), u8(crop(f, int(dy), int(dx)), rng, 2)
    fam["smooth_big_shift"] = smooth_big_shift

    def scene_cut(rng):
        return u8(crop(smooth(rng, 2.5), 0, 0), rng, 2), u8(crop(smooth(rng, rng.choice([2.5, 6.0])), 0, 0), rng, 2)
    fam["scene_cut"] = scene_cut

    def pink_shift(rng):
        f = pink(rng, rng.uniform(0.8, 1.6))
        dx, dy = rng.integers(-8, 9, 2)
        return u8(crop(f, 0, 0)), u8(crop(f, int(dy), int(dx)))
    fam["pink_shift"] = pink_shift

    def pink_cut(rng):
        return u8(crop(pink(rng, 1.2), 0, 0)), u8(crop(pink(rng, 1.0), 0, 0))
    fam["pink_cut"] = pink_cut

    def white(rng):
        return rng.integers(0, 256, (S, S), dtype=np.uint8), rng.integers(0, 256, (S, S), dtype=np.uint8)
    fam["white_noise"] = white

    def duplicate(rng):
        a = u8(crop(smooth(rng, 3.0), 0, 0), rng, 2)
        return a, a.copy()
    fam["duplicate"] = duplicate

    def letterbox(rng):
        f = smooth(rng, 3.0)
        dx, dy = rng.integers(-4, 5, 2)
        a, b = u8(crop(f, 0, 0), rng, 1), u8(crop(f, int(dy), int(dx)), rng, 1)
        bar = int(rng.integers(20, 70))
        lvl = int(rng.choice([0, 16]))
        for im in (a, b):
            im[:bar] = lvl
            im[S - bar:] = lvl
        return a, b
    fam["letterbox"] = letterbox

    def pillarbox_flatnoise(rng):
        f = pink(rng, 1.2)
        a, b = u8(crop(f, 0, 0)), u8(crop(f, 1, 2))
        bar = int(rng.integers(20, 70))
        for im in (a, b):
            im[:, :bar] = 16
            im[:, S - bar:] = 16
        return a, b
    fam["pillarbox"] = pillarbox_flatnoise

    def saturated(rng):
        f = smooth(rng, 6.0) * rng.uniform(1.5, 3.0) - rng.uniform(0, 150)
        dx, dy = rng.integers(-4, 5, 2)
        return u8(crop(f, 0, 0)), u8(crop(f, int(dy), int(dx)))
    fam["saturated"] = saturated

    def flat_tiny_noise(rng):
        v = int(rng.integers(10, 245))
        return u8(np.full((S, S), v), rng, 1), u8(np.full((S, S), v), rng, 1)
    fam["flat_noise1"] = flat_tiny_noise

    def flat_const(rng):
        return np.full((S, S), int(rng.integers(0, 256)), np.uint8), np.full((S, S), int(rng.integers(0, 256)), np.uint8)
    fam["flat_const"] = flat_const

    def step_edges(rng):
        a = np.zeros((S, S), np.uint8)
        b = np.zeros((S, S), np.uint8)
        x0 = int(rng.integers(40, 280))
        a[:, x0:] = int(rng.integers(100, 256))
        b[:, x0 + int(rng.integers(-40, 41)):] = int(rng.integers(100, 256))
        if rng.random() < 0.5:
            a, b = a.T.copy(), b.T.copy()
        return a, b
    fam["step_edges"] = step_edges

    def ramp_roll(rng):
        k = int(rng.choice([1, 2, 3]))
        ramp = ((np.add.outer(np.arange(S) * int(rng.integers(0, 2)), np.arange(S)) * k) % 256).astype(np.uint8)
        return ramp, np.roll(ramp, int(rng.integers(1, 40)), axis=1)
    fam["ramp_roll"] = ramp_roll

    def linear_gradient(rng):
        xx = np.arange(S, dtype=np.float64)
        gx, gy = rng.uniform(-0.4, 0.4, 2)
        g = 128 + gx * (xx[None, :] - 160) + gy * (xx[:, None] - 160)
        sh = rng.integers(-10, 11)
        g2 = 128 + gx * (xx[None, :] - 160 - sh) + gy * (xx[:, None] - 160)
        n = int(rng.integers(0, 2))
        return u8(g, rng, n), u8(g2, rng, n)
    fam["linear_gradient"] = linear_gradient

    def stripes(rng):
        xx = np.arange(S)
        per = rng.uniform(4, 60)
        ph = rng.uniform(0, per)
        th = rng.uniform(0, np.pi)
        u = np.cos(th) * xx[None, :] + np.sin(th) * xx[:, None]
        a = 127 + 120 * np.sin(2 * np.pi * u / per)
        b = 127 + 120 * np.sin(2 * np.pi * (u + ph) / per)
        n = int(rng.integers(0, 2))
        return u8(a, rng, n), u8(b, rng, n)
    fam["stripes"] = stripes

    def checker(rng):
        c = int(rng.choice([2, 4, 8, 16, 32]))
        yy, xx = np.mgrid[0:S + 64, 0:S + 64]
        f = (((yy // c) + (xx // c)) % 2 * int(rng.integers(60, 256))).astype(np.float64)
        dx, dy = rng.integers(0, c + 1, 2)
        return u8(f[:S, :S]), u8(f[dy:dy + S, dx:dx + S])
    fam["checker"] = checker

    def boxes(rng):
        a = np.full((S, S), int(rng.integers(60, 200)), np.float64)
        b = a.copy()
        for _ in range(int(rng.integers(1, 8))):
            y, x = rng.integers(10, 250, 2)
            hh, ww = rng.integers(6, 90, 2)
            v = int(rng.integers(0, 256))
            dy, dx = rng.integers(-6, 7, 2)
            a[y:y + hh, x:x + ww] = v
            b[max(y + dy, 0):y + dy + hh, max(x + dx, 0):x + dx + ww] = v
        return u8(a), u8(b)
    fam["boxes_on_flat"] = boxes

    def text_like(rng):
        a = np.full((S, S), 235.0)
        for _ in range(int(rng.integers(30, 200))):
            y, x = rng.integers(0, S - 12, 2)
            a[y:y + int(rng.integers(1, 4)), x:x + int(rng.integers(2, 12))] = int(rng.integers(0, 80))
        sh = int(rng.integers(0, 6))
        return u8(a), u8(np.roll(a, sh, axis=0))
    fam["text_like"] = text_like

    def fade(rng):
        f = crop(smooth(rng, 4.0), 0, 0)
        g = rng.uniform(0.2, 1.0)
        return u8(f), u8(f * g + rng.uniform(0, 40))
    fam["fade"] = fade

    def to_black(rng):
        return u8(crop(smooth(rng, 4.0), 0, 0), rng, 1), np.zeros((S, S), np.uint8)
    fam["cut_to_black"] = to_black

    def zoom_rot(rng):
        f = smooth(rng, 4.0)
        size = f.shape[0]
        yy, xx = np.mgrid[0:S, 0:S].astype(np.float64)
        ang = rng.uniform(-0.05, 0.05)
        z = rng.uniform(0.95, 1.05)
        cx = cy = 160.0
        xs = (np.cos(ang) * (xx - cx) - np.sin(ang) * (yy - cy)) * z + cx + 80
        ys = (np.sin(ang) * (xx - cx) + np.cos(ang) * (yy - cy)) * z + cy + 80
        x0 = np.clip(np.floor(xs).astype(int), 0, size - 2)
        y0 = np.clip(np.floor(ys).astype(int), 0, size - 2)
        fx, fy = xs - x0, ys - y0
        b = (f[y0, x0] * (1 - fx) * (1 - fy) + f[y0, x0 + 1] * fx * (1 - fy) + f[y0 + 1, x0] * (1 - fx) * fy + f[y0 + 1, x0 + 1] * fx * fy)
        return u8(crop(f, 0, 0), rng, 1), u8(b, rng, 1)
    fam["zoom_rot"] = zoom_rot

    def half_flat(rng):
        """textured left half, flat (saturated / black) right half, moving"""
        f = pink(rng, 1.2)
        a, b = u8(crop(f, 0, 0)), u8(crop(f, int(rng.integers(-3, 4)), int(rng.integers(-3, 4))))
        v = int(rng.choice([0, 255, 128]))
        x0 = int(rng.integers(80, 240))
        a[:, x0:] = v
        b[:, x0:] = v
        return a, b
    fam["half_flat"] = half_flat

    def blocky(rng):
        """8 x 8 block-constant image (heavy compression look), shifted"""
        small = rng.integers(0, 256, (60, 60)).astype(np.float64)
        f = np.kron(small, np.ones((8, 8)))
        dx, dy = rng.integers(0, 9, 2)
        return u8(f[:S, :S]), u8(f[dy:dy + S, dx:dx + S])
    fam["blocky"] = blocky

    def static_noise(rng):
        """a static textured scene with fresh sensor noise in every frame (no motion at all)"""
        f = crop(smooth(rng, rng.choice([2.5, 5.0])), 0, 0)
        n = int(rng.integers(1, 4))
        return u8(f, rng, n), u8(f, rng, n)
    fam["static_noise"] = static_noise

    def static_local_change(rng):
        """screen-recording-like: two bit-identical frames except for a small patch that changes (cursor, clock digit)"""
        a = u8(crop(pink(rng, 1.2), 0, 0))
        b = a.copy()
        y, x = rng.integers(20, 290, 2)
        hh, ww = rng.integers(1, 12, 2)
        b[y:y + hh, x:x + ww] = np.clip(b[y:y + hh, x:x + ww].astype(int) + int(rng.integers(-60, 61)), 0, 255).astype(np.uint8)
        return a, b
    fam["static_local_change"] = static_local_change

    def checker_any(rng):
        """checkerboards of ANY cell size (2 .. 80 px, not only powers of two), shifted by up to two cells: exactly periodic content whose
        true flow is zero wherever the frames alias -- everything cv2 computes there is the rounding residue of its running sums"""
        c = int(rng.integers(2, 81))
        yy, xx = np.mgrid[0:S + 200, 0:S + 200]
        f = (((yy // c) + (xx // c)) % 2 * int(rng.integers(60, 256))).astype(np.float64)
        dx, dy = rng.integers(0, 2 * c + 1, 2)
        return u8(f[:S, :S]), u8(f[dy:dy + S, dx:dx + S])
    fam["checker_any"] = checker_any

    def checker_overlay(rng):
        """a checkerboard that is NOT exactly periodic: a static logo, one pixel off by one grey level, or +-1 noise on top"""
        c = int(rng.choice([2, 4, 16, 32, 64]))
        yy, xx = np.mgrid[0:S + 200, 0:S + 200]
        f = (((yy // c) + (xx // c)) % 2 * int(rng.integers(60, 256))).astype(np.uint8)
        dx, dy = rng.integers(0, c + 1, 2)
        a, b = f[:S, :S].copy(), f[dy:dy + S, dx:dx + S].copy()
        kind = int(rng.integers(0, 3))
        if kind == 0:
            a[10:20, 10:30] = 77
            b[10:20, 10:30] = 77
        elif kind == 1:
            a[int(rng.integers(0, S)), int(rng.integers(0, S))] ^= 1
        else:
            a, b = u8(a, rng, 1), u8(b, rng, 1)
        return a, b
    fam["checker_overlay"] = checker_overlay

    return fam
