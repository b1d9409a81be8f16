"""An INDEPENDENT float64 formulation of cv2.calcOpticalFlowFarneback(prev, next, None, 0.5, 3, 15, 3, 5, 1.2, 0)
(reference site: app/analyzers/video.py:45), written from the algorithm, not from oracle/avd_oracle.c:

  * Farneback's polynomial expansion as the weighted least-squares fit it is: basis (1, x, y, x^2, y^2, xy), Gaussian
    applicability (sigma 1.2, 11 x 11), solved with numpy's general inverse of the 6 x 6 moment matrix (OpenCV hard-codes
    four entries of that inverse; if one of them were wrong this file would disagree);
  * the displacement update in its textbook matrix form: A = (A1 + A2(x + d)) / 2, db = (b1 - b2(x + d)) / 2 + A d,
    G = sum A^T A, h = sum A^T db over a 15 x 15 box, d = G^-1 h -- with x / y in natural order, 2 x 2 matrices per
    pixel, and the box sums formed DIRECTLY (a cumulative-sum box filter over an edge-replicated array; no running sums,
    no float32, no stripes);
  * OpenCV's published choices that are part of the function's definition: the pyramid (blur the full-resolution image
    with sigma = (1 / scale - 1) / 2, then resize), four scales for levels = 3 at 320 px, the five-pixel border
    attenuation, "outside the image: A = A1, db = b1 / 2", the + 1e-3 in the determinant, the x 2 bilinear up-sampling of
    the flow between scales.

Test infrastructure (tests/test_farneback_crosscheck.py): it cross-checks the single-author C oracle on well-posed clips
(agreement to ~1e-4 px is what float32 vs float64 leaves; a wrong channel order, border factor or branch shows as pixels).
"""
import numpy as np


def _resize_linear(src, dh, dw):
    """cv2.resize(..., INTER_LINEAR): centre-aligned coordinates, edge pixels replicated."""
    sh, sw = src.shape[:2]

    def taps(dn, sn):
        c = (np.arange(dn) + 0.5) * (sn / dn) - 0.5
        i0 = np.floor(c).astype(int)
        f = c - i0
        lo = i0 < 0
        i0[lo], f[lo] = 0, 0.0
        hi = i0 >= sn - 1
        i0[hi], f[hi] = sn - 1, 0.0
        return i0, np.minimum(i0 + 1, sn - 1), f

    y0, y1, fy = taps(dh, sh)
    x0, x1, fx = taps(dw, sw)
    fx = fx.reshape((1, dw) + (1,) * (src.ndim - 2))
    fy = fy.reshape((dh, 1) + (1,) * (src.ndim - 2))
    rows0, rows1 = src[y0], src[y1]
    top = rows0[:, x0] * (1 - fx) + rows0[:, x1] * fx
    bot = rows1[:, x0] * (1 - fx) + rows1[:, x1] * fx
    return top * (1 - fy) + bot * fy


def _gaussian_blur(img, ksize, sigma):
    """cv2.GaussianBlur with BORDER_REFLECT_101; sigma <= 0 with ksize 3 is OpenCV's fixed [1, 2, 1] / 4 kernel."""
    r = ksize // 2
    if sigma > 0:
        x = np.arange(-r, r + 1, dtype=np.float64)
        k = np.exp(-x * x / (2 * sigma * sigma))
        k /= k.sum()
    else:
        assert ksize == 3
        k = np.array([0.25, 0.5, 0.25])
    p = np.pad(img, r, mode="reflect")
    tmp = sum(k[i] * p[:, i:i + img.shape[1]] for i in range(ksize))
    return sum(k[i] * tmp[i:i + img.shape[0]] for i in range(ksize))


def _poly_exp(img, n=5, sigma=1.2):
    """-> bx, by, axx, ayy, axy: coefficients of the local fit f(u, v) ~ c + bx u + by v + axx u^2 + ayy v^2 + axy u v."""
    h, w = img.shape
    t = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-t * t / (2 * sigma * sigma))
    g /= g.sum()
    U, V = np.meshgrid(t, t)                                   # u = x offset (columns), v = y offset (rows)
    wgt = np.outer(g, g)
    basis = np.stack([np.ones_like(U), U, V, U * U, V * V, U * V])          # 6 x 11 x 11
    G = np.einsum("ayx,byx,yx->ab", basis, basis, wgt)
    Ginv = np.linalg.inv(G)
    p = np.pad(img, n, mode="edge")
    mom = np.zeros((6, h, w))
    for j in range(2 * n + 1):
        for i in range(2 * n + 1):
            patch = p[j:j + h, i:i + w]
            for a in range(6):
                c = basis[a, j, i] * wgt[j, i]
                if c != 0.0:
                    mom[a] += c * patch
    coef = np.einsum("ab,byx->ayx", Ginv, mom)
    return coef[1], coef[2], coef[3], coef[4], coef[5]


def _box15(a):
    """15 x 15 box SUM with replicated edges, formed directly from cumulative sums of the padded array."""
    p = np.pad(a, 7, mode="edge")
    c = np.cumsum(np.cumsum(p, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    h, w = a.shape
    return c[15:15 + h, 15:15 + w] - c[0:h, 15:15 + w] - c[15:15 + h, 0:w] + c[0:h, 0:w]


_BORDER = np.array([0.14, 0.14, 0.4472, 0.4472, 0.4472])


def _border_scale(n):
    s = np.ones(n)
    s[:5] *= _BORDER
    s[n - 5:] *= _BORDER[::-1]
    return s


def _iteration(P0, P1, flow):
    """One displacement update: flow [h, w, 2] (x, y) -> new flow."""
    bx0, by0, axx0, ayy0, axy0 = P0
    h, w = bx0.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    # the warped position is a float32 in OpenCV (float fx = x + dx): whether it falls inside the image is part of the function's
    # definition -- 319 + (-1e-8) IS 319.0f, i.e. outside -- so this one quantity is formed in float32 here as well
    px = (xx.astype(np.float32) + flow[..., 0].astype(np.float32)).astype(np.float64)
    py = (yy.astype(np.float32) + flow[..., 1].astype(np.float32)).astype(np.float64)
    x1, y1 = np.floor(px).astype(int), np.floor(py).astype(int)
    inside = (x1 >= 0) & (x1 < w - 1) & (y1 >= 0) & (y1 < h - 1)
    fx, fy = px - x1, py - y1
    xc, yc = np.clip(x1, 0, w - 2), np.clip(y1, 0, h - 2)

    def sample(c):
        v = (c[yc, xc] * (1 - fx) * (1 - fy) + c[yc, xc + 1] * fx * (1 - fy) + c[yc + 1, xc] * (1 - fx) * fy + c[yc + 1, xc + 1] * fx * fy)
        return v

    bx1, by1, axx1, ayy1, axy1 = (sample(c) for c in P1)
    # A = mean of the two quadratic forms ([[axx, axy / 2], [axy / 2, ayy]]); outside the image the second one is unknown
    axx = np.where(inside, (axx0 + axx1) / 2, axx0)
    ayy = np.where(inside, (ayy0 + ayy1) / 2, ayy0)
    axy_half = np.where(inside, (axy0 + axy1) / 4, axy0 / 2)
    dbx = np.where(inside, (bx0 - bx1) / 2, bx0 / 2) + axx * flow[..., 0] + axy_half * flow[..., 1]
    dby = np.where(inside, (by0 - by1) / 2, by0 / 2) + axy_half * flow[..., 0] + ayy * flow[..., 1]
    s = np.outer(_border_scale(h), _border_scale(w))
    axx, ayy, axy_half, dbx, dby = axx * s, ayy * s, axy_half * s, dbx * s, dby * s
    # G = sum A^T A, hvec = sum A^T db
    gxx = _box15(axx * axx + axy_half * axy_half) / 225
    gyy = _box15(ayy * ayy + axy_half * axy_half) / 225
    gxy = _box15((axx + ayy) * axy_half) / 225
    hx = _box15(axx * dbx + axy_half * dby) / 225
    hy = _box15(axy_half * dbx + ayy * dby) / 225
    idet = 1.0 / (gxx * gyy - gxy * gxy + 1e-3)
    return np.stack([(gyy * hx - gxy * hy) * idet, (gxx * hy - gxy * hx) * idet], axis=-1)


def farneback(prev, nxt, levels=3, iterations=3, pyr_scale=0.5, min_size=32):
    """prev, nxt: uint8 [h, w] -> flow float64 [h, w, 2] (x, y)."""
    h, w = prev.shape
    k, scale = 0, 1.0
    while k < levels:
        scale *= pyr_scale
        if w * scale < min_size or h * scale < min_size:
            break
        k += 1
    levels = k
    flow = None
    for k in range(levels, -1, -1):
        scale = pyr_scale ** k
        sigma = (1.0 / scale - 1) * 0.5
        ksize = max(int(np.rint(sigma * 5)) | 1, 3)             # np.rint = cvRound: half to even
        lw, lh = int(np.rint(w * scale)), int(np.rint(h * scale))
        flow = np.zeros((lh, lw, 2)) if flow is None else _resize_linear(flow, lh, lw) * (1.0 / pyr_scale)
        P = []
        for img in (prev, nxt):
            blurred = _gaussian_blur(img.astype(np.float64), ksize, sigma)
            P.append(_poly_exp(_resize_linear(blurred, lh, lw)))
        for _ in range(iterations):
            flow = _iteration(P[0], P[1], flow)
    return flow
