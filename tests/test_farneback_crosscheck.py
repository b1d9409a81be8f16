"""Cross-check of the C oracle's Farneback (oracle/avd_oracle.c, a restatement of OpenCV's optflowgf.cpp in float32 /
double with cv2's running sums) against an INDEPENDENT float64 numpy formulation written from the algorithm
(tests/farneback_numpy.py: least-squares polynomial expansion through a general 6 x 6 inverse, A / db in matrix form, direct
box sums).  Reference site: cv2.calcOpticalFlowFarneback(prev, small, None, 0.5, 3, 15, 3, 5, 1.2, 0), app/analyzers/video.py:45.

What it can and cannot show.  Two derivations by the same author share their reading of OpenCV's published choices (four
pyramid scales for levels = 3, the border table, the + 1e-3): those stay unpinned until a real cv2 runs
tests/golden/make_cv2_golden.py.  What it does catch is a structural error in the 950-line C restatement -- a swapped
channel, a wrong inverse entry, the wrong branch outside the image, a border factor on the wrong axis, an off-by-one in the
window -- each of which moves the flow by tenths of a pixel or more; float32 against float64 leaves 1e-5 .. 3e-4 px.
"""
import numpy as np
import pytest

from avd_hip import synth
from tests import farneback_numpy as F
from tests.content_families import families

TOL = 1e-3      # px


def _small(oracle, clip):
    return np.stack([oracle.resize_linear(oracle.bgr2gray(f), 320, 320) for f in clip])


def test_smooth_translation_and_scene_cut(oracle):
    small = _small(oracle, synth.make_clip(4, 360, 640, seed=11, dup_every=0))       # pair 1 is the scene cut (flow up to ~40 px)
    worst = 0.0
    for p in range(3):
        want = F.farneback(small[p], small[p + 1])
        got = oracle.farneback(small[p], small[p + 1])
        d = float(np.abs(got - want).max())
        worst = max(worst, d)
        assert d <= TOL, (p, d)
    # and the statistic the reference derives (video.py:46-47) from both
    m_np = float(np.sqrt((want ** 2).sum(axis=2)).mean())
    assert oracle.flow_stats(got)[0] == pytest.approx(m_np, rel=1e-5)
    print(f"[crosscheck] smooth + scene cut: max |oracle - numpy| = {worst:.3g} px")


@pytest.mark.parametrize("family,seed", [("letterbox", 3), ("smooth_big_shift", 5), ("pink_shift", 7), ("zoom_rot", 9), ("half_flat", 11),
                                          ("saturated", 13)])
def test_borders_large_motion_and_flat_regions(oracle, family, seed):
    """Letterbox bars and half-flat frames (zero normal equations, the regulariser decides), shifts of tens of pixels (warped
    positions leave the image: the 'outside' branch and the border attenuation), zoom / rotation (non-uniform flow)."""
    a, b = families()[family](np.random.default_rng(seed))
    want = F.farneback(a, b)
    got = oracle.farneback(a, b)
    d = np.abs(got - want)
    print(f"[crosscheck] {family}: max |oracle - numpy| = {d.max():.3g} px (flow up to {np.abs(want).max():.1f} px)")
    assert d.max() <= TOL, (family, float(d.max()))


def test_the_general_inverse_has_only_the_entries_opencv_uses(oracle):
    """OpenCV hard-codes ig11, ig03, ig33, ig55 of the inverse moment matrix; the numpy fit uses the full inverse.  They can only
    agree if every other entry is zero or implied by symmetry: shown here directly on the 6 x 6 matrix."""
    n, sigma = 5, 1.2
    t = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-t * t / (2 * sigma * sigma))
    g /= g.sum()
    U, V = np.meshgrid(t, t)
    basis = np.stack([np.ones_like(U), U, V, U * U, V * V, U * V])
    Ginv = np.linalg.inv(np.einsum("ayx,byx,yx->ab", basis, basis, np.outer(g, g)))
    ig = oracle.poly_prepare()[3]                                                # ig11, ig03, ig33, ig55 as the oracle computes them
    assert Ginv[1, 1] == pytest.approx(ig[0], rel=1e-6) and Ginv[2, 2] == pytest.approx(ig[0], rel=1e-6)
    assert Ginv[0, 3] == pytest.approx(ig[1], rel=1e-6) and Ginv[3, 3] == pytest.approx(ig[2], rel=1e-6)
    assert Ginv[5, 5] == pytest.approx(ig[3], rel=1e-6)
    assert abs(Ginv[3, 4]) < 1e-12 and abs(Ginv[1, 2]) < 1e-12 and abs(Ginv[1, 3]) < 1e-12       # no xx-yy coupling, no odd terms
