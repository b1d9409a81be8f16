"""Oracle (and, on a GPU box, the HIP path) against stage-level vectors captured from a REAL OpenCV 4.10 by
tests/golden/make_cv2_golden.py.  The fixture does not exist yet: cv2 cannot be installed in the build image
(no wheel, no network), so these tests SKIP and parity at the OpenCV boundary stays unpinned (oracle/README.md).
The day ``tests/golden/cv2_stage_golden.npz`` is committed they pin rows A3-A7 of SURVEY.md section 8 to the
reference's actual dependency (app/analyzers/video.py:4-8,36-52)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "cv2_stage_golden.npz")

pytestmark = pytest.mark.skipif(not os.path.exists(FIXTURE), reason="no cv2-generated fixture yet (run tests/golden/make_cv2_golden.py where cv2 4.10 exists)")


@pytest.fixture(scope="module")
def golden():
    z = np.load(FIXTURE)
    info = json.loads(bytes(z["info"]).decode())
    return z, info


def _frames(info):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_cv2_golden", os.path.join(HERE, "golden", "make_cv2_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return list(gen.clips())                  # regenerated from seeds: numpy only


def test_fixture_comes_from_the_pinned_opencv(golden):
    _, info = golden
    assert info["cv2"].startswith("4.10"), info["cv2"]


def test_oracle_stage_by_stage_against_cv2(golden, oracle):
    z, info = golden
    for (h, w), frames in _frames(info):
        key = f"{h}x{w}"
        gray = np.stack([oracle.bgr2gray(f) for f in frames])
        assert np.array_equal(gray[:, ::max(1, h // 8), :], z[key + "/gray_rows"]), key
        small, hsh, s, q = oracle.preprocess_bgr(frames)
        area = np.stack([oracle.resize_area(g, 32, 32) for g in gray])
        assert np.array_equal(area, z[key + "/area32"]), key
        assert np.array_equal(hsh, z[key + "/hash"]), key
        assert np.array_equal(small, z[key + "/small320"]), key
        assert np.array_equal(s, z[key + "/lap_sum"]) and np.array_equal(q, z[key + "/lap_sumsq"]), key
        fm, fv = oracle.farneback_pairs(z[key + "/small320"])          # cv2's own 320x320 inputs: isolates Farneback
        np.testing.assert_allclose(fm, z[key + "/flow_mean"], rtol=1e-4, atol=1e-6, err_msg=key)
        np.testing.assert_allclose(fv, z[key + "/flow_var"], rtol=1e-3, atol=1e-6, err_msg=key)
        if key + "/flow" in z.files:
            for p, ref in enumerate(z[key + "/flow"]):
                got = oracle.farneback(z[key + "/small320"][p], z[key + "/small320"][p + 1])
                # bit-exactness is the aim; report the distance if the wheel took another dispatch path
                assert np.array_equal(got, ref) or float(np.abs(got - ref).max()) < 1e-3, (key, p, float(np.abs(got - ref).max()))


@pytest.mark.gpu
def test_hip_path_against_cv2(golden, ctx):
    z, info = golden
    for (h, w), frames in _frames(info):
        key = f"{h}x{w}"
        small, hsh, s, q = ctx.preprocess_bgr(frames)
        assert np.array_equal(small, z[key + "/small320"]) and np.array_equal(hsh, z[key + "/hash"]), key
        assert np.array_equal(s, z[key + "/lap_sum"]) and np.array_equal(q, z[key + "/lap_sumsq"]), key
        fm, fv = ctx.farneback_pairs(z[key + "/small320"])
        np.testing.assert_allclose(fm, z[key + "/flow_mean"], rtol=1e-4, atol=1e-6, err_msg=key)
