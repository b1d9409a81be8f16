"""Audio analyzer (reference app/analyzers/audio.py).

Pins: tests/golden/audio_golden.json holds outputs of the REFERENCE's own ``analyze`` (its ffmpeg / soundfile I/O replaced by
a seeded synthetic waveform, nothing else: tests/golden/make_audio_golden.py).  CPU: the oracle restatement reproduces them
exactly, and the product's host tail reproduces the oracle's.  GPU: the HIP per-window features against numpy (float64
sums: rtol 1e-9; integer fields exact) and the whole result against the golden outputs within 1e-6 -- the north_star
tolerance for floating point is 1e-4; the spectral sums differ from numpy's pocketfft by ~1e-13 relative and the RMS is
accumulated in double where the reference uses float32."""
import json
import os
import wave

import numpy as np
import pytest

from avd_hip import audio as host_audio
from oracle import audio_oracle

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(HERE, "golden", "audio_golden.json")))["cases"]


def _wave_of(case):
    if "seed" in case:
        return audio_oracle.synth_wave(case["seconds"], case["seed"])
    return {"silence": np.zeros(16000 * 2, np.float32), "dc": np.full(16000 * 2 + 123, 0.25, np.float32)}[case["named"]]


def _close(a, b, tol):
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_close(a[k], b[k], tol) for k in a)
    if isinstance(a, list):
        return len(a) == len(b) and all(_close(x, y, tol) for x, y in zip(a, b))
    return abs(a - b) <= tol * max(1.0, abs(b))


# ---- CPU ----------------------------------------------------------------------------------------
def test_oracle_equals_the_reference_outputs(golden):
    for case in golden:
        if case.get("named") == "extract_fails":
            continue
        assert audio_oracle.analyze_wav(_wave_of(case), 16000) == case["out"], case.get("seed", case.get("named"))


def test_host_tail_equals_oracle_tail(golden):
    for case in golden:
        if "seed" not in case:
            continue
        wav = _wave_of(case)
        vals = audio_oracle.window_features(wav, 16000)
        assert host_audio.features_to_result(vals, len(wav) / 16000) == case["out"]


def test_wav_reader_and_failure_contract(tmp_path, golden):
    from app.analyzers import audio
    x = audio_oracle.synth_wave(1.0, 3)
    pcm = np.clip(np.rint(x * 32768.0), -32768, 32767).astype("<i2")
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.stack([pcm, -pcm], axis=1).tobytes())
    got, sr = host_audio.read_wav_pcm(str(p))
    assert sr == 16000 and np.array_equal(got, pcm.astype(np.float32) / np.float32(32768))      # first channel, soundfile scaling
    # no ffmpeg here: a 44.1 kHz file cannot be resampled -> the reference's failure shape (audio.py:111-118)
    q = tmp_path / "b.wav"
    with wave.open(str(q), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(44100)
        w.writeframes(pcm.tobytes())
    want = next(c for c in golden if c.get("named") == "extract_fails")["out"]
    for path in (str(q), str(tmp_path / "missing.mp4")):
        out = audio.analyze(path, {"duration": 3.4})
        assert out == want or (out["scores"] == {} and out["timeline"] == want["timeline"] and "error" in out["flags_audio"])


# ---- GPU ----------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("seconds,seed", [(6.0, 0), (9.25, 1), (0.26, 4), (1.0001, 6), (4.49, 7), (60.0, 9)])
def test_window_features_against_numpy(ctx, seconds, seed):
    wav = audio_oracle.synth_wave(seconds, seed)
    rec = ctx.audio_features(wav, 8000)
    assert len(rec) == -(-len(wav) // 8000)
    for i, r in enumerate(rec):
        seg = wav[i * 8000:(i + 1) * 8000]
        mag = np.abs(np.fft.rfft(seg * np.hanning(len(seg)))) + 1e-9
        assert r["length"] == len(seg) and r["nbins"] == len(mag)
        assert r["zero_cross"] == int(np.abs(np.diff(np.sign(seg))).sum())
        np.testing.assert_allclose(r["sumsq"], np.sum((seg ** 2).astype(np.float64)), rtol=1e-12)   # float32 squares (audio.py:44), double sum
        np.testing.assert_allclose(r["sum_mag"], np.sum(mag), rtol=1e-9)
        np.testing.assert_allclose(r["sum_log"], np.sum(np.log(mag)), rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(r["sum_fmag"], np.sum(np.linspace(0.0, 1.0, len(mag)) * mag), rtol=1e-9)
        s, idx = 0.0, 0
        cutoff = 0.85 * np.sum(mag)
        for k, m in enumerate(mag):
            s += m
            if s >= cutoff:
                idx = k
                break
        assert r["rolloff_index"] == idx


@pytest.mark.gpu
@pytest.mark.parametrize("win,n", [(1001, 2500), (999, 2997), (1002, 4000), (6, 20), (8190, 20000), (1, 5)])
def test_window_lengths_that_are_not_multiples_of_four(ctx, win, n):
    """avd.h lets the caller choose the window length (1..8192).  A length that is not a multiple of 4 has no quarter
    period in its cosine table, so the sines come from a table of their own -- one per LENGTH: the full windows and the
    shorter last window must not share it (full windows of 1001 samples used to read the 498-sample table of the last
    window, past its end)."""
    wav = (0.3 * np.random.default_rng(win).standard_normal(n)).astype(np.float32)
    rec = ctx.audio_features(wav, win)
    assert len(rec) == -(-n // win)
    for i, r in enumerate(rec):
        seg = wav[i * win:(i + 1) * win]
        mag = np.abs(np.fft.rfft(seg * np.hanning(len(seg)))) + 1e-9
        assert r["length"] == len(seg) and r["nbins"] == len(mag)
        np.testing.assert_allclose(r["sum_mag"], np.sum(mag), rtol=1e-9)
        np.testing.assert_allclose(r["sum_fmag"], np.sum(np.linspace(0.0, 1.0, len(mag)) * mag), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r["sum_log"], np.sum(np.log(mag)), rtol=1e-9, atol=1e-6)


@pytest.mark.gpu
def test_released_context_keeps_its_weights(ctx):
    """avd_release_workspace gives scratch back; uploaded weights are state and survive it (a pooled context that is
    trimmed and borrowed again must still know them)."""
    import avd_hip
    rng = np.random.default_rng(1)
    w = (rng.standard_normal((768, 768)) * 0.02).astype(np.float32)
    frames = rng.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
    with avd_hip.Context(0) as c:
        c.vit_set_weights(w, None)
        a, _ = c.vit_patch_embed(frames)
        c.analyze_frames(frames)
        c.release_workspace()
        b, _ = c.vit_patch_embed(frames)
        assert np.array_equal(a, b)
        assert len(c.analyze_frames(frames)) == 2


@pytest.mark.gpu
def test_analyze_against_the_reference_outputs(ctx, golden):
    for case in golden:
        if case.get("named") == "extract_fails":
            continue
        got = host_audio.analyze_wave(_wave_of(case), 16000, ctx)
        assert _close(got, case["out"], 1e-6), (case.get("seed", case.get("named")), got, case["out"])


@pytest.mark.gpu
def test_dropin_module_on_a_wav_file_and_in_the_pipeline(tmp_path, golden):
    from app.analyzers import audio
    from avd_hip import pipeline
    case = next(c for c in golden if c.get("seed") == 0)
    x = _wave_of(case)
    pcm = np.clip(np.rint(x.astype(np.float64) * 32768.0), -32768, 32767).astype("<i2")
    p = tmp_path / "speech.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(pcm.tobytes())
    out = audio.analyze(str(p), {"duration": 6.0})
    ref = audio_oracle.analyze_wav(pcm.astype(np.float32) / np.float32(32768), 16000)       # what the reference computes from this file
    assert "error" not in out["flags_audio"] and _close(out, ref, 1e-6)
    body = pipeline.analyze_path(str(p), {"duration": 6.0}, video_analyzer=lambda pth, m: {"timeline": [0.5] * 6, "summary": {}, "timeline_ai": [0.5] * 6})
    assert body["audio"]["scores"]["speech_ratio"] == pytest.approx(ref["scores"]["speech_ratio"]) and "audio_error" not in body["hints"]
