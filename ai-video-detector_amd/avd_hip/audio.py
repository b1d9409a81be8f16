"""Host side of the audio analyzer: waveform loading and the scalar tail of reference
``app/analyzers/audio.py``.

The per-window loop (audio.py:40-61: RMS, zero crossings, Hann window, rFFT magnitudes, flatness / roll-off / centroid) runs
in the HIP kernels behind ``avd_audio_features`` for all windows of the file at once; what is left is O(windows) float64
numpy -- percentile, variances, tts_like, timeline (audio.py:63-110) -- kept on the host so that the reductions are the
very calls the reference makes.  There is no CPU fallback for the spectral part."""
from __future__ import annotations

import os
import shutil
import subprocess
import tempfile
import wave

import numpy as np

SAMPLE_RATE = 16000          # audio.py:10: ffmpeg -ac 1 -ar 16000


def read_wav_pcm(path: str):
    """-> (float32 samples of channel 0 scaled like soundfile's dtype='float32', sample rate); PCM 8/16/32-bit WAV."""
    with wave.open(path, "rb") as w:
        nch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        a = np.frombuffer(raw, "<i2").astype(np.float32) / np.float32(32768.0)
    elif width == 4:
        a = (np.frombuffer(raw, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif width == 1:
        a = (np.frombuffer(raw, np.uint8).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    else:
        raise RuntimeError("soundfile_read_failed")
    if nch > 1:
        a = a.reshape(-1, nch)[:, 0]          # audio.py:34: first channel
    return np.ascontiguousarray(a), int(sr)


def extract_wav_16k(path: str):
    """audio.py:7-20: ffmpeg -> 16 kHz mono wav -> float32.  Without ffmpeg (this image has none) a file that already IS a
    16 kHz PCM wav is read directly; anything else fails the way the reference fails when ffmpeg does."""
    if shutil.which("ffmpeg"):
        tmp = tempfile.NamedTemporaryFile(delete=False, suffix=".wav")
        tmp.close()
        try:
            proc = subprocess.run(["ffmpeg", "-y", "-i", path, "-ac", "1", "-ar", str(SAMPLE_RATE), "-f", "wav", tmp.name],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if proc.returncode != 0:
                raise RuntimeError("ffmpeg_convert_failed")
            try:
                return read_wav_pcm(tmp.name)
            except Exception:
                raise RuntimeError("soundfile_read_failed")
        finally:
            try:
                os.unlink(tmp.name)
            except OSError:
                pass
    try:
        wav, sr = read_wav_pcm(path)
    except Exception:
        raise RuntimeError("ffmpeg_convert_failed")
    if sr != SAMPLE_RATE:
        raise RuntimeError("ffmpeg_convert_failed")          # resampling is ffmpeg's job
    return wav, sr


def window_values(rec: np.ndarray) -> dict:
    """avd_audio_window records -> the five per-window lists of audio.py:38-61 (python floats)."""
    n = rec["length"].astype(np.float64)
    nb = rec["nbins"].astype(np.float64)
    rms = np.sqrt(rec["sumsq"] / n)
    # np.mean over a float32 array divides in float32 (audio.py:45), then / 2.0
    with np.errstate(invalid="ignore", divide="ignore"):
        zcr = (rec["zero_cross"].astype(np.float32) / (rec["length"] - 1).astype(np.float32)).astype(np.float64) / 2.0
    flat = np.exp(rec["sum_log"] / nb) / (rec["sum_mag"] / nb)
    roll = rec["rolloff_index"].astype(np.float64) / np.maximum(1.0, nb)
    sc = rec["sum_fmag"] / rec["sum_mag"]
    return {k: [float(x) for x in v] for k, v in (("rms", rms), ("zcr", zcr), ("flat", flat), ("roll", roll), ("sc", sc))}


def _norm01(x):
    x = np.asarray(x, dtype=float)
    if x.size == 0:
        return np.zeros(1)
    lo, hi = float(np.min(x)), float(np.max(x))
    return (x - lo) / (hi - lo + 1e-9)


def features_to_result(values: dict, dur: float) -> dict:
    """The tail of audio.py (63-110): same numpy calls, same order."""
    arr = {k: (np.array(v) if v else np.zeros(1)) for k, v in values.items()}
    rms, zcr, flat, roll, sc = arr["rms"], arr["zcr"], arr["flat"], arr["roll"], arr["sc"]
    speech_ratio = float(np.mean(rms >= np.percentile(rms, 60)))
    flat_mean = float(np.mean(flat))
    sc_var, roll_var, zcr_var = float(np.var(sc)), float(np.var(roll)), float(np.var(zcr))
    tts_base = 0.7 * flat_mean + 0.15 * (1.0 / (1e-6 + zcr_var)) + 0.15 * (1.0 / (1e-6 + roll_var))
    tts_like = float(np.clip(tts_base * (1.0 / (1.0 + 5.0 * (sc_var + roll_var + zcr_var))), 0.0, 1.0))
    if sc_var + roll_var + zcr_var > 0.005:                  # "cap del TTS" when the variability is not negligible
        tts_like = float(min(tts_like, 0.90))
    dzcr = np.diff(np.concatenate([[zcr[0]], zcr]))
    droll = np.diff(np.concatenate([[roll[0]], roll]))
    line = np.clip(0.5 * _norm01(flat) + 0.3 * (1.0 - _norm01(dzcr ** 2)) + 0.2 * (1.0 - _norm01(np.abs(droll))), 0.0, 1.0).tolist()
    tlen = int(max(1, round(dur)))
    line = line + [line[-1] if line else 0.5] * (tlen - len(line)) if len(line) < tlen else line[:tlen]
    return {"scores": {"speech_ratio": speech_ratio, "tts_like": tts_like},
            "flags_audio": {"speech_ratio": speech_ratio, "tts_like": tts_like, "rms_var": float(np.var(rms)),
                            "zcr_var": zcr_var, "roll_var": roll_var, "sc_var": sc_var},
            "timeline": line}


def analyze_wave(wav: np.ndarray, sr: int, ctx) -> dict:
    """audio.py:33-110 for a decoded waveform, spectral features on the GPU."""
    if wav.ndim > 1:
        wav = wav[:, 0]
    dur = len(wav) / sr if sr > 0 else 0.0
    win = max(1, int(sr * 0.5)) if sr else 1
    rec = ctx.audio_features(np.ascontiguousarray(wav, np.float32), win)
    return features_to_result(window_values(rec), dur)
