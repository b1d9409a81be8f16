"""MI355X drop-in for reference ``app/analyzers/audio.py`` -- same name, signature, result.

``analyze(path, meta)`` keeps the reference's contract (audio.py:29-122): it never raises -- any failure (no ffmpeg, an
unreadable file, a HIP error) becomes ``{"scores": {}, "flags_audio": {"error": str(e)}, "timeline": [0.5] * tlen}``
exactly as the reference's own ``except`` does (audio.py:111-118).  The per-window spectral features run in the HIP
kernels behind the C-ABI (``avd_audio_features``); there is no CPU fallback for them."""
from __future__ import annotations

import os

from avd_hip import analyzer as _analyzer
from avd_hip import audio as _audio


def analyze(path: str, meta: dict):
    try:
        wav, sr = _audio.extract_wav_16k(path)
        with _analyzer.default_pool().borrow(int(os.getenv("AVD_DEVICE", "0"))) as ctx:
            return _audio.analyze_wave(wav, sr, ctx)
    except Exception as e:          # noqa: BLE001 -- the reference swallows everything here (audio.py:111)
        tlen = int(max(1, round(meta.get("duration") or 0.0)))
        return {"scores": {}, "flags_audio": {"error": str(e)}, "timeline": [0.5] * tlen}
