#!/usr/bin/env python3
"""Golden vectors for the audio analyzer from the REFERENCE's own code (build container only).

    python tests/golden/make_audio_golden.py        # writes tests/golden/audio_golden.json

reference app/analyzers/audio.py cannot run as it stands: it imports ``soundfile`` (absent) and shells out to ``ffmpeg``
(absent) to obtain the 16 kHz mono waveform (audio.py:7-20).  Everything AFTER that point -- the per-window features and
the scalar tail, audio.py:33-110 -- is plain numpy and is exactly what the build accelerates.  So this script loads the
reference module BY FILE PATH with
  * a placeholder ``soundfile`` entry in sys.modules (an empty module object, only so that the import statement succeeds;
    no function of it is ever called), and
  * ``_extract_wav_16k`` replaced by a function that returns a seeded synthetic waveform (oracle.audio_oracle.synth_wave),
and calls the reference's own ``analyze``.  No arithmetic of the reference is replaced.  Inputs are stored as seeds
(the generator is deterministic numpy), outputs verbatim.  Nothing of the reference's source is copied."""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AVD_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
from oracle import audio_oracle  # noqa: E402

CASES = [(0, 6.0), (1, 9.25), (2, 3.0), (3, 0.5), (4, 0.26), (5, 12.0), (6, 1.0001), (7, 4.49)]   # (seed, seconds)


def main():
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))       # placeholder, never called
    spec = importlib.util.spec_from_file_location("ref_audio", os.path.join(REF, "app", "analyzers", "audio.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = []
    for seed, seconds in CASES:
        wav = audio_oracle.synth_wave(seconds, seed)
        ref._extract_wav_16k = lambda path, w=wav: (None, w, 16000)           # I/O only
        res = ref.analyze("synthetic.wav", {"duration": seconds})
        assert "error" not in res["flags_audio"], res
        out.append({"seed": seed, "seconds": seconds, "samples": int(len(wav)), "out": res})
    # silence and a constant: degenerate windows (log of the 1e-9 floor, zero variance)
    for name, wav in (("silence", np.zeros(16000 * 2, np.float32)), ("dc", np.full(16000 * 2 + 123, 0.25, np.float32))):
        ref._extract_wav_16k = lambda path, w=wav: (None, w, 16000)
        out.append({"named": name, "samples": int(len(wav)), "out": ref.analyze("synthetic.wav", {"duration": 2.0})})
    ref._extract_wav_16k = lambda path: (_ for _ in ()).throw(RuntimeError("ffmpeg_convert_failed"))
    out.append({"named": "extract_fails", "meta_duration": 3.4, "out": ref.analyze("x.mp4", {"duration": 3.4})})
    path = os.path.join(HERE, "audio_golden.json")
    json.dump({"generator": "tests/golden/make_audio_golden.py", "numpy": np.__version__, "cases": out}, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
