"""Round 5: stress of the waiting-thread settlement (avd_capi.hip, tail_help_others): four threads with one context each, a fifth thread that drives two more contexts
in turn (the single-thread in-flight loop) and keeps creating / destroying a third one with an unsettled call, clips of different length and flag density.  Every record array
must equal the one the same clip gives alone.  Usage: python tools/r05_tail_stress.py [ROUNDS]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ai-video-detector_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import avd_hip  # noqa: E402
from tests.test_gpu_fbfast import _flagged_mix  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    clips = [np.repeat(_flagged_mix(n, seed)[..., None], 3, axis=3) for n, seed in ((40, 21), (9, 22), (64, 23), (3, 24), (25, 26))]
    clips.append(np.repeat(_flagged_mix(30, 25)[::2][..., None], 3, axis=3))
    with avd_hip.Context(0) as c:
        c.set_option("tail_help", 0)
        want = [c.analyze_frames(k).tobytes() for k in clips]
    bad, done = [], [0]
    lock = threading.Lock()

    def worker(j):
        with avd_hip.Context(0) as ctx:
            rec = np.zeros(max(len(k) for k in clips), avd_hip.RECORD_DTYPE)
            for i in range(rounds):
                k = (i * 7 + j) % len(clips)
                r = rec[:len(clips[k])]
                ctx.analyze_frames_async(clips[k], r)
                if (i + j) % 5 == 0:
                    time.sleep(0.0003 * ((i + j) % 3))           # sometimes let others find this tail first
                ctx.synchronize()
                if r.tobytes() != want[k]:
                    with lock:
                        bad.append(("worker", j, i, k))
                with lock:
                    done[0] += 1

    def driver():
        a, b = avd_hip.Context(0), avd_hip.Context(0)
        ra = np.zeros(max(len(k) for k in clips), avd_hip.RECORD_DTYPE)
        rb = np.zeros_like(ra)
        rx = np.zeros_like(ra)
        try:
            for i in range(rounds):
                ka, kb, kx = i % len(clips), (i + 2) % len(clips), (i + 4) % len(clips)
                a.analyze_frames_async(clips[ka], ra[:len(clips[ka])])
                b.analyze_frames_async(clips[kb], rb[:len(clips[kb])])
                x = avd_hip.Context(0)
                x.analyze_frames_async(clips[kx], rx[:len(clips[kx])])
                if i % 2:
                    x.close()                                     # destroyed with an unsettled call
                    x = None
                (b if i % 3 else a).synchronize()
                (a if i % 3 else b).synchronize()
                if x is not None:
                    x.synchronize()
                    if rx[:len(clips[kx])].tobytes() != want[kx]:
                        with lock:
                            bad.append(("driver x", i, kx))
                    x.close()
                if ra[:len(clips[ka])].tobytes() != want[ka] or rb[:len(clips[kb])].tobytes() != want[kb]:
                    with lock:
                        bad.append(("driver", i, ka, kb))
                with lock:
                    done[0] += 1
        finally:
            a.close(); b.close()

    ths = [threading.Thread(target=worker, args=(j,)) for j in range(4)] + [threading.Thread(target=driver)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    while any(t.is_alive() for t in ths):
        time.sleep(20)
        print(f"[stress] {done[0]} calls checked, {len(bad)} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
    for t in ths:
        t.join()
    print(f"[stress] {done[0]} calls, mismatches: {bad[:10]}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
