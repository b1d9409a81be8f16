#!/usr/bin/env python3
"""CPU experiment (round 4): label pairs of 320 x 320 frames as well- / ill-posed for the fast Farneback level kernel and
record candidate indicators (fb_illposed_exp.c).  Test infrastructure only (uses the oracle); NOT product code.

  python tools/experiments/fb_illposed_run.py [pairs_per_family] [seed] > /tmp/illposed.jsonl
"""
import ctypes as C
import json
import os
import subprocess
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = "/tmp/fb_illposed_exp.so"
S = 320


def build():
    src = os.path.join(HERE, "fb_illposed_exp.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["gcc", "-std=gnu11", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared", "-o", SO, src, "-lm"],
                       check=True, cwd=HERE)


sys.path.insert(0, os.path.join(ROOT, "tests"))
from content_families import families  # noqa: E402


_lib = None


def run_pair(args):
    global _lib
    name, seed, with_sens = args
    if _lib is None:
        _lib = C.CDLL(SO)
        _lib.exp_illposed.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int]
    rng = np.random.default_rng(seed)
    a, b = families()[name](rng)
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    out = (C.c_double * 40)()
    _lib.exp_illposed(a.ctypes.data, b.ctypes.data, out, with_sens)
    o = list(out)
    return {"family": name, "seed": seed, "identical": bool(np.array_equal(a, b)), "D": o[0], "dmean": o[1], "dvar": o[6], "sens_pyr": o[2], "sens_flow": o[3], "mean": o[4], "var": o[5],
            "ind": [o[8 + 8 * l: 8 + 8 * l + 8] for l in range(4)]}


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    build()
    jobs = [(name, seed0 + 7919 * i + 17 * j, 1) for j, name in enumerate(families()) for i in range(per)]
    with ProcessPoolExecutor(max_workers=8) as ex:
        for r in ex.map(run_pair, jobs, chunksize=2):
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
