"""tools/pmc_to_json.py: counter_collection.csv passes -> the per-kernel summary bench.py reads its `traffic` figures from.
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; the guide's gfx950 correction), separate passes merged per kernel,
several instantiations of one kernel (k_fb_fast<320, UP = false / true>) averaged with their launch counts."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_pass(d, rows):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "1_counter_collection.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for r in rows:
            w.writerow(dict(zip(("Kernel_Name", "Counter_Name", "Counter_Value"), r)))


def test_summary_of_separate_passes(tmp_path):
    plain = "void (anonymous namespace)::k_fb_fast<(anonymous namespace)::FGeo<320, 3, 2, 2, 1>, false>(float const*, float*)"
    up = "void (anonymous namespace)::k_fb_fast<(anonymous namespace)::FGeo<320, 3, 2, 2, 1>, true>(float const*, float*)"
    pyr = "(anonymous namespace)::k_pyramid_all(unsigned char const*, int)"
    pre = "void (anonymous namespace)::k_preprocess_vec<8, 256>(unsigned char const*, int)"
    fetch = [(plain, "FETCH_SIZE", 1000), (plain, "FETCH_SIZE", 3000), (up, "FETCH_SIZE", 500), (pyr, "FETCH_SIZE", 100), (pre, "FETCH_SIZE", 4000)]
    write = [(plain, "WRITE_SIZE", 100), (plain, "WRITE_SIZE", 300), (up, "WRITE_SIZE", 200), (pyr, "WRITE_SIZE", 50), (pre, "WRITE_SIZE", 10)]
    _write_pass(str(tmp_path / "fetch" / "x"), fetch)
    _write_pass(str(tmp_path / "write" / "x"), write)
    out = str(tmp_path / "pmc.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_to_json.py"), out, "test", str(tmp_path / "fetch"), str(tmp_path / "write")],
                   check=True, capture_output=True)
    res = json.load(open(out))
    k = res["kernels"]
    name_plain = "k_fb_fast<FGeo<320, 3, 2, 2, 1>, false>"
    assert k[name_plain]["launches"] == 2 and k[name_plain]["FETCH_SIZE"] == 2000
    assert k[name_plain]["hbm_bytes"] == 2 * 1024 * 2000 + 1024 * 200
    up_bytes = 2 * 1024 * 500 + 1024 * 200
    want = (2 * k[name_plain]["hbm_bytes"] + 1 * up_bytes) / 3            # launch-weighted over the two instantiations
    assert abs(res["k_fb_fast<320>"]["hbm_bytes"] - want) < 1e-6 and res["k_fb_fast<320>"]["launches"] == 3
    assert res["k_preprocess_vec"]["hbm_bytes"] == 2 * 1024 * 4000 + 1024 * 10
    # per clip: the pyramid kernel runs exactly once per clip
    assert res["farneback_stage"]["clips_in_trace"] == 1
    assert abs(res["farneback_stage"]["hbm_bytes"] - (2 * k[name_plain]["hbm_bytes"] + up_bytes + 2 * 1024 * 100 + 1024 * 50)) < 1e-6
