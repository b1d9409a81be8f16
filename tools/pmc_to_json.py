#!/usr/bin/env python3
"""rocprofv3 --pmc passes (counter_collection.csv) -> profiles/rNN_pmc.json: mean per launch and kernel.
usage: pmc_to_json.py OUT.json LABEL DIR [DIR...]   (every DIR holds one pass; counters of all passes are merged)
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request for wide coalesced reads
(MI355X_MICROARCH.md, HBM): hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE, as the guide prescribes."""
import collections
import csv
import glob
import json
import re
import sys

out_path, label, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"\(.*$", "", name)
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = {}
for name, ctr in sorted(acc.items()):
    k = {"launches": max(len(v) for v in ctr.values())}
    for c, v in sorted(ctr.items()):
        k[c] = sum(v) / len(v)
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        k["fetch_bytes_x2"] = 2 * 1024 * k["FETCH_SIZE"]
        k["write_bytes"] = 1024 * k["WRITE_SIZE"]
        k["hbm_bytes"] = k["fetch_bytes_x2"] + k["write_bytes"]
    kernels[name] = k
res = {"label": label, "note": "mean per launch; FETCH_SIZE / WRITE_SIZE in KiB, separate --pmc passes without tracing domains; "
                               "hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of the guide); SQ_* cycle counters count quad-cycles",
       "kernels": kernels}
# short names bench.py looks up
for short, pat in (("k_fb_level<320>", r"k_fb_level<320"), ("k_fb_fast<320>", r"k_fb_fast<(FGeo<)?320"), ("k_preprocess_vec", r"k_preprocess_vec"),
                   ("k_preprocess_nv12", r"k_preprocess_nv12")):
    # several instantiations of one kernel (k_fb_fast<320, UP = false / true>): the launch-weighted mean over all of them
    hit = [(name, k) for name, k in kernels.items() if re.search(pat, name) and "hbm_bytes" in k]
    if hit:
        nl = sum(k["launches"] for _, k in hit)
        res[short] = {"hbm_bytes": sum(k["hbm_bytes"] * k["launches"] for _, k in hit) / nl, "kernel": " + ".join(n for n, _ in hit),
                      "launches": nl}
        # fractions of the launch the vector ALUs / the texture addresser were busy: SQ_ACTIVE_INST_VALU counts quad-cycles summed over the
        # chip's 1024 SIMDs, TA_BUSY_avr cycles averaged over the TA instances, GRBM_GUI_ACTIVE the cycles of the launch SUMMED over the
        # chip's 8 XCDs (each has its own GRBM: 2.66 M for a 141-us launch = 8 x 333 k cycles at ~2.36 GHz)
        def wmean(c):
            v = [(k[c], k["launches"]) for _, k in hit if c in k]
            return sum(a * b for a, b in v) / sum(b for _, b in v) if v else None
        act, valu, ta = wmean("GRBM_GUI_ACTIVE"), wmean("SQ_ACTIVE_INST_VALU"), wmean("TA_BUSY_avr")
        if act:
            act /= 8.0
            res[short]["gui_active_cycles"] = act
            if valu is not None:
                res[short]["valu_issue_frac"] = round(valu * 4 / 1024 / act, 4)
            if ta is not None:
                res[short]["ta_busy_frac"] = round(ta / act, 4)
fb = [k for n, k in kernels.items() if re.search(r"k_fb_level|k_fb_fast|k_pyramid|k_polyexp|k_flow_up|k_stats|k_uv|k_hscan", n) and "hbm_bytes" in k]
if fb:
    # per clip: launches per clip = launches / clips in the trace; every kernel above is launched a fixed number of times per clip
    # (the pyramid kernel exactly once)
    clips = min([k["launches"] for n, k in kernels.items() if re.search(r"k_pyramid_all", n)] or [1]) or 1
    res["farneback_stage"] = {"hbm_bytes": sum(k["hbm_bytes"] * k["launches"] for k in fb) / clips, "clips_in_trace": clips}
    res["whole_clip"] = {"hbm_bytes": sum(k["hbm_bytes"] * k["launches"] for k in kernels.values() if "hbm_bytes" in k) / clips}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
for n, k in kernels.items():
    if re.search(r"k_fb_fast|k_fb_level|k_polyexp|k_preprocess", n):
        print(n, json.dumps({c: (round(v) if abs(v) > 100 else v) for c, v in k.items()}))
