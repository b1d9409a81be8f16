#!/usr/bin/env python3
"""gpurun_out/ext_pmc_* -> gpurun_out/r04_pmc_extensions.json (copied to profiles/): HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB units, the
gfx950 correction of the guide) of the CNN forward pass (launch by launch, last pass in the trace) and of the GEMM."""
import csv, glob, json, re, sys
sys.path.insert(0, "ai-video-detector_amd")
from avd_hip import cnn

def rows(d):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    out = []
    for r in csv.DictReader(open(f)):
        out.append((int(r.get("Dispatch_Id", 0)), re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")), r["Counter_Name"], float(r["Counter_Value"])))
    return out

def per_dispatch(d, counter):
    acc = {}
    for disp, name, ctr, val in rows(d):
        if ctr == counter:
            acc.setdefault(disp, [name, 0.0])[1] += val
    return [acc[k] for k in sorted(acc)]

res = {"note": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB), separate --pmc passes without tracing domains"}
f = [x for x in per_dispatch("gpurun_out/ext_pmc_fetch_cnn", "FETCH_SIZE") if any(k in x[0] for k in ("k_conv", "k_slab", "k_cnn", "k_maxpool", "k_avgpool", "k_linear"))]
w = [x for x in per_dispatch("gpurun_out/ext_pmc_write_cnn", "WRITE_SIZE") if any(k in x[0] for k in ("k_conv", "k_slab", "k_cnn", "k_maxpool", "k_avgpool", "k_linear"))]
per = int(sys.argv[1]) if len(sys.argv) > 1 else 50          # launches per forward pass (50 with the fused blocks, 57 layer by layer)
out_name = sys.argv[2] if len(sys.argv) > 2 else "r04_pmc_extensions.json"
f, w = f[-per:], w[-per:]
lay = []
tot = 0.0
for i, ((n1, fv), (n2, wv)) in enumerate(zip(f, w)):
    b = 2 * 1024 * fv + 1024 * wv
    tot += b
    lay.append({"launch": i, "kernel": n1[:40], "read_bytes": 2 * 1024 * fv, "write_bytes": 1024 * wv})
res["cnn_forward_120_frames"] = {"hbm_bytes": tot, "launches": lay}
g = [x for x in per_dispatch("gpurun_out/ext_pmc_fetch_vit", "FETCH_SIZE") if "k_gemm" in x[0]]
gw = [x for x in per_dispatch("gpurun_out/ext_pmc_write_vit", "WRITE_SIZE") if "k_gemm" in x[0]]
if g and gw:
    res["patch_embed_gemm_960_frames"] = {"read_bytes": 2 * 1024 * g[-1][1], "write_bytes": 1024 * gw[-1][1], "algorithmic_read": 188160 * 768 * 2 + 768 * 768 * 2, "algorithmic_write": 188160 * 768 * 2}
json.dump(res, open("gpurun_out/" + out_name, "w"), indent=1)
print("CNN forward: %.2f GB of HBM traffic per 120 frames (%.1f MB per frame)" % (tot / 1e9, tot / 120e6))
for l in lay[:16]:
    print(l["launch"], l["kernel"], "read %.0f MB  write %.0f MB" % (l["read_bytes"] / 1e6, l["write_bytes"] / 1e6))
print(json.dumps(res.get("patch_embed_gemm_960_frames")))
