#!/bin/bash
# timing-only ablation builds of the fast level kernel, each built and timed in this box (results of the ablated builds are wrong)
cd $GRAFT_REPO_ROOT/ai-video-detector_amd/csrc
for flags in "" "-DAVD_FBF_NOGATHER" "-DAVD_FBF_NOGATHER -DAVD_FBF_NOINLOAD" "-DAVD_FBF_NOSOLVE" "-DAVD_FBF_NOGATHER -DAVD_FBF_NOINLOAD -DAVD_FBF_NOSOLVE"; do
  make -B EXTRA="$flags" > /dev/null 2>&1 || { echo build failed; exit 1; }
  cd $GRAFT_REPO_ROOT
  python bench.py --inflight 1 --steps 10 --warmup 3 --cpu-frames 0 --repeats 3 --no-pcie --no-vit --no-extras > gpurun_out/r03_abl.json 2> gpurun_out/r03_abl.err || exit 1
  python - <<PY
import json
d=json.load(open('gpurun_out/r03_abl.json'))
print('flags [$flags]', 'ms/step', d['ms_per_step'], 'level0 x3', d['stages_ms']['level0_all_iterations'], 'farneback', d['stages_ms']['farneback_and_flow_stats'])
PY
  cd $GRAFT_REPO_ROOT/ai-video-detector_amd/csrc
done
