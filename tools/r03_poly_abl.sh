#!/bin/bash
# timing-only ablation builds of k_polyexp_all (results of the ablated builds are wrong)
cd $GRAFT_REPO_ROOT/ai-video-detector_amd/csrc
i=0
for flags in "" "-DAVD_POLY_ABL=4" "-DAVD_POLY_ABL=8" "-DAVD_POLY_ABL=12"; do
  i=$((i+1))
  make -B EXTRA="$flags" > /dev/null 2>&1 || { echo build failed; exit 1; }
  cd $GRAFT_REPO_ROOT
  echo "flags [$flags]"; bash tools/r03_kt.sh pabl$i | grep -E "polyexp"
  cd $GRAFT_REPO_ROOT/ai-video-detector_amd/csrc
done
make -B > /dev/null 2>&1
