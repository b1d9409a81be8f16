python -m pytest tests/test_norm.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05_rerun_kt2 -o kt -- python3 $GRAFT_REPO_ROOT/tools/r05_rerun_cost.py --trace 12 > /dev/null 2>&1
