#!/bin/bash
# Round 5: K3d (hare_kdtree_dense) -- its tests, then the kd-tree bench lines with K3d and with the one-ray-per-lane kernel; quads at scale.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_round5.py tests/test_quads_at_scale.py -x -v -m gpu > $O/check2_tests.log 2>&1; echo "tests rc $?" >> $O/check2_tests.log
timeout -k 10 300 python -m pytest tests/test_gpu_tight.py tests/test_gpu_parity.py tests/test_gpu_ties.py -x -q -m gpu -k "kd or kdtree or tie" >> $O/check2_tests.log 2>&1; echo "tests2 rc $?" >> $O/check2_tests.log
echo done >> $O/check2_tests.log
