#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py -x -q -m gpu > $O/check9_tests.log 2>&1; echo "tests rc $?" >> $O/check9_tests.log
bash tools/k2d_refill_by_size.sh > $O/k2d_refill_by_size.log 2>&1
timeout -k 10 300 python tools/occlusion_rate.py > $O/occlusion_rate.log 2>&1
echo done >> $O/check9_tests.log
