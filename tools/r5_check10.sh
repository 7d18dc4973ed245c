#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -m gpu > $O/check10_tests.log 2>&1; echo "tests rc $?" >> $O/check10_tests.log
SEEDS=0:300 timeout -k 10 400 python tools/fuzz_parity.py > $O/check10_fuzz.log 2>&1; echo "fuzz rc $?" >> $O/check10_fuzz.log
timeout -k 10 200 python tools/occlusion_rate.py > $O/occlusion_rate.log 2>&1
tail -3 $O/check10_tests.log; tail -3 $O/check10_fuzz.log
