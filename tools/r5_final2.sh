#!/bin/bash
# the closing check of HEAD on the GPU box: the GPU suite, the smoke, the default bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05final2
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" >> $O/progress.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/progress.log
python3 bench.py > $O/default_bench.json 2> $O/default_bench.err; echo "default rc $?" >> $O/progress.log
cat $O/progress.log; tail -2 $O/gpu_tests.log; wc -c $O/default_bench.json
