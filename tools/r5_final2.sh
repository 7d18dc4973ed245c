#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/final_default_bench.json 2> $O/final_default_bench.err; echo "bench rc $?" > $O/final.log; wc -c $O/final_default_bench.json >> $O/final.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" >> $O/final.log 2>&1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/final_tests.log 2>&1; echo "suite rc $?" >> $O/final.log
echo done >> $O/final.log
