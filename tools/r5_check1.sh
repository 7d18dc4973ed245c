#!/bin/bash
# Round 5: the default bench line with own / issue fractions, its length; the bench tests and the tight-box tests on the GPU.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py > $O/bench_own.json 2> $O/bench_own.err; echo "bench rc $?" > $O/check1.log
wc -c $O/bench_own.json >> $O/check1.log
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_tight.py -x -q -m gpu >> $O/check1.log 2>&1
echo "done" >> $O/check1.log
