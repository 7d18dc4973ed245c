#!/bin/bash
# Round-5 closing run on the GPU box: the bench line of every profiled configuration (with traffic.json of the same kernel sources in place),
# the default line, the GPU suite, the smoke.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05final
mkdir -p $O
cd $R
b() { tag=$1; shift; python3 bench.py "$@" --no-extra-configs > $O/${tag}_bench.json 2> $O/${tag}_bench.err; echo "$tag rc $?" >> $O/progress.log; }
b c2 --steps 20 --warmup 3
b c2_4M --rays 4194304 --steps 8 --warmup 2
b c3 --kind octree --steps 5 --warmup 1
b c3_262k --kind octree --rays 262144 --steps 8 --warmup 2
b c4shard --scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2
b c5 --scene cathedral --domain 128 --bounces 8 --steps 3 --warmup 1
b kd --kind kdtree --scene shoebox --rays 1048576 --steps 5 --warmup 1
b kd_hall --kind kdtree --scene hall --rays 1048576 --steps 5 --warmup 1
b c2_quads --scene hall_quads --steps 10 --warmup 2
python3 bench.py > $O/default_bench.json 2> $O/default_bench.err; echo "default rc $?" >> $O/progress.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" >> $O/progress.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/progress.log
cat $O/progress.log; tail -3 $O/gpu_tests.log
