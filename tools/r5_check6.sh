#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
ERR=$O/check6.err
for lib in base qb2 qb8 base; do
  L=""; [ "$lib" != base ] && L="HARE_LIB=$R/hare_amd/libhare_hip_$lib.so"
  env $L timeout -k 10 200 python bench.py --scene hall_quads --steps 10 --warmup 2 --no-e2e --no-extra-configs --no-cpu-baseline 2>>$ERR | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('quads $lib', j['value'], j['ms_per_step'])" >> $O/check6.log
done
for lib in base k2dp1 k2dp3 base; do
  L=""; [ "$lib" != base ] && L="HARE_LIB=$R/hare_amd/libhare_hip_$lib.so"
  for n in 1048576 262144 4194304; do
    env $L timeout -k 10 200 python bench.py --kind octree --rays $n --steps 6 --warmup 2 --no-e2e --no-extra-configs --no-cpu-baseline 2>>$ERR | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('octree $lib n=$n', j['value'], j['ms_per_step'])" >> $O/check6.log
  done
done
timeout -k 10 500 python bench.py > $O/bench_default2.json 2> $O/bench_default2.err; echo "bench rc $?" >> $O/check6.log; wc -c $O/bench_default2.json >> $O/check6.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/check6_tests.log 2>&1; echo "suite rc $?" >> $O/check6.log
echo done >> $O/check6.log
