#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
ERR=$O/check7.err
for lib in base k2dp1 k2dp3 k2de12 base; do
  L=""; [ "$lib" != base ] && L="HARE_LIB=$R/hare_amd/libhare_hip_$lib.so"
  for n in 1048576 262144 4194304; do
    env $L timeout -k 10 200 python bench.py --kind octree --rays $n --steps 6 --warmup 2 --no-e2e --no-extra-configs --no-cpu-baseline 2>>$ERR | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('octree $lib n=$n', j['value'], j['ms_per_step'])" >> $O/check7.log
  done
done
timeout -k 10 1100 python -m pytest tests -q -m gpu -v > $O/check7_tests.log 2>&1; echo "suite rc $?" >> $O/check7.log
SEEDS=0:250 timeout -k 10 900 python tools/fuzz_parity.py > $O/fuzz.log 2>&1; echo "fuzz rc $?" >> $O/check7.log
echo done >> $O/check7.log
