#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -m gpu > $O/check5_tests.log 2>&1; echo "tests rc $?" >> $O/check5_tests.log
LIBS="k3p1 k3p3 k3s2 k3s5 k3e8 k3e48 k3c32 k3r24" bash tools/k3d_variants.sh > $O/k3d_variants2.log 2>&1
for t in 8 32; do
  HARE_DEV=1 HARE_TICKET=$t timeout -k 10 200 python bench.py --kind kdtree --scene hall --rays 1048576 --steps 5 --warmup 1 --no-e2e --no-extra-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('ticket $t hall 1M', j['value'], j['ms_per_step'])" >> $O/k3d_variants2.log
done
for sr in 32 64 128; do
  HARE_DEV=1 HARE_K2P_STATIC_RAYS=$sr timeout -k 10 200 python bench.py --kind kdtree --scene hall --rays 1048576 --steps 5 --warmup 1 --no-e2e --no-extra-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('static $sr hall 1M', j['value'], j['ms_per_step'])" >> $O/k3d_variants2.log
done
bash tools/bounce_sort_pmc.sh > $O/bounce_sort_pmc.log 2>&1
echo done >> $O/check5_tests.log
