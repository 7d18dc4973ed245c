#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
for ord in 0 1 2; do
  HARE_DEV=1 HARE_VOXEL_ORDER=$ord timeout -k 10 300 python bench.py --scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2 --no-e2e --no-cpu-baseline --no-extra-configs 2>>$O/check4.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c4 shard order $ord', j['value'], j['ms_per_step'])" >> $O/check4.log
  HARE_DEV=1 HARE_VOXEL_ORDER=$ord timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-e2e --no-cpu-baseline --no-extra-configs 2>>$O/check4.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c2 order $ord', j['value'], j['ms_per_step'], j.get('two_streams'))" >> $O/check4.log
  HARE_DEV=1 HARE_VOXEL_ORDER=$ord timeout -k 10 300 python bench.py --scene cathedral --domain 128 --rays 16777216 --steps 4 --warmup 1 --no-e2e --no-cpu-baseline --no-extra-configs 2>>$O/check4.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c4 16M order $ord', j['value'], j['ms_per_step'])" >> $O/check4.log
done
timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "cost_order" >> $O/check4.log 2>&1
bash tools/k3d_variants.sh > $O/k3d_variants.log 2>&1
echo done >> $O/check4.log
