#!/bin/bash
# Round 5: the kd-tree and quad bench lines first, then the tests (verbose: one line per test keeps the watchdog informed)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
for sc in shoebox hall; do
  timeout -k 10 300 python bench.py --kind kdtree --scene $sc --rays 1048576 --steps 5 --warmup 1 --no-e2e > $O/kd_${sc}_dense.json 2> $O/kd_${sc}_dense.err; echo "kd $sc dense rc $?" >> $O/check2b.log
  HARE_DEV=1 HARE_KDTREE_KERNEL=simple timeout -k 10 300 python bench.py --kind kdtree --scene $sc --rays 1048576 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > $O/kd_${sc}_simple.json 2> $O/kd_${sc}_simple.err; echo "kd $sc simple rc $?" >> $O/check2b.log
done
timeout -k 10 300 python bench.py --scene hall_quads --steps 10 --warmup 2 --no-e2e --no-extra-configs > $O/c2_quads.json 2> $O/c2_quads.err; echo "quads rc $?" >> $O/check2b.log
for ord in 0 1; do
  HARE_DEV=1 HARE_VOXEL_ORDER=$ord timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-e2e --no-extra-configs --no-cpu-baseline > $O/c2_order$ord.json 2> $O/c2_order$ord.err
  HARE_DEV=1 HARE_VOXEL_ORDER=$ord timeout -k 10 300 python bench.py --scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2 --no-e2e --no-cpu-baseline > $O/c4s_order$ord.json 2> $O/c4s_order$ord.err
done
RAYS=4194304 timeout -k 10 200 python tools/bounce_open_scene.py > $O/bounce_open_scene.log 2>&1
bash tools/r5_check2.sh
