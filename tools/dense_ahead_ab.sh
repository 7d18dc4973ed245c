#!/bin/bash
# Developer A/B: the dense windows as a pipeline (HARE_K2D_AHEAD; K2d and K3d) against the build without: tools/dense_ahead_ab.sh base a0.  GPU box.
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in "$@"; do
  L=""; [ "$lib" != base ] && L="HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so"
  for cfg in "--kind octree --rays 1048576" "--kind octree --rays 4194304" "--kind octree --rays 262144" "--kind octree --rays 131072" "--kind kdtree --scene hall --rays 1048576" "--kind kdtree --scene shoebox --rays 1048576" "--kind octree --scene cathedral --rays 1048576"; do
    env $L timeout -k 10 200 python bench.py $cfg --steps 6 --warmup 2 --no-e2e --no-extra-configs --no-cpu-baseline 2>/dev/null |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib | $cfg |', j['value'], j['ms_per_step'], j['x_event_parity_vs_oracle'])" || echo "$lib $cfg FAILED"
  done
done
done
