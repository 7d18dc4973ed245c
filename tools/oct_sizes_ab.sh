#!/bin/bash
# Developer A/B of octree builds over batch sizes: tools/oct_sizes_ab.sh <variant> ...   ("base" = hare_amd/libhare_hip.so).  GPU box.
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in "$@"; do
  L=""; [ "$lib" != base ] && L="HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so"
  for n in 65536 131072 262144 524288 1048576; do
    env $L timeout -k 10 200 python bench.py --kind octree --rays $n --steps 8 --warmup 2 --no-e2e --no-extra-configs --no-cpu-baseline 2>/dev/null |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib | $n |', j['value'], j['ms_per_step'], j['roofline'] and j['roofline']['kernel'])" || echo "$lib $n FAILED"
  done
done
done
