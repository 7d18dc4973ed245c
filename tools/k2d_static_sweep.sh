#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer: K2d's static first chunk (HARE_K2P_STATIC_RAYS) and ticket size over batch sizes, hall octree 8/16.  GPU box.
cd "$(dirname "$0")/.."
one() { local label=$1 n=$2; shift 2
  env HARE_DEV=1 HARE_OCTREE_KERNEL=dense "$@" timeout -k 10 120 python bench.py --kind octree --rays $n --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label n=$n', j['value'], j['ms_per_step'])" || { echo "$label FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }; }
for n in ${RAYS:-262144 327680 393216 524288}; do
  one "default" $n
  for st in ${STATICS:-32 64 96 128}; do one "static=$st" $n HARE_K2P_STATIC_RAYS=$st; done
  one "static=64 ticket=8" $n HARE_K2P_STATIC_RAYS=64 HARE_TICKET=8
  one "static=64 ticket=32" $n HARE_K2P_STATIC_RAYS=64 HARE_TICKET=32
done
