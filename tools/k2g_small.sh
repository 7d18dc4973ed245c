#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer: K2g variant builds at the batch sizes it serves.  GPU box.
cd "$(dirname "$0")/.."
for lib in "$@"; do
  L=""; [ "$lib" != base ] && L="HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so"
  for n in 65536 262144 524288; do
    env HARE_DEV=1 HARE_OCTREE_KERNEL=group $L timeout -k 10 120 python bench.py --kind octree --rays $n --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib n=$n', j['value'], j['ms_per_step'])" || { echo "$lib FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
  done
done
