#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer sweep of K2p's round shape (variant builds) with the K2g tail behind it.  GPU box.
cd "$(dirname "$0")/.."
for lib in "$@"; do
  for n in 1048576 4194304; do
    L=""; [ "$lib" != base ] && L="HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so"
    env HARE_DEV=1 HARE_OCTREE_KERNEL=persist $L timeout -k 10 120 python bench.py --kind octree --rays $n --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib n=$n', j['value'], j['ms_per_step'])" || { echo "$lib FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
  done
done
