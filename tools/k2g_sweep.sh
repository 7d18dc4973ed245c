#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer sweep of K2g's launch knobs (tickets, static first chunk) and of variant builds: tools/k2g_sweep.sh [lib ...]
# Prints Mrays/s and ms per step for C3 at 1M and 4M rays.  Run on the GPU box.
cd "$(dirname "$0")/.."
run() {   # label, env...
  local label=$1; shift
  for n in 1048576 4194304; do
    env HARE_DEV=1 HARE_OCTREE_KERNEL=group "$@" timeout -k 10 120 python bench.py --kind octree --rays $n --steps 5 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label n=$n', j['value'], j['ms_per_step'], j['kernel_only_mrays_s'])" || return 1
  done
}
if [ $# -eq 0 ]; then
  for t in 8 16 32 64; do run "ticket=$t" HARE_TICKET=$t || exit 1; done
  for s in 8 16 64 128; do run "ticket=16 static=$s" HARE_TICKET=16 HARE_K2P_STATIC_RAYS=$s || exit 1; done
else
  for lib in "$@"; do run "$lib" HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so HARE_TICKET=${TICKET:-16} || exit 1; done
fi
