#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
cd "$(dirname "$0")/.."
one() {  # label, rays, env...
  local label=$1 n=$2; shift 2
  env HARE_DEV=1 HARE_OCTREE_KERNEL=persist "$@" timeout -k 10 120 python bench.py --kind octree --rays $n --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label n=$n', j['value'], j['ms_per_step'], j['kernel_only_mrays_s'])" || { echo "$label n=$n FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
}
for mx in 64 40 24; do for pat in 24 48 64 96 128; do one "K2g-tail max=$mx patience=$pat" 1048576 HARE_OCTREE_TAIL=2 HARE_K2P_TAIL_MAX=$mx HARE_K2P_TAIL_PATIENCE=$pat; done; done
one "K2t" 1048576 HARE_OCTREE_TAIL=1
