#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
cd "$(dirname "$0")/.."
one() {  # label, rays, env...
  local label=$1 n=$2; shift 2
  env HARE_DEV=1 "$@" timeout -k 10 120 python bench.py --kind octree --rays $n --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label n=$n', j['value'], j['ms_per_step'], j['kernel_only_mrays_s'])" || { echo "$label n=$n FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
}
for n in 524288 655360 786432 1048576 2097152 4194304; do
  one "K2p+K2t" $n HARE_OCTREE_KERNEL=persist HARE_OCTREE_TAIL=1
  one "K2p+K2g-tail(64,32)" $n HARE_OCTREE_KERNEL=persist HARE_OCTREE_TAIL=2 HARE_K2P_TAIL_MAX=64 HARE_K2P_TAIL_PATIENCE=32
  one "K2g" $n HARE_OCTREE_KERNEL=group
done
