#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer sweep: octree kernels K2p / K2g / hybrid over batch sizes and hybrid shares.  GPU box.
cd "$(dirname "$0")/.."
one() {  # label, rays, extra bench args, env...
  local label=$1 n=$2; shift 2
  env HARE_DEV=1 "$@" timeout -k 10 120 python bench.py --kind octree --rays $n --steps 5 --warmup 2 --no-e2e $BARGS 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label n=$n', j['value'], j['ms_per_step'], j['kernel_only_mrays_s'], j['x_event_parity_vs_oracle'])" || { echo "$label n=$n FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
}
BARGS=""
for sh in 150 250 350 450; do one "hybrid share=$sh" 1048576 HARE_OCTREE_KERNEL=hybrid HARE_HYBRID_SHARE=$sh; done
BARGS="--no-cpu-baseline"
for n in 65536 262144 524288 2097152 4194304; do
  one persist $n HARE_OCTREE_KERNEL=persist
  one group $n HARE_OCTREE_KERNEL=group
  one "hybrid 300" $n HARE_OCTREE_KERNEL=hybrid HARE_HYBRID_SHARE=300
done
