#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer sweep: K2p with its tail kernels (K2t: a wave per ray; K2g-tail: eight lanes per ray) and hand-over rules.  GPU box.
cd "$(dirname "$0")/.."
one() {  # label, rays, env...
  local label=$1 n=$2; shift 2
  env HARE_DEV=1 HARE_OCTREE_KERNEL=persist "$@" timeout -k 10 120 python bench.py --kind octree --rays $n --steps 5 --warmup 2 --no-e2e $BARGS 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label n=$n', j['value'], j['ms_per_step'], j['kernel_only_mrays_s'], j['x_event_parity_vs_oracle'])" || { echo "$label n=$n FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
}
BARGS=""
one "K2t (16,64)" 1048576 HARE_OCTREE_TAIL=1
one "K2g-tail all at drain" 1048576 HARE_OCTREE_TAIL=2
BARGS="--no-cpu-baseline"
for mx in 64 48 32; do for pat in 0 8 32; do one "K2g-tail max=$mx patience=$pat" 1048576 HARE_OCTREE_TAIL=2 HARE_K2P_TAIL_MAX=$mx HARE_K2P_TAIL_PATIENCE=$pat; done; done
for n in 262144 524288 2097152 4194304; do one "K2t" $n HARE_OCTREE_TAIL=1; one "K2g-tail" $n HARE_OCTREE_TAIL=2; done
