#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer: K2d's hand-over rule to K2g-tail and the crossover with K2g.  GPU box.
cd "$(dirname "$0")/.."
one() { local label=$1 n=$2; shift 2
  env HARE_DEV=1 "$@" timeout -k 10 120 python bench.py --kind octree --rays $n --steps 6 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label n=$n', j['value'], j['ms_per_step'], j['roofline'] and j['roofline']['kernel'])" || { echo "$label FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }; }
for pat in 8 16 32 48 64 96; do one "dense tail patience=$pat" 1048576 HARE_OCTREE_KERNEL=dense HARE_K2P_TAIL_MAX=64 HARE_K2P_TAIL_PATIENCE=$pat; done
one "dense K2t" 1048576 HARE_OCTREE_KERNEL=dense HARE_OCTREE_TAIL=1
one "dense no tail" 1048576 HARE_OCTREE_KERNEL=dense HARE_OCTREE_TAIL=0
for n in 131072 262144 393216 524288 655360 786432; do one dense $n HARE_OCTREE_KERNEL=dense; one group $n HARE_OCTREE_KERNEL=group; done
