#!/bin/bash
# Round 5: K2d's hand-over to a tail kernel, re-swept on the round-4 FINAL K2d (the sweep of k2d_rule.sh was made on its first
# build, 493 Mrays/s, and only with "every ray a wave holds"): tail kernel (2 = K2g-tail, eight lanes per ray; 1 = K2t, a wave per ray) x
# rays a wave may still hold when it hands over x rounds it waits after its tickets ran dry.  GPU box.
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"
cd "$(dirname "$0")/.."
ERRLOG="${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log"
one() { local label=$1 n=$2; shift 2
  env HARE_DEV=1 "$@" timeout -k 10 120 python bench.py --kind octree --rays $n --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>>"$ERRLOG" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label n=$n', j['value'], j['ms_per_step'])" || { echo "$label FAILED -- stderr tail:"; tail -n 8 "$ERRLOG"; }; }
for n in ${SIZES:-1048576 262144}; do
  one "no tail" $n
  one "no tail" $n
  for t in 2 1; do for mx in 4 8 16 32 64; do for pat in 0 8 24 48; do
    one "tail=$t max=$mx patience=$pat" $n HARE_OCTREE_TAIL=$t HARE_K2P_TAIL_MAX=$mx HARE_K2P_TAIL_PATIENCE=$pat
  done; done; done
done
