#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer A/B: the subtree tight boxes of K2d / K2p (scene option octree_tight) on and off over batch sizes, with parity.  GPU box.
cd "$(dirname "$0")/.."
for t in 1 0 1 0; do
  for n in ${RAYS:-262144 1048576 4194304}; do
    env HARE_DEV=1 HARE_OCTREE_TIGHT=$t timeout -k 10 200 python bench.py --kind octree --rays $n --steps 6 --warmup 2 --no-e2e ${EXTRA:---no-cpu-baseline} 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('tight=$t n=$n', j['value'], j['ms_per_step'], j.get('x_event_parity_vs_oracle'))" || { echo "tight=$t FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
  done
done
