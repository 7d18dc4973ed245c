#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer: K2d against K2g over batch sizes (the rule of launch.cpp: K2d from N rays), hall octree 8/16.  GPU box.
cd "$(dirname "$0")/.."
one() { local label=$1 n=$2; shift 2
  env HARE_DEV=1 "$@" timeout -k 10 120 python bench.py --kind octree --rays $n --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label n=$n', j['value'], j['ms_per_step'], j['roofline'] and j['roofline']['kernel'])" || { echo "$label FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }; }
for n in ${RAYS:-16384 65536 131072 196608 262144 327680 393216 524288}; do one dense $n HARE_OCTREE_KERNEL=dense; one group $n HARE_OCTREE_KERNEL=group; done
