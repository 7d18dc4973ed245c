#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer: K2d's tickets and static first chunk (scene options through the environment).  GPU box.
cd "$(dirname "$0")/.."
one() { local label=$1 n=$2; shift 2
  env HARE_DEV=1 HARE_OCTREE_KERNEL=dense "$@" timeout -k 10 120 python bench.py --kind octree --rays $n --steps 6 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label n=$n', j['value'], j['ms_per_step'])" || { echo "$label FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }; }
for n in 1048576 2097152; do
  one "default" $n
  for t in 16 64 128; do one "ticket=$t" $n HARE_TICKET=$t; done
  for s in 64 96 128 192 256; do one "static=$s" $n HARE_K2P_STATIC_RAYS=$s; done
done
