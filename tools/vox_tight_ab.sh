#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer A/B: the voxels' tight boxes of K1q (scene option voxel_tight) on and off: C2, 4M rays, C4 shard, the hall's and the
# cathedral's 8-bounce loops, with parity.  GPU box.
cd "$(dirname "$0")/.."
run() { local label=$1; shift
  for t in 1 0 1 0; do
    env HARE_DEV=1 HARE_VOXEL_TIGHT=$t timeout -k 10 250 python bench.py "$@" --no-e2e --no-extra-configs ${EXTRA:---no-cpu-baseline} 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label tight=$t', j['value'], j['ms_per_step'], j.get('x_event_parity_vs_oracle'))" || { echo "$label tight=$t FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
  done; }
run "C2 hall 1M" --steps 20 --warmup 3
run "hall 4M" --rays 4194304 --steps 8 --warmup 2
run "C4 shard cathedral 2M" --scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2
run "hall 8 bounces" --bounces 8 --steps 4 --warmup 1
run "C5 shard cathedral 8 bounces" --scene cathedral --domain 128 --bounces 8 --steps 3 --warmup 1
