#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer A/B: variant builds of the voxel tight-box paths (main walk only / + wide walk / + cooperative tail) over the voxel workloads.  GPU box.
cd "$(dirname "$0")/.."
run() { local label=$1; shift
  for lib in "${LIBS[@]}"; do
    L=""; [ "$lib" != base ] && L="HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so"
    env HARE_DEV=1 $L timeout -k 10 250 python bench.py "$@" --no-e2e --no-extra-configs --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$label $lib', j['value'], j['ms_per_step'])" || { echo "$label $lib FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
  done; }
LIBS=("$@")
run "C2 hall 1M" --steps 20 --warmup 3
run "C4 shard cathedral 2M" --scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2
run "hall 8 bounces" --bounces 8 --steps 4 --warmup 1
run "C5 shard cathedral 8 bounces" --scene cathedral --domain 128 --bounces 8 --steps 3 --warmup 1
