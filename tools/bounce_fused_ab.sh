#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer A/B of the fused bounce kernel (hare_voxel_bounce_*) against the launch-per-cast loop, over variant builds.  GPU box.
cd "$(dirname "$0")/.."
for lib in "$@"; do
  L=""; F=1
  case "$lib" in base) ;; percast) F=0 ;; *) L="HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so" ;; esac
  for a in "--scene cathedral --domain 128 --bounces 8 --steps 3 --warmup 1" "--scene cathedral --domain 128 --bounces 8 --rays 4194304 --steps 2 --warmup 1" "--bounces 8 --steps 3 --warmup 1"; do
    env HARE_DEV=1 HARE_BOUNCE_FUSED=$F $L timeout -k 10 300 python bench.py $a --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib | $a |', j['value'], j['ms_per_step'])" || { echo "$lib FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
  done
done
