#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer A/B of two K1q builds over the bench workloads: tools/k1q_ab.sh <variant> [<variant> ...]  ("base" = hare_amd/libhare_hip.so).  GPU box.
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in "$@"; do
  L=""; [ "$lib" != base ] && L="HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so"
  for a in "--steps 30 --warmup 5" "--rays 4194304 --steps 10 --warmup 2" "--rays 262144 --steps 30 --warmup 5" "--scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2" "--scene cathedral --domain 128 --bounces 8 --steps 3 --warmup 1" "--bounces 8 --steps 3 --warmup 1"; do
    env $L timeout -k 10 200 python bench.py $a --no-e2e --no-extra-configs --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib | $a |', j['value'], j['ms_per_step'])" || { echo "$lib $a FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
  done
done
done
