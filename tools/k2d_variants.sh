#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer A/B: K2d (hare_octree_dense) over variant builds (tools/build_variants.sh) and batch sizes.  GPU box.
#   tools/k2d_variants.sh base s2 s4 ...      (RAYS="524288 1048576 4194304" by default)
cd "$(dirname "$0")/.."
for lib in "${@:-base}"; do
  L=""; [ "$lib" != base ] && L="HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so"
  for n in ${RAYS:-524288 1048576 4194304}; do
    env HARE_DEV=1 HARE_OCTREE_KERNEL=${KERNEL:-dense} $L timeout -k 10 120 python bench.py --kind octree --rays $n --steps 6 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib n=$n', j['value'], j['ms_per_step'], j.get('x_event_parity_vs_oracle'))" || { echo "$lib FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
  done
done
