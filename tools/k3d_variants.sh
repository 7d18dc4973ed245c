#!/bin/bash
# Round 5: K3d (hare_kdtree_dense) over variant builds (tools/build_variants.sh "k3s2:-DHARE_K3D_STEPS=2" ...): the hall and the shoebox at 1M rays,
# the hall at 262 144 rays.  GPU box.
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"
cd "$(dirname "$0")/.."
ERRLOG="${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log"
one() { local lib=$1; shift
  local L=""; [ "$lib" != base ] && L="HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so"
  env $L timeout -k 10 200 python bench.py --kind kdtree "$@" --steps 5 --warmup 1 --no-e2e --no-extra-configs --no-cpu-baseline 2>>"$ERRLOG" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib | $* |', j['value'], j['ms_per_step'])" || { echo "$lib $* FAILED -- stderr tail:"; tail -n 8 "$ERRLOG"; }; }
for lib in base ${LIBS:-k3s2 k3s4 k3pm1 k3pm24 k3p2 k3p6 k3e12 k3e40 k3r8 k3r32 k3c128 k3w3} base; do
  one $lib --scene hall --rays 1048576
  one $lib --scene shoebox --rays 1048576
  one $lib --scene hall --rays 262144
done
