#!/bin/bash
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# Developer: Octree.Shoot with the library's own kernel choice over batch sizes (which kernel, Mrays/s, ms).  GPU box.
cd "$(dirname "$0")/.."
python - <<'P'
import hare_amd as H
m = H.scenes.shoebox(); g = H.Octree([H.Topology(m.verts, m.nverts)], 4, 8)
print({n: g.kernel_name(n) for n in (65536, 131072, 196607, 196608, 262144, 1 << 20)})
P
for n in ${RAYS:-65536 131072 196608 262144 393216 524288 655360 786432 1048576}; do
  timeout -k 10 120 python bench.py --kind octree --rays $n --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('n=$n', j['value'], j['ms_per_step'])" || { echo "n=$n FAILED -- stderr tail:"; tail -n 8 "${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}"; }
done
