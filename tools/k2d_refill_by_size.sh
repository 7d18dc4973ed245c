#!/bin/bash
# Round 5: K2d's refill threshold (idle lanes that make a wave draw rays) over batch sizes, through the developer knobs (HARE_TUNE=steps,refill,chunk).
mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"
cd "$(dirname "$0")/.."
ERRLOG="${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log"
for n in 196608 262144 393216 524288 655360 786432 1048576; do
  for r in 16 24 32 48; do
    env HARE_DEV=1 HARE_TUNE=10,$r,128 timeout -k 10 120 python bench.py --kind octree --rays $n --steps 8 --warmup 2 --no-e2e --no-extra-configs --no-cpu-baseline 2>>"$ERRLOG" |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('n=$n refill=$r', j['value'], j['ms_per_step'])" || { echo "n=$n refill=$r FAILED"; tail -n 5 "$ERRLOG"; }
  done
done
