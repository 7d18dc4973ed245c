#!/bin/bash
# Developer: per-kernel durations (rocprofv3 --kernel-trace --stats) of one bench command: tools/kt_quick.sh <tag> <bench args...>  (env exported by the caller)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4/kt_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py "$@" --no-cpu-baseline --no-e2e --no-extra-configs > $O/run.log 2>&1 || { echo "failed"; tail -3 $O/run.log; exit 1; }
python3 - $O <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("hare_"):
            print("  %-28s calls %4s  avg %10.1f us  total %6.2f %%" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
P
