#!/bin/bash
# The per-configuration bench lines of profiles/<round>_<cfg>_bench.json, re-run AFTER tools/condense_profiles.py has written profiles/traffic.json
# (the lines read their `traffic` / `issue` figures from it, keyed by the kernel-source hash), plus the default run's whole line.  GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUND=${ROUND:-r06}
O=$R/gpurun_out/${ROUND}lines
mkdir -p $O
run() { tag=$1; shift; echo "[$tag]" >> $O/progress.log; timeout -k 10 400 python3 $R/bench.py "$@" --no-extra-configs > $O/${ROUND}_${tag}_bench.json 2> $O/${tag}.err || echo "  failed" >> $O/progress.log; }
want() { [ -z "$ONLY" ] || [[ " $ONLY " == *" $1 "* ]]; }
want c2 && run c2 --steps 20 --warmup 3
want c2_4M && run c2_4M --rays 4194304 --steps 8 --warmup 2
want c2_quads && run c2_quads --scene hall_quads --steps 10 --warmup 2
want c3 && run c3 --kind octree --steps 5 --warmup 1
want c3_262k && run c3_262k --kind octree --rays 262144 --steps 8 --warmup 2
want c4shard && run c4shard --scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2
want c5 && run c5 --scene cathedral --domain 128 --bounces 8 --steps 3 --warmup 1
want kd && run kd --kind kdtree --scene shoebox --rays 1048576 --steps 5 --warmup 1
want kd_hall && run kd_hall --kind kdtree --scene hall --rays 1048576 --steps 5 --warmup 1
want c4 && run c4 --scene cathedral --domain 128 --rays 16777216 --steps 3 --warmup 1
want c5full && run c5full --scene cathedral --domain 128 --rays 8388608 --bounces 8 --steps 2 --warmup 1
if want default; then echo "[default]" >> $O/progress.log; timeout -k 10 600 python3 $R/bench.py > $O/${ROUND}_default_bench.json 2> $O/default.err || echo "  failed" >> $O/progress.log; fi
cat $O/progress.log
