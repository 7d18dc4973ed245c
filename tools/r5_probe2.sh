#!/bin/bash
# Round 5, second GPU call: cost-homogeneous windows (physical vs through ShootIO::order), the octree timeline.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
export HARE_DEV=1
step() { echo "== $1" | tee -a $O/progress.log; }
step "window sort hall 1M" && timeout -k 10 400 python tools/c2_window_sort.py > $O/win_hall_1M.log 2>&1 &&
step "window sort cathedral 2M" && SCENE=cathedral DOMAIN=128 RAYS=2097152 WINDOWS=4096,8192,16384 timeout -k 10 400 python tools/c2_window_sort.py > $O/win_cath_2M.log 2>&1 &&
step "window sort cathedral bounce 5" && SCENE=cathedral DOMAIN=128 RAYS=1048576 BOUNCE=5 WINDOWS=2048,8192,32768 timeout -k 10 400 python tools/c2_window_sort.py > $O/win_cath_b5.log 2>&1 &&
step "timeline oct" && timeout -k 10 200 python tools/timeline_oct.py > $O/timeline_oct.log 2>&1 &&
step "done2"
