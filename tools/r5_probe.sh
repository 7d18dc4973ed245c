#!/bin/bash
# Round 5, first GPU call: the state of the tree on this box + the go / no-go experiments of VERDICT items 1, 2, 4.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
export HARE_DEV=1
step() { echo "== $1" | tee -a $O/progress.log; }
step "valu_rate" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && timeout -k 10 120 /tmp/valu_rate > $O/valu_rate.log 2>&1 &&
step "counters" && (rocprofv3 -L > $O/counters_list.txt 2>&1 || true) &&
step "bench default" && timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err &&
step "c2 heavy first" && timeout -k 10 300 python tools/c2_heavy_first.py > $O/c2_heavy_first.log 2>&1 &&
step "c2 timeline" && HARE_VOXEL_KERNEL=pool timeout -k 10 120 python tools/timeline_prof.py > $O/c2_timeline.log 2>&1 &&
step "bounce coherence 4" && BOUNCE=4 timeout -k 10 300 python tools/bounce_coherence_exp.py > $O/bounce_coh4.log 2>&1 &&
step "bounce coherence 6" && BOUNCE=6 timeout -k 10 300 python tools/bounce_coherence_exp.py > $O/bounce_coh6.log 2>&1 &&
step "c5 timeline" && timeout -k 10 200 python tools/c5_timeline.py > $O/c5_timeline.log 2>&1 &&
step "done"
