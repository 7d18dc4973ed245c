#!/bin/bash
# Round 5: the quadrilateral pre-cull (audit + quad tests + bench), the retired-ticket growth (open scene), the order pass under rocprof
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py tests/test_quads_at_scale.py -x -v -m gpu > $O/check3_tests.log 2>&1; echo "tests rc $?" >> $O/check3_tests.log
timeout -k 10 300 python bench.py --scene hall_quads --steps 10 --warmup 2 --no-e2e --no-extra-configs > $O/c2_quads_cull.json 2> $O/c2_quads_cull.err
RAYS=4194304 timeout -k 10 200 python tools/bounce_open_scene.py > $O/bounce_open_scene2.log 2>&1
cd /tmp && export TMPDIR=/tmp
HARE_DEV=1 HARE_VOXEL_ORDER=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/order_kt -- python3 $R/bench.py --scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2 --no-e2e --no-cpu-baseline --no-extra-configs > $O/order_kt.log 2>&1
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tight.py tests/test_gpu_ties.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -x -q -m gpu >> $O/check3_tests.log 2>&1; echo "suite rc $?" >> $O/check3_tests.log
echo done >> $O/check3_tests.log
