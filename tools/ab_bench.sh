#!/bin/bash
export HARE_DEV=1   # developer overrides (HARE_VOXEL_KERNEL, HARE_TICKET, ...) are only read in a process that opted in
# Developer A/B of the two voxel kernels through bench.py (parity against the oracle included): C2, C4 shard, C5
for k in persist pool; do
  for cfg in "--steps 20 --warmup 3" "--rays 2097152 --steps 10 --warmup 2" "--rays 4194304 --steps 6 --warmup 2" "--scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2" "--scene cathedral --domain 128 --bounces 8 --steps 4 --warmup 1"; do
    echo "== $k $cfg"
    HARE_VOXEL_KERNEL=$k timeout -k 10 400 python bench.py $cfg | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('   value %.1f  kernel-only %.1f  parity %s  frac %s  kernel %s %.4f ms' % (j['value'], j['kernel_only_mrays_s'], j['x_event_parity_vs_oracle'], r and r['frac'], r and r['kernel'], r and r['kernel_ms']))
"
  done
done
