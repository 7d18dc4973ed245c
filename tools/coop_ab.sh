mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# developer A/B of the cooperative tail's switch-over rule (HARE_K1[PQ]_COOP_NOW / _MAX / _PATIENCE builds): late bounce casts in the
# cathedral (K1q and K1p forced) and the bench lines that use them
export HARE_DEV=1
O=gpurun_out/r3/coop_ab; mkdir -p $O
for lib in default "$@"; do
  if [ $lib = default ]; then unset HARE_LIB; else export HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so; fi
  echo "=== $lib"
  (timeout -k 10 120 python tools/c5_timeline.py 0; KERNEL=persist timeout -k 10 120 python tools/c5_timeline.py 0) 2>&1 | grep "==\|end p10"
  timeout -k 10 200 python bench.py --scene cathedral --domain 128 --bounces 8 --steps 3 --warmup 1 --no-e2e 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  c5 cathedral', j['value'], j['roofline']['kernel'], j['x_event_parity_vs_oracle'], j['roofline']['per_cast_ms'])"
  timeout -k 10 200 python bench.py --bounces 8 --steps 3 --warmup 1 --no-e2e 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  b8 hall     ', j['value'], j['roofline']['kernel'], j['x_event_parity_vs_oracle'], j['roofline']['per_cast_ms'])"
  timeout -k 10 200 python bench.py --no-extra-configs --no-e2e --steps 20 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  c2          ', j['value'], j['roofline']['kernel'], j['roofline']['kernel_ms'], j['x_event_parity_vs_oracle'])"
  timeout -k 10 200 python bench.py --scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2 --no-e2e 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  c4 shard    ', j['value'], j['roofline']['kernel'], j['roofline']['kernel_ms'], j['x_event_parity_vs_oracle'])"
done
