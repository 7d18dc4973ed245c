mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
# developer A/B of octree kernel builds through bench.py (parity against the oracle included): C3 at 1M and 4M rays
for lib in default "$@"; do
  if [ $lib = default ]; then unset HARE_LIB; else export HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so; fi
  for n in 1048576 4194304; do
    timeout -k 10 300 python bench.py --kind octree --rays $n --steps 5 --warmup 1 --no-e2e 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib n=$n', j['value'], j['roofline']['kernel'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['x_event_parity_vs_oracle'])"
  done
done
