mkdir -p "${GRAFT_REPO_ROOT:-.}/gpurun_out"   # the failing command's stderr is kept there (ERRLOG)
for rep in 1 2; do for lib in default nocoop now0; do
  if [ $lib = default ]; then unset HARE_LIB; else export HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so; fi
  timeout -k 10 200 python bench.py --no-extra-configs --no-e2e --steps 40 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib c2', j['value'], j['roofline']['kernel_ms'], j['x_event_parity_vs_oracle'])"
done; done
for lib in default now0; do
  if [ $lib = default ]; then unset HARE_LIB; else export HARE_LIB=$PWD/hare_amd/libhare_hip_$lib.so; fi
  timeout -k 10 200 python bench.py --bounces 8 --steps 3 --warmup 1 --no-e2e 2>>"${ERRLOG:=${GRAFT_REPO_ROOT:-.}/gpurun_out/tools_stderr.log}" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib b8 hall', j['value'], j['roofline']['per_cast_ms'])"
done
