#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
ERR=$O/check8.err
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_tight.py tests/test_gpu_ties.py tests/test_quads_at_scale.py -x -q -m gpu -k "oct or Oct or tree or tie or tight or quad" > $O/check8_tests.log 2>&1; echo "tests rc $?" >> $O/check8.log
for lib in base k2ds3 k2ds1 k2dr8 k2dr32 base; do
  L=""; [ "$lib" != base ] && L="HARE_LIB=$R/hare_amd/libhare_hip_$lib.so"
  for n in 1048576 262144 4194304; do
    env $L timeout -k 10 200 python bench.py --kind octree --rays $n --steps 6 --warmup 2 --no-e2e --no-extra-configs --no-cpu-baseline 2>>$ERR | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('octree $lib n=$n', j['value'], j['ms_per_step'], j.get('two_streams',{}).get('value') if j.get('two_streams') else '')" >> $O/check8.log
  done
done
SEEDS=250:1500 timeout -k 10 600 python tools/fuzz_parity.py > $O/fuzz2.log 2>&1; echo "fuzz rc $?" >> $O/check8.log
echo done >> $O/check8.log
