# developer sweep over K1q builds: tools/k1q_variants.sh name1 name2 ...  (hare_amd/libhare_hip_<name>.so, tools/build_variants.sh)
export HARE_DEV=1   # developer overrides (HARE_VOXEL_KERNEL, HARE_TICKET, ...) are only read in a process that opted in
# hall D=64 (1M, 4M rays) and cathedral D=128 (1M, 2M rays) through tools/ab_pool.py (parity + kernel time), then the 8-bounce
# loop in the cathedral (parity per bounce + per-cast times).  Nothing runs after a leg that failed or timed out.
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r2_k1qvar
mkdir -p $O
specs="pool:default"
for v in "$@"; do specs="$specs pool:hare_amd/libhare_hip_$v.so"; done
RAYS=1048576,4194304 timeout -k 10 500 python3 tools/ab_pool.py $specs > $O/hall.log 2>&1 || { cut -c1-300 $O/hall.log; exit 1; }
SCENE=cathedral DOMAIN=128 RAYS=1048576,2097152 timeout -k 10 500 python3 tools/ab_pool.py $specs > $O/cath.log 2>&1 || { cut -c1-300 $O/cath.log; exit 1; }
for lib in default "$@"; do
  if [ $lib = default ]; then unset HARE_LIB; else export HARE_LIB=$R/hare_amd/libhare_hip_$lib.so; fi
  timeout -k 10 250 python3 bench.py --scene cathedral --domain 128 --bounces 8 --steps 5 --warmup 1 --no-e2e > $O/b8_$lib.json 2> $O/b8_$lib.err || exit 1
done
echo "hall D=64"; cut -c1-44,118-330 $O/hall.log
echo "cathedral D=128"; cut -c1-44,118-330 $O/cath.log
python3 - $O <<'PY'
import json,glob,os,sys
for f in sorted(glob.glob(sys.argv[1] + "/b8_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print("%-18s parity %s value %8.2f per_cast_ms %s" % (os.path.basename(f), j.get("x_event_parity_vs_oracle"), j["value"], r.get("per_cast_ms")))
PY
