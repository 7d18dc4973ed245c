# developer A/B on a coarse-bitmap grid (cathedral, D = 128): default build vs hare_amd/libhare_hip_$1.so, same session.
export HARE_DEV=1   # developer overrides (HARE_VOXEL_KERNEL, HARE_TICKET, ...) are only read in a process that opted in
# Parity first (burst, soup, every bounce cast against the oracle), then times.  Each leg has its own limit; nothing runs after
# a leg that failed or timed out.
V=${1:?variant name}
R=$GRAFT_REPO_ROOT
cd $R
L=hare_amd/libhare_hip_$V.so
O=gpurun_out/r2_$V
mkdir -p $O
SCENE=cathedral DOMAIN=128 RAYS=1048576,2097152 timeout -k 10 500 python3 tools/ab_pool.py pool:default pool:$L persist:default persist:$L > $O/voxel.log 2>&1 || { cut -c1-300 $O/voxel.log; exit 1; }
for lib in default $V; do
  if [ $lib = default ]; then unset HARE_LIB; else export HARE_LIB=$R/$L; fi
  timeout -k 10 250 python3 bench.py --scene cathedral --domain 128 --bounces 8 --steps 5 --warmup 1 --no-e2e > $O/b8_cath_$lib.json 2> $O/b8_cath_$lib.err || exit 1
  timeout -k 10 250 python3 bench.py --scene cathedral --domain 128 --rays 2097152 --steps 10 --warmup 2 --no-e2e > $O/c4_$lib.json 2> $O/c4_$lib.err || exit 1
done
cut -c1-330 $O/voxel.log
python3 - $O <<'PY'
import json,glob,os,sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), "unreadable", e); continue
    r=j.get("roofline") or {}
    print("%-26s parity %s value %8.2f kernel_only %8.2f %s kernel_ms %s per_cast_ms %s" % (os.path.basename(f), j.get("x_event_parity_vs_oracle"), j["value"], j.get("kernel_only_mrays_s") or 0, r.get("kernel"), r.get("kernel_ms"), r.get("per_cast_ms")))
PY
