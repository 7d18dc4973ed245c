# developer A/B: pre-cull records gathered from the dense 48-byte copy (-DHARE_CULL_DENSE=1, hare_amd/libhare_hip_dense.so)
export HARE_DEV=1   # developer overrides (HARE_VOXEL_KERNEL, HARE_TICKET, ...) are only read in a process that opted in
# against the heads of the 128-byte records (default build), same session.  Parity lines first, then times.  Every leg has
# its own time limit; nothing runs after a leg that failed or timed out.
R=$GRAFT_REPO_ROOT
cd $R
D=hare_amd/libhare_hip_dense.so
O=gpurun_out/r2_dense
mkdir -p $O
RAYS=1048576,4194304 timeout -k 10 400 python3 tools/ab_pool.py persist:default persist:$D pool:default pool:$D > $O/voxel.log 2>&1 &&
timeout -k 10 400 python3 tools/ab_oct.py persist:default persist:$D > $O/octree.log 2>&1 &&
for lib in default dense; do
  if [ $lib = dense ]; then export HARE_LIB=$R/$D; else unset HARE_LIB; fi
  timeout -k 10 250 python3 bench.py --scene cathedral --domain 128 --bounces 8 --steps 5 --warmup 1 --no-e2e > $O/b8_cath_$lib.json 2> $O/b8_cath_$lib.err || exit 1
  HARE_VOXEL_KERNEL=pool timeout -k 10 250 python3 bench.py --scene hall --domain 64 --bounces 8 --steps 5 --warmup 1 --no-e2e > $O/b8_hall_pool_$lib.json 2> $O/b8_hall_pool_$lib.err || exit 1
  timeout -k 10 250 python3 bench.py --scene cathedral --domain 128 --rays 2097152 --steps 10 --warmup 2 --no-e2e > $O/c4_$lib.json 2> $O/c4_$lib.err || exit 1
done
rc=$?
cut -c1-400 $O/voxel.log $O/octree.log
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r2_dense/*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), "unreadable", e); continue
    r=j.get("roofline") or {}
    print("%-26s parity %s value %8.2f kernel_only %8.2f %s kernel_ms %s per_cast_ms %s" % (os.path.basename(f), j.get("x_event_parity_vs_oracle"), j["value"], j.get("kernel_only_mrays_s") or 0, r.get("kernel"), r.get("kernel_ms"), r.get("per_cast_ms")))
PY
exit $rc
