# developer: K1p at 1M rays over ticket size (HARE_TICKET) and static chunk (HARE_K1P_STATIC_RAYS)
export HARE_DEV=1   # developer overrides (HARE_VOXEL_KERNEL, HARE_TICKET, ...) are only read in a process that opted in
R=$GRAFT_REPO_ROOT; cd $R
for st in 128 160 192; do for tk in 16 32 48 64 96 128; do
  echo -n "static $st ticket $tk: "; HARE_K1P_STATIC_RAYS=$st HARE_TICKET=$tk RAYS=1048576 AB_TIMEOUT=120 timeout -k 10 150 python3 tools/ab_pool.py persist:default | cut -c118-200 || exit 1
done; done
