# developer: K2p static first chunk per wave (HARE_K2P_STATIC_RAYS) over batch sizes; default = the host's rule
export HARE_DEV=1   # developer overrides (HARE_VOXEL_KERNEL, HARE_TICKET, ...) are only read in a process that opted in
R=$GRAFT_REPO_ROOT; cd $R
for st in default 32 64 96 128; do
  if [ $st = default ]; then unset HARE_K2P_STATIC_RAYS; else export HARE_K2P_STATIC_RAYS=$st; fi
  echo "static $st"; RAYS=65536,262144,524288,1048576,2097152 AB_TIMEOUT=250 timeout -k 10 280 python3 tools/ab_oct.py persist:default | cut -c38-400 || exit 1
done
