# Developer profile: is the vector-memory front end (TA address processing / TCP tag lookups) what bounds the traversal kernels?
export HARE_DEV=1   # developer overrides (HARE_VOXEL_KERNEL, HARE_TICKET, ...) are only read in a process that opted in
# usage: bash tools/prof_ta.sh k1p k1q [k2p k2q]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # name, env assignment, bench args
  name=$1; shift; envs=$1; shift
  i=0
  for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    echo "[$name set $i] $set" >> $R/gpurun_out/r2_ta_progress.log
    env $envs timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/r2_ta_${name}_$i -- python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_ta_${name}_$i.log 2>&1 || echo "   failed rc=$?" >> $R/gpurun_out/r2_ta_progress.log
  done
  python3 $R/tools/pmc_summary.py $R/gpurun_out/r2_ta_${name}_1 $R/gpurun_out/r2_ta_${name}_2 $R/gpurun_out/r2_ta_${name}_3 $R/gpurun_out/r2_ta_${name}_4 | grep "hare_voxel_p\|hare_octree_p\|=="
}
for k in "$@"; do
  case $k in
    k1p) run k1p HARE_VOXEL_KERNEL=persist ;;
    k1q) run k1q HARE_VOXEL_KERNEL=pool ;;
    k2p) run k2p HARE_OCTREE_KERNEL=persist --kind octree ;;
    k2q) run k2q HARE_OCTREE_KERNEL=pool --kind octree ;;
  esac
done
