cd /tmp && export TMPDIR=/tmp
export HARE_DEV=1   # developer overrides (HARE_VOXEL_KERNEL, HARE_TICKET, ...) are only read in a process that opted in
R=$GRAFT_REPO_ROOT
for k in ${KERNELS:-persist pool}; do
  HARE_OCTREE_KERNEL=$k rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/r2_pmc_oct_$k -- python3 $R/bench.py --kind octree --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_pmc_oct_$k.log 2>&1
done
