# Profiles of a round (ROUND=r06 by default; run on the MI355X box): rocprofv3 kernel-trace stats and PMC passes for the bench configurations.
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), the SQ counters in two more, TA / TCP in a fifth; never combined with other
# trace domains.  Condensed into profiles/<round>_* by tools/condense_profiles.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r06}
O=$R/gpurun_out/${ROUND}prof
mkdir -p $O
pmc() {  # tag, pass name, counters..., then "--", bench args
  tag=$1; name=$2; shift 2
  ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  echo "[$tag] $name" >> $O/progress.log
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $O/${tag}_$name -- python3 $R/bench.py "$@" --no-cpu-baseline --no-e2e --no-extra-configs > $O/${tag}_$name.log 2>&1 || echo "  failed" >> $O/progress.log
}
prof() {  # tag, bench args...
  tag=$1; shift
  echo "[$tag] kernel trace" >> $O/progress.log
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_kt -- python3 $R/bench.py "$@" --no-cpu-baseline --no-e2e --no-extra-configs > $O/${tag}_kt.log 2>&1 || echo "  failed" >> $O/progress.log
  if [ -z "$KT_ONLY" ]; then
    pmc $tag FETCH_SIZE FETCH_SIZE -- "$@"
    pmc $tag WRITE_SIZE WRITE_SIZE -- "$@"
    pmc $tag SQ SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -- "$@"
    # lane-level VALU utilisation = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64; vector-memory instructions by direction
    pmc $tag SQ2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES -- "$@"
    pmc $tag TA TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum -- "$@"
    # wave-level VALU instructions by FP64 class: what bench.py's roofline.issue prices with the measured cost of each class (tools/valu_rate.hip)
    pmc $tag F64 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -- "$@"
  fi
  # the bench line of the same command, un-profiled, with the oracle (roofline, cpu_baseline, parity)
  python3 $R/bench.py "$@" --no-extra-configs > $O/${tag}_bench.json 2> $O/${tag}_bench.err
}
want() { [ -z "$ONLY" ] || [[ " $ONLY " == *" $1 "* ]]; }      # ONLY="c2 c3": a subset (one gpurun call holds ~3 configurations)
want c2 && prof c2 --steps 20 --warmup 3
want c2_4M && prof c2_4M --rays 4194304 --steps 8 --warmup 2
want c3 && prof c3 --kind octree --steps 5 --warmup 1
want c3_262k && prof c3_262k --kind octree --rays 262144 --steps 8 --warmup 2
want c4shard && prof c4shard --scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2
want c5 && prof c5 --scene cathedral --domain 128 --bounces 8 --steps 3 --warmup 1
want kd && prof kd --kind kdtree --scene shoebox --rays 1048576 --steps 5 --warmup 1
want kd_hall && prof kd_hall --kind kdtree --scene hall --rays 1048576 --steps 5 --warmup 1
want c2_quads && prof c2_quads --scene hall_quads --steps 10 --warmup 2
# configs 4 and 5 at FULL size on one GPU (the N = 1 point of the strong-scaling curve): their traffic / issue figures
want c4 && prof c4 --scene cathedral --domain 128 --rays 16777216 --steps 3 --warmup 1
want c5full && prof c5full --scene cathedral --domain 128 --rays 8388608 --bounces 8 --steps 2 --warmup 1
echo "done: $ONLY" >> $O/progress.log
[ -n "$NO_CONDENSE" ] || ROUND=$ROUND python3 $R/tools/condense_profiles.py $O > $O/summary.txt 2>&1
