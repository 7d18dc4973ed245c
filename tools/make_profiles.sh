# Round-3 profiles (same recipe as round 2) (run on the MI355X box): rocprofv3 kernel-trace stats and PMC passes for the bench configurations.
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), the SQ counters in a third; never combined with other trace domains.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03prof
mkdir -p $O
prof() {  # tag, bench args...
  tag=$1; shift
  echo "[$tag] kernel trace" >> $O/progress.log
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_kt -- python3 $R/bench.py "$@" --no-cpu-baseline --no-e2e --no-extra-configs > $O/${tag}_kt.log 2>&1 || echo "  failed" >> $O/progress.log
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "[$tag] $c" >> $O/progress.log
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${tag}_$c -- python3 $R/bench.py "$@" --no-cpu-baseline --no-e2e --no-extra-configs > $O/${tag}_$c.log 2>&1 || echo "  failed" >> $O/progress.log
  done
  echo "[$tag] SQ" >> $O/progress.log
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/${tag}_SQ -- python3 $R/bench.py "$@" --no-cpu-baseline --no-e2e --no-extra-configs > $O/${tag}_SQ.log 2>&1 || echo "  failed" >> $O/progress.log
  echo "[$tag] TA" >> $O/progress.log
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/${tag}_TA -- python3 $R/bench.py "$@" --no-cpu-baseline --no-e2e --no-extra-configs > $O/${tag}_TA.log 2>&1 || echo "  failed" >> $O/progress.log
  # the bench line of the same command, un-profiled, with the oracle (roofline, cpu_baseline, parity)
  python3 $R/bench.py "$@" --no-extra-configs > $O/${tag}_bench.json 2> $O/${tag}_bench.err
}
prof c2 --steps 20 --warmup 3
prof c2_4M --rays 4194304 --steps 8 --warmup 2
prof c3 --kind octree --steps 5 --warmup 1
prof c4shard --scene cathedral --domain 128 --rays 2097152 --steps 8 --warmup 2
prof c5 --scene cathedral --domain 128 --bounces 8 --steps 3 --warmup 1
python3 $R/tools/condense_profiles.py $O > $O/summary.txt 2>&1
