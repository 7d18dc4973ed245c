#!/bin/bash
# Developer PMC passes for one bench command: tools/pmc_quick.sh <tag> <bench args...>   (environment: exported by the caller)
# Counters in separate passes of at most ~8 (SQ slots); per-kernel sums of the hare_* shoot kernels are printed.  GPU box only.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${PMC_DIR:-r6}/pmc_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py "${ARGS[@]}" --no-cpu-baseline --no-e2e --no-extra-configs > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $O/$name.log; return 1; }
}
ARGS=("$@")
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS &&
pass sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
pass ta TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
python3 - $O <<'P'
import csv, glob, sys, collections
O = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("hare_"): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] in ("SQ_WAVES", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"):
            seen.add(key)
for k, d in sorted(tot.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s %.4g" % (c, v))
    g = d.get
    if g("SQ_INSTS_VALU") and g("SQ_WAVE_CYCLES"):
        print("   -> active_inst_any/wave_cycles %.3f  wait_any/wave_cycles %.3f" % (g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")))
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        print("   -> valu lane utilisation (THREAD_CYCLES_VALU / ACTIVE_INST_VALU / 64) %.3f" % (g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") / 64))
P
