#!/bin/bash
# Round 5: PMC of a late bounce cast, rays as the loop leaves them vs physically sorted by (4^3-voxel block of the origin,
# direction octant).  Separate --pmc passes; the last 5 dispatches of the voxel kernel per run are averaged by the summary below.  GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5/bounce_sort_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for order in given sorted; do
  export ORDER=$order
  python3 $R/tools/bounce_sort_pmc.py > $O/${order}_time.log 2>&1
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/${order}_p$i -- python3 $R/tools/bounce_sort_pmc.py > $O/${order}_p$i.log 2>&1 || echo "pass $order $i failed"
  done
done
python3 - $O <<'P'
import csv, glob, sys, collections
O = sys.argv[1]
for order in ("given", "sorted"):
    print(open(O + "/%s_time.log" % order).read().strip().splitlines()[-1])
    tot = {}
    for f in glob.glob(O + "/%s_p*/**/*counter_collection.csv" % order, recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("hare_voxel_pool"):
                per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in per.items():
            v.sort()
            last = [x for _, x in v[-5:]]
            tot[c] = sum(last) / len(last)
    for c in sorted(tot): print("   %-32s %.5g per cast" % (c, tot[c]))
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot: print("   -> HBM-side bytes per cast (2 x FETCH + WRITE) x 1024 = %.3f GB" % ((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / 1e9))
    if "SQ_THREAD_CYCLES_VALU" in tot: print("   -> lane utilisation %.3f" % (tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_ACTIVE_INST_VALU"] / 64))
P
