"""Developer tool: per-kernel means of rocprofv3 --pmc counter CSVs (counter_collection.csv files under a directory)."""
import csv, glob, os, sys
from collections import defaultdict
for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", d)
    for k, cs in acc.items():
        if not k.startswith("hare_"):
            continue
        print(" ", k, {c: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for c, v in sorted(cs.items())})
