"""Developer tool: the results table of DESIGN.md section 6 from the files under profiles/ -- per configuration the bench line
(profiles/<round>_<cfg>_bench.json), rocprofv3's average for the kernel (profiles/<round>_<cfg>_kernel_stats.csv) and, when given, the
default run's whole line (profiles/<round>_default_bench.json: c4 / c5 at full size).   ROUND=r05 python tools/results_table.py"""
import csv, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r05")
CFG = [("c2", "C2: 1M rays, hall, `Voxel_Grid` D = 64 (headline)"), ("c2_4M", "same scene, 4M rays"), ("c2_quads", "the same rays into `hall_quads` (39 263 quads + 22 382 triangles)"),
       ("c3", "C3: 1M rays, hall, `Octree` 8 / 16"), ("c3_262k", "octree, 262 144 rays"), ("c4shard", "C4 shard: 2M rays, cathedral, D = 128"),
       ("c5", "C5 shard: 1M rays x 8 bounces, cathedral"), ("kd_hall", "`KDTree` 16 / 8: 1M rays, the hall"), ("kd", "`KDTree` 12 / 16: 1M rays, the 972-triangle shoebox")]


def rocprof_avg(cfg, kernel):
    p = os.path.join(R, "profiles", f"{ROUND}_{cfg}_kernel_stats.csv")
    if not os.path.exists(p):
        return None
    for r in csv.DictReader(open(p)):
        if r["Name"].split("(")[0] == kernel:
            return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    return None


def row(label, l):
    rf = l.get("roofline") or {}
    own, issue, cpu = rf.get("own") or {}, rf.get("issue") or {}, l.get("cpu_baseline") or {}
    two = (l.get("two_streams") or {}).get("value")
    return [label, rf.get("kernel", ""), f"**{l['value']:.0f}**" + (f" (two streams: {two:.0f})" if two else ""), f"{rf.get('kernel_ms', 0):.3f}",
            f"{rf.get('frac', 0):.2f}", f"{own.get('frac'):.2f}" if own.get("frac") else "", f"{issue.get('frac'):.2f}" if issue.get("frac") else "",
            f"{issue.get('lane_util'):.2f}" if issue.get("lane_util") else "", f"{rf['traffic'] / 1e9:.2f}" if rf.get("traffic") else "",
            f"{cpu.get('value')} ({cpu.get('cores')})" if cpu else "", "bit-exact" if l.get("x_event_parity_vs_oracle") else str(l.get("x_event_parity_vs_oracle"))]


print("| config | kernel | Mrays/s (bench `value`) | kernel ms (rocprofv3 average x calls) | `frac` (SURVEY 8(d)) | `own.frac` | `issue.frac` | lane utilisation | HBM-side GB per launch | CPU oracle Mrays/s (threads) | parity |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for cfg, label in CFG:
    p = os.path.join(R, "profiles", f"{ROUND}_{cfg}_bench.json")
    if not os.path.exists(p):
        continue
    l = json.loads(open(p).read().strip().splitlines()[-1])
    r = row(label, l)
    ra = rocprof_avg(cfg, (l.get("roofline") or {}).get("kernel", ""))
    if ra:
        r[3] += f" ({ra[0]:.1f} us x {ra[1]})"
    print("| " + " | ".join(r) + " |")
p = os.path.join(R, "profiles", f"{ROUND}_default_bench.json")
if os.path.exists(p):
    l = json.loads(open(p).read().strip().splitlines()[-1])
    print("| **default run** (`python bench.py`), headline | " + " | ".join(row("", l)[1:]) + " |")
    for name, label in (("c3", "... c3"), ("c2_quads", "... c2_quads"), ("c4_shard", "... c4_shard"), ("c5_shard", "... c5_shard"), ("c4", "**C4: 16 777 216 rays**, cathedral, one GPU"), ("c5", "**C5: 8 388 608 rays x 8 bounces**, one GPU")):
        if name in l.get("configs", {}):
            print("| " + " | ".join(row(label, l["configs"][name])) + " |")
    print(f"\n(the default line: {len(open(p).read().strip())} bytes)")
