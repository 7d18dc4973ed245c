"""Condenses the rocprofv3 output of tools/make_profiles.sh into the small files kept under profiles/:
per configuration the kernel-stats table (top rows), the per-dispatch means of the PMC counters for the hare_* shoot
kernels, and profiles/traffic.json (HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, FETCH doubled per the
gfx950 note of MI355X_MICROARCH.md; keyed like bench.py's workload key, stamped with the kernel-source hash)."""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha)

SHOOT = ("hare_voxel_persist", "hare_voxel_pool", "hare_octree_persist", "hare_octree_pool", "hare_octree_dense", "hare_octree_group", "hare_octree_tail", "hare_kdtree",
         "hare_cost_order",        # the order pass in front of the pool kernel on large batches of primary rays: part of the cast
         "hare_reflect")
NSHOOT = len(SHOOT) - 1          # the kernels a shoot consists of (hare_reflect is listed in the summary, not priced)
ROUND = os.environ.get("ROUND", "r06")
KEYS = {"c2": "hall-voxel-D64-n1048576", "c2_4M": "hall-voxel-D64-n4194304", "c3": "hall-octree-n1048576", "c3_262k": "hall-octree-n262144",
        "c4shard": "cathedral-voxel-D128-n2097152", "c5": "cathedral-voxel-D128-n1048576-b8", "kd": "shoebox-kdtree-n1048576",
        "kd_hall": "hall-kdtree-n1048576", "c2_quads": "hall_quads-voxel-D64-n1048576",
        "c4": "cathedral-voxel-D128-n16777216", "c5full": "cathedral-voxel-D128-n8388608-b8"}


def counters(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def issue_side(cs, key):
    """What the SQ / TA counters say about the issue side, as fractions: a wave is issuing / sits at a waitcnt
    (of its own cycles), the texture-address units are busy (of the kernel's cycles), L1 line accesses per ray."""
    out = {}
    wc = cs.get("SQ_WAVE_CYCLES")
    if wc:
        if cs.get("SQ_ACTIVE_INST_ANY"): out["sq_active_inst_any_frac"] = round(cs["SQ_ACTIVE_INST_ANY"] / wc, 4)
        if cs.get("SQ_WAIT_ANY"): out["sq_wait_any_frac"] = round(cs["SQ_WAIT_ANY"] / wc, 4)
    if cs.get("SQ_THREAD_CYCLES_VALU") and cs.get("SQ_ACTIVE_INST_VALU"):
        # lanes that really execute, of the 64 a vector instruction has: the measured counterpart of "lane occupancy"
        out["valu_lane_util"] = round(cs["SQ_THREAD_CYCLES_VALU"] / cs["SQ_ACTIVE_INST_VALU"] / 64.0, 4)
    if cs.get("SQ_INSTS_VMEM_RD"):
        out["vmem_rd_insts"] = int(cs["SQ_INSTS_VMEM_RD"])
        out["vmem_wr_insts"] = int(cs.get("SQ_INSTS_VMEM_WR", 0))
    if cs.get("TCP_TCC_READ_REQ_sum"):
        out["tcc_read_req"] = int(cs["TCP_TCC_READ_REQ_sum"])
        out["tcc_write_req"] = int(cs.get("TCP_TCC_WRITE_REQ_sum", 0))
    if cs.get("TA_TA_BUSY_sum") and cs.get("GRBM_GUI_ACTIVE"):
        out["ta_busy_frac"] = round(cs["TA_TA_BUSY_sum"] / 256.0 / (cs["GRBM_GUI_ACTIVE"] / 8.0), 4)
    try:
        n = int([p for p in key.split("-") if p.startswith("n")][0][1:])
        if cs.get("TCP_TOTAL_CACHE_ACCESSES_sum"): out["l1_accesses_per_ray"] = round(cs["TCP_TOTAL_CACHE_ACCESSES_sum"] / n, 1)
    except Exception:
        pass
    return out


def main(src):
    out = os.path.join(ROOT, "profiles")
    # gpurun MERGES a run's output into the local directory: files of an earlier run (other process ids in their names) would be
    # averaged in with this one's.  Refuse a directory that holds more than one generation.
    times = [os.path.getmtime(f) for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True)]
    if times and max(times) - min(times) > 3600:
        sys.exit("condense_profiles: %s holds csv files more than an hour apart -- remove the older run's files first" % src)
    rows = [("config", "kernel", "counter", "mean_per_dispatch", "dispatches")]
    traffic = {}
    # ONLY="c3_262k": a re-take of some configurations -- the other configurations' rows of <round>_pmc_summary.csv and their entries of
    # traffic.json are kept as they are (without it, everything not in `src` is dropped from both files)
    only = os.environ.get("ONLY", "").split()
    kept_rows, kept_traffic = [], {}
    if only:
        ps = os.path.join(out, f"{ROUND}_pmc_summary.csv")
        if os.path.exists(ps):
            kept_rows = [tuple(ln.split(",")) for ln in open(ps).read().splitlines()[1:] if ln and ln.split(",")[0] not in only]
        tj0 = os.path.join(out, "traffic.json")
        if os.path.exists(tj0):
            kept_traffic = {k: v for k, v in json.load(open(tj0)).items() if k not in [KEYS[t] for t in only if t in KEYS]}
    for tag, key in KEYS.items():
        if only and tag not in only:
            continue
        ks = glob.glob(os.path.join(src, tag + "_kt", "**", "*kernel_stats.csv"), recursive=True)
        if ks:
            lines = open(ks[0]).read().splitlines()
            with open(os.path.join(out, f"{ROUND}_{tag}_kernel_stats.csv"), "w") as f:
                f.write("\n".join(lines[:9]) + "\n")
        per = {}
        for cset in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2", "TA", "F64"):
            for k, cs in counters(os.path.join(src, f"{tag}_{cset}")).items():
                if not k.startswith(SHOOT):
                    continue
                for c, v in cs.items():
                    if cset == "SQ2" and c == "SQ_ACTIVE_INST_VALU":
                        c = "SQ_ACTIVE_INST_VALU_pass2"      # measured again beside SQ_THREAD_CYCLES_VALU, so that the ratio is of one pass
                    if cset == "F64" and c == "SQ_INSTS_VALU":
                        c = "SQ_INSTS_VALU_pass_f64"         # the FP64 instruction classes of round 5 (bench.py: roofline.issue) and the total beside them
                    rows.append((tag, k, c, "%.6g" % (sum(v) / len(v)), len(v)))
                    per.setdefault(k, {})[c] = sum(v) / len(v)
        # a shoot may be two kernels on the stream (K2p + its tail): the launch's figures are their sums; `kernel` names the longer one
        shoot = {k: cs for k, cs in per.items() if k.startswith(SHOOT[:NSHOOT])}
        if shoot and all("FETCH_SIZE" in cs and "WRITE_SIZE" in cs for cs in shoot.values()):
            tot = defaultdict(float)
            for cs in shoot.values():
                for c, v in cs.items():
                    tot[c] += v
            if "SQ_ACTIVE_INST_VALU_pass2" in tot:
                tot["SQ_ACTIVE_INST_VALU"] = tot["SQ_ACTIVE_INST_VALU_pass2"] if "SQ_THREAD_CYCLES_VALU" in tot else tot.get("SQ_ACTIVE_INST_VALU", 0)
            cs = dict(tot)
            k = max(shoot, key=lambda kk: shoot[kk].get("SQ_WAVE_CYCLES", 0))
            if True:
                traffic[key] = {"kernel": k, "kernels": sorted(shoot), "FETCH_SIZE_KB": cs["FETCH_SIZE"], "WRITE_SIZE_KB": cs["WRITE_SIZE"],
                                "hbm_bytes_per_launch": int((2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024),
                                "SQ_INSTS_VALU": cs.get("SQ_INSTS_VALU"), "kernel_sha16": bench.kernel_source_sha(),
                                **{c: cs.get(c) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")
                                   if cs.get(c) is not None},
                                **issue_side(cs, key),
                                "note": "separate --pmc passes (rocprofv3); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH doubled per "
                                        "MI355X_MICROARCH.md (gfx950 tallies 128-B reads at 64 B; calibrated for wide coalesced reads, so an "
                                        "upper bound for this kernel's 16-B gathers)"}
        b = os.path.join(src, tag + "_bench.json")
        if os.path.exists(b):
            txt = [ln for ln in open(b).read().splitlines() if ln.startswith("{")]
            if txt:
                open(os.path.join(out, f"{ROUND}_{tag}_bench.json"), "w").write(txt[-1] + "\n")
    rows = rows[:1] + kept_rows + rows[1:]
    with open(os.path.join(out, f"{ROUND}_pmc_summary.csv"), "w") as f:
        for r in rows:
            f.write(",".join(str(x) for x in r) + "\n")
    old = {}
    tj = os.path.join(out, "traffic.json")
    if os.path.exists(tj):
        old = {k: v for k, v in json.load(open(tj)).items() if k not in traffic and "(" in k}     # keep the labelled history entries
    old.update(kept_traffic)
    old.update(traffic)
    json.dump(old, open(tj, "w"), indent=1)
    for r in rows:
        print(*r)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
