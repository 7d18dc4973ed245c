"""Regenerates the measured tables of the documents from the committed files under profiles/ (prose must not drift from the
files): DESIGN.md section 6's table, README.md's table, and the generated part of profiles/README.md's round section -- everything between the
markers <!-- generated:NAME --> and <!-- /generated:NAME -->.   ROUND=r06 python tools/fill_docs.py"""
import json, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r06")
res = subprocess.run([sys.executable, os.path.join(R, "tools", "results_table.py"), "--readme"], capture_output=True, text=True, env=dict(os.environ, ROUND=ROUND)).stdout
cut = res.index("\nrocprofv3's average for the kernel")
design_tbl, prof_tbl = res[:cut].strip(), res[cut:].strip()

d = json.loads(open(os.path.join(R, "profiles", f"{ROUND}_default_bench.json")).read().strip())
c = d["configs"]
PREV = {"c2": "2 626", "c2_quads": "2 250", "c3": "759", "c4_shard": "1 725", "c5_shard": "764", "c4": "1 959", "c5": "959", "kd_hall": "682", "kd": "1 336"}


def row(label, l, prev):
    rf = l["roofline"]
    own = rf.get("own") or {}
    two = (l.get("two_streams") or {}).get("value")
    return f"| {label} | **{l['value']:.0f}**" + (f" (alternating two streams: {two:.0f})" if two else "") + f" | {rf['frac']:.2f} | {own.get('frac', 0):.2f} | {prev} |"


def line(cfg):
    return json.loads(open(os.path.join(R, "profiles", f"{ROUND}_{cfg}_bench.json")).read().strip())


rows = [row("C2 headline: 1 048 576 burst rays -> 100 908-triangle hall, `Voxel_Grid` D = 64", d, PREV["c2"]),
        row("the same rays into `hall_quads` (39 263 quads + 22 382 triangles, the same surfaces)", c["c2_quads"], PREV["c2_quads"]),
        row("C3: the same rays, `Octree` 8/16", c["c3"], PREV["c3"]),
        row("C4 shard: 2M rays -> 986 416-triangle cathedral, D = 128", c["c4_shard"], PREV["c4_shard"]),
        row("C5 shard: 1M rays x 8 specular bounces, cathedral, device-resident (Mcasts/s)", c["c5_shard"], PREV["c5_shard"]),
        row("**C4: 16 777 216 rays**, cathedral, on one GPU", c["c4"], PREV["c4"]),
        row("**C5: 8 388 608 rays x 8 bounces**, cathedral, on one GPU (Mcasts/s)", c["c5"], PREV["c5"]),
        row("`KDTree.Shoot`, 1M rays, the 100 908-triangle hall (16 / 8)", line("kd_hall"), PREV["kd_hall"]),
        row("`KDTree.Shoot`, 1M rays, 972-triangle shoebox (12 / 16)", line("kd"), PREV["kd"])]
readme_tbl = ("| workload | Mrays/s | `frac` (the reference's bytes per ray, SURVEY 8(d)) | `own.frac` (the bytes the kernel's own algorithm touches) | round 5 (driver's run) |\n"
              "|---|---|---|---|---|\n" + "\n".join(rows))


def fill(path, name, text):
    s = open(path).read()
    a, b = f"<!-- generated:{name} -->", f"<!-- /generated:{name} -->"
    if a not in s or b not in s:
        sys.exit(f"{path}: markers for {name} missing")
    s = s[:s.index(a) + len(a)] + "\n" + text + "\n" + s[s.index(b):]
    open(path, "w").write(s)


fill(os.path.join(R, "DESIGN.md"), "results", design_tbl)
fill(os.path.join(R, "README.md"), "results", readme_tbl)
fill(os.path.join(R, "profiles", "README.md"), f"{ROUND}_kernel_ms", prof_tbl)
# ---- DESIGN.md section 5's counters table of the round, from profiles/traffic.json (written by tools/condense_profiles.py)
def counters_table():
    t = json.load(open(os.path.join(R, "profiles", "traffic.json")))
    cols = [("C2: K1q, 1M", "hall-voxel-D64-n1048576"), ("K1q, 4M", "hall-voxel-D64-n4194304"), ("C3: K2d, 1M", "hall-octree-n1048576"), ("K2d, 262k rays", "hall-octree-n262144"),
            ("C4 shard: K1q `_g`", "cathedral-voxel-D128-n2097152"), ("C5: K1q `_g`, per cast", "cathedral-voxel-D128-n1048576-b8"), ("C4, 16.7M rays", "cathedral-voxel-D128-n16777216"),
            ("C5, 8.4M rays, per cast", "cathedral-voxel-D128-n8388608-b8")]
    cols = [(c, k) for c, k in cols if k in t]
    rows = [("VALU lane utilisation", "valu_lane_util", lambda v: f"{v:.2f}"), ("wave-VALU instructions", "SQ_INSTS_VALU", lambda v: f"{v:.2e}".replace("e+0", "e")),
            ("a wave is issuing / sits at a `s_waitcnt`", None, None), ("TA busy", "ta_busy_frac", lambda v: f"{100 * v:.0f} %"),
            ("L1 line accesses per ray", "l1_accesses_per_ray", lambda v: f"{v:.0f}"), ("HBM-side bytes per launch", "hbm_bytes_per_launch", lambda v: f"{v / 1e9:.2f} GB")]
    out = ["| | " + " | ".join(c for c, _ in cols) + " |", "|" + "---|" * (len(cols) + 1)]
    for name, key, fmt in rows:
        if key is None:
            out.append("| " + name + " | " + " | ".join(f"{100 * t[k]['sq_active_inst_any_frac']:.0f} % / {100 * t[k]['sq_wait_any_frac']:.0f} %" for _, k in cols) + " |")
        else:
            out.append("| " + name + " | " + " | ".join(fmt(t[k][key]) if t[k].get(key) is not None else "" for _, k in cols) + " |")
    return "\n".join(out)


fill(os.path.join(R, "DESIGN.md"), f"{ROUND}_counters", counters_table())
print("filled DESIGN.md, README.md, profiles/README.md from profiles/%s_*" % ROUND)
