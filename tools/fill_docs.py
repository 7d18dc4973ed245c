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
print("filled DESIGN.md, README.md, profiles/README.md from profiles/%s_*" % ROUND)
