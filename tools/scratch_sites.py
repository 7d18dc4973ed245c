"""Where the scratch (private segment) accesses of the production kernels sit.
    python tools/scratch_sites.py [kernel ...] > profiles/r06_scratch_sites.md
Reads hare_amd/csrc/build/hare_kernels.s (written by the library's Makefile).  For every scratch_load / scratch_store of a kernel: the basic
block it is in, the loop nest that block belongs to (the compiler's own "in Loop: Header=... Depth=d" annotation), whether the block also
holds the instructions of a hot phase (FP64 compares / adds of the walk, cull or exact arithmetic), and the compiler's spill note.  Depth 1
is the kernel's round loop (one pass per ROUND of a wave: tens of rays' phases), depth >= 2 an inner loop (per step / per candidate)."""
import os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "hare_amd", "csrc", "build", "hare_kernels.s")).read().split("\n")
want = sys.argv[1:] or ["hare_voxel_pool_tri", "hare_voxel_pool_quad", "hare_voxel_pool_tri_g", "hare_voxel_bounce_tri_g", "hare_octree_dense", "hare_kdtree_dense"]

def kernel_lines(name):
    out, on = [], False
    for i, l in enumerate(src):
        if l.startswith(name + ":"):
            on = True
        if on:
            out.append((i + 1, l))
            if l.startswith(".Lfunc_end"):
                break
    return out

meta = {}
txt = "\n".join(src)
for m in re.finditer(r"\.name:\s+(\w+)\n(.*?)\.wavefront_size", txt, re.S):
    g = lambda k: int(re.search(k + r":\s+(\d+)", m.group(2)).group(1))
    meta[m.group(1)] = (g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.sgpr_count"), g(r"\.sgpr_spill_count"), g(r"\.private_segment_fixed_size"))

print("# Scratch accesses of the production kernels, by loop (round 6; `tools/scratch_sites.py`)\n")
print("Depth 1 = the kernel's round loop (executed once per round of a wave, on the path named); depth >= 2 = an inner loop (per DDA step, per")
print("candidate, per pop).  \"hot block\" = the basic block also holds FP64 arithmetic of a phase.\n")
for k in want:
    L = kernel_lines(k)
    if not L:
        continue
    v = meta.get(k)
    print(f"## `{k}` — {v[0]} VGPRs ({v[1]} spilled), {v[2]} SGPRs ({v[3]} spilled to VGPR lanes), {v[4]} B scratch per lane\n")
    n_readlane = sum(1 for _, l in L if "v_readlane_b32" in l)
    n_writelane = sum(1 for _, l in L if "v_writelane_b32" in l)
    rows = []
    block, depth, header, block_start = "entry", 0, "-", 0
    blocks = {}
    for idx, (ln, l) in enumerate(L):
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %bb\.(\d+):", l)
        if m:
            block = m.group(1)
            block_start = idx
            d = re.search(r"Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            nxt = L[idx + 1][1] if idx + 1 < len(L) else ""
            d = d or re.search(r"Loop Header: Depth=(\d+)", l) or re.search(r"Inner Loop Header: Depth=(\d+)", l)
            if d and d.lastindex == 2:
                header, depth = d.group(1), int(d.group(2))
            elif d:
                header, depth = block.lstrip(".L"), int(d.group(1))
            elif "Loop" not in l:
                header, depth = "-", 0
        if "scratch_load" in l or "scratch_store" in l:
            # the block's instruction mix
            j = block_start
            f64 = 0
            while j < len(L) and (j == block_start or not (re.match(r"^\.LBB\d+_\d+:", L[j][1]) or re.match(r"^; %bb\.", L[j][1]))):
                if re.search(r"v_(add|mul|fma|cmp_\w+)_f64", L[j][1]):
                    f64 += 1
                j += 1
            note = l.split(";")[-1].strip() if ";" in l else ""
            rows.append((ln, l.split()[0], block, depth, header, "yes" if f64 >= 3 else "no", note))
    print(f"SGPR spill traffic: {n_writelane} `v_writelane` / {n_readlane} `v_readlane` (lane moves, no memory).\n")
    if not rows:
        print("No scratch access.\n")
        continue
    print("| line in hare_kernels.s | instruction | block | loop depth | loop header | hot block | compiler's note |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print("| " + " | ".join(str(x) for x in r) + " |")
    print()
