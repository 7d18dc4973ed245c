"""Register / spill / scratch / LDS figures of every kernel in the built code object, from the metadata the compiler
writes into hare_amd/csrc/build/hare_kernels.s (made next to the code object by the Makefile)."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "hare_amd/csrc/build/hare_kernels.s"
txt = open(path).read()
meta = txt[txt.index("amdhsa.kernels:"):]
rows = []
for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
    def g(key):
        m = re.search(r"\." + key + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    rows.append((g("name"), g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
                 g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print(f"{'kernel':34s} {'vgpr':>5s} {'vspill':>6s} {'sgpr':>5s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}")
for r in sorted(rows):
    if len(sys.argv) > 2 and sys.argv[2] not in r[0]:
        continue
    print(f"{r[0]:34s} {r[1]:>5s} {r[2]:>6s} {r[3]:>5s} {r[4]:>6s} {r[5]:>7s} {r[6]:>6s}")
