"""Static conformance of the C# P/Invoke layer with the C-ABI.

No .NET toolchain exists in the build image, so bindings/csharp/HareHip.cs has never met a compiler.  What CAN be checked
without one is checked here, by parsing both files:
  * every DllImport (its method name, or its EntryPoint = "..." alias) names a function include/hare_hip.h declares, and
    every declared export has at least one managed declaration;
  * parameter count, scalar widths (int <-> int32_t, uint <-> uint32_t, long <-> int64_t, double) and pointer positions
    (IntPtr / arrays / T* / ref / out / string <-> C pointers and array parameters) agree, as does the pointee where both sides
    name one (hare_ray[] <-> hare_ray*, int[] <-> const int32_t*, ...), and the return type;
  * every [StructLayout(Sequential)] struct has the field sequence, offsets and size of the header's struct of the same
    name, computed with natural alignment (the wire sizes 48 / 56 / 64 / 16 / 32 are also stated explicitly);
  * the HARE_* constants the C# side repeats have the header's values.
The reference seam the binding serves: Spatial_Partition.cs:27-35; wire types Hare_Geometry_Primitives.cs:393-481."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hare_hip.h")
CSHARP = os.path.join(ROOT, "bindings", "csharp", "HareHip.cs")

C_SCALARS = {"int32_t": ("int", 4), "int": ("int", 4), "uint32_t": ("uint", 4), "int64_t": ("long", 8), "uint64_t": ("ulong", 8),
             "double": ("double", 8)}
CS_SCALARS = {"int": 4, "uint": 4, "long": 8, "ulong": 8, "double": 8, "float": 4}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def strip_cs_comments(text):
    text = re.sub(r"///[^\n]*", " ", text)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


# ------------------------------------------------------------------------------------------------ the header
def c_param(p):
    """-> ("scalar", cs_name) | ("ptr", pointee or None)"""
    p = p.strip()
    is_array = "[" in p
    p = re.sub(r"\[[^\]]*\]", "", p)
    stars = p.count("*")
    toks = [t for t in re.split(r"[\s*]+", p) if t and t != "const"]
    base = toks[0] if toks[0] != "struct" else toks[1]
    if stars or is_array:
        if base == "char":
            return ("ptr", "char")
        if base in C_SCALARS:
            return ("ptr", C_SCALARS[base][0]) if stars + (1 if is_array else 0) == 1 else ("ptr", None)
        if base == "void":
            return ("ptr", None)
        if base == "hare_scene":
            return ("ptr", "scene" if stars == 1 else "scene*")
        return ("ptr", base)
    assert base in C_SCALARS, p
    return ("scalar", C_SCALARS[base][0])


def header_functions():
    text = strip_c_comments(open(HEADER).read())
    fns = {}
    for m in re.finditer(r"HARE_API\s+([^;(]*?)\b(hare_[a-z_0-9]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        ps = [] if params in ("", "void") else [c_param(p) for p in split_params(params)]
        rk = "ptr" if "*" in ret else ("void" if ret == "void" else "int")
        fns[name] = (rk, ps)
    return fns


def header_structs():
    text = strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ptr = "*" in decl
            toks = [t for t in re.split(r"[\s*]+", decl.replace(",", " , ")) if t and t != "const"]
            base = toks[0]
            for nm in "".join(toks[1:]).split(","):
                n = 1
                am = re.search(r"\[(\d+)\]", nm)
                if am:
                    n = int(am.group(1))
                size = 8 if ptr else C_SCALARS[base][1]
                fields += [size] * n
        out[m.group(3)] = fields
    return out


def header_constants():
    text = strip_c_comments(open(HEADER).read())
    return {k: int(v.rstrip("u")) for k, v in re.findall(r"#define\s+(HARE_[A-Z_]+)\s+\(?(-?\d+u?)\)?", text)}


# ------------------------------------------------------------------------------------------------ the C# file
def cs_param(p):
    p = re.sub(r"\[[A-Za-z, ]+\]\s*", "", p).strip()          # [In], [Out], [In, Out]
    toks = p.split()
    mod = toks[0] if toks[0] in ("ref", "out", "in") else None
    typ = toks[1] if mod else toks[0]
    if typ == "IntPtr":
        return ("ptr", "IntPtr*" if mod else None)
    if typ == "string":
        return ("ptr", "char")
    if typ == "IntPtr[]":
        return ("ptr", "scene*")
    if typ.endswith("[]") or typ.endswith("*") or mod:
        return ("ptr", typ.rstrip("[]*"))
    assert typ in CS_SCALARS, p
    return ("scalar", typ)


def cs_imports():
    text = strip_cs_comments(open(CSHARP).read())
    out = []
    for m in re.finditer(r"\[DllImport\(([^\]]*)\)\]\s*public\s+static\s+extern\s+(unsafe\s+)?(\w+)\s+(\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        attrs, ret, name, params = m.group(1), m.group(3), m.group(4), m.group(5)
        ep = re.search(r'EntryPoint\s*=\s*"(\w+)"', attrs)
        assert "CallingConvention.Cdecl" in attrs, name
        ps = [cs_param(p) for p in split_params(params)] if params.strip() else []
        rk = {"IntPtr": "ptr", "void": "void", "int": "int"}[ret]
        out.append((ep.group(1) if ep else name, name, rk, ps))
    return out


def cs_structs():
    text = strip_cs_comments(open(CSHARP).read())
    out = {}
    for m in re.finditer(r"\[StructLayout\(([^\]]*)\)\]\s*public\s+struct\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        assert "LayoutKind.Sequential" in m.group(1), m.group(2)
        fields = []
        for decl in m.group(3).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            toks = decl.split()
            assert toks[0] == "public", decl
            typ = toks[1]
            n = len("".join(toks[2:]).split(","))
            fields += [8 if typ == "IntPtr" else CS_SCALARS[typ]] * n
        out[m.group(2)] = fields
    return out


def layout(fields):
    """Natural-alignment layout of a sequence of primitive sizes: (offsets, total size)."""
    off, offs, amax = 0, [], 1
    for sz in fields:
        off = (off + sz - 1) // sz * sz
        offs.append(off)
        off += sz
        amax = max(amax, sz)
    return offs, (off + amax - 1) // amax * amax


# ------------------------------------------------------------------------------------------------ tests
def test_parsers_see_what_is_there():
    fns, imps, hs, cs = header_functions(), cs_imports(), header_structs(), cs_structs()
    assert len(fns) >= 32 and "hare_shoot_batch" in fns and len(imps) >= 34
    assert fns["hare_shoot_batch"][1][3] == ("scalar", "long") and fns["hare_shoot_batch"][1][4] == ("ptr", "hare_ray")
    assert {"hare_ray", "hare_xevent", "hare_counters", "hare_topology_desc", "hare_slim_event", "hare_slim_event_uv",
            "hare_voxel_info", "hare_tree_info"} <= set(hs) and set(hs) <= set(cs) | {"hare_scene"}


def test_every_export_is_bound_and_every_binding_names_an_export():
    fns = header_functions()
    bound = {ep for ep, _, _, _ in cs_imports()}
    assert bound <= set(fns), sorted(bound - set(fns))
    assert set(fns) <= bound, "exports without a managed declaration: %s" % sorted(set(fns) - bound)


# the one place where the managed pointee deliberately differs from the header's: `out` of the slim calls is declared hare_xevent* /
# const void* in C and receives hare_slim_event records (include/hare_hip.h, HARE_SHOOT_SLIM_EVENTS)
SLIM_OK = {("hare_shoot_batch_sharded_slim", 9): "hare_slim_event", ("hare_expand_events", 4): "hare_slim_event"}


@pytest.mark.parametrize("imp", cs_imports(), ids=lambda i: i[1] + "/" + str(len(i[3])))
def test_signature_matches_the_header(imp):
    ep, name, rk, ps = imp
    crk, cps = header_functions()[ep]
    assert rk == crk, f"{name}: return {rk} vs {crk}"
    assert len(ps) == len(cps), f"{name}: {len(ps)} parameters, header has {len(cps)}"
    for k, (a, b) in enumerate(zip(ps, cps)):
        assert a[0] == b[0], f"{name} parameter {k}: managed {a} vs C {b}"
        if a[0] == "scalar":
            assert a[1] == b[1], f"{name} parameter {k}: managed {a[1]} vs C {b[1]}"
        else:
            want = b[1]
            got = a[1]
            if (name, k) in SLIM_OK:
                assert got == SLIM_OK[(name, k)]
                continue
            if got is None or want is None:            # IntPtr <-> any pointer; void* <-> any array
                continue
            if want == "scene":                         # hare_scene* is an IntPtr; hare_scene** an `out IntPtr`
                assert got is None, f"{name} parameter {k}"
                continue
            if want == "scene*":
                assert got in ("scene*", "IntPtr*"), f"{name} parameter {k}: {got}"
                continue
            assert got == want, f"{name} parameter {k}: managed pointee {got} vs C {want}"


@pytest.mark.parametrize("name,size", [("hare_ray", 48), ("hare_xevent", 56), ("hare_counters", 64), ("hare_slim_event", 16),
                                       ("hare_slim_event_uv", 32), ("hare_topology_desc", 80), ("hare_voxel_info", 128),
                                       ("hare_tree_info", 24)])
def test_struct_layouts_match_the_header(name, size):
    c, m = header_structs()[name], cs_structs()[name]
    assert c == m, f"{name}: field sizes {m} (C#) vs {c} (header)"
    assert layout(c) == layout(m) and layout(c)[1] == size
    # ... and the header's own layout is what the compiler makes of it (gcc, this box)
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write('#include "hare_hip.h"\n#include <stdio.h>\nint main(void){printf("%%zu", sizeof(%s));return 0;}\n' % name)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(d, "s")])
        assert int(subprocess.check_output([os.path.join(d, "s")])) == size


def test_constants_repeat_the_header():
    text = strip_cs_comments(open(CSHARP).read())
    consts = header_constants()
    seen = 0
    for grp in re.findall(r"public\s+const\s+u?int\s+([^;]+);", text):
        for k, v in re.findall(r"(HARE_[A-Z_]+)\s*=\s*(-?\d+)", grp):
            assert consts[k] == int(v), k
            seen += 1
    assert seen >= 6
