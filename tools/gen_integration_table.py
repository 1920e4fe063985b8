"""Regenerate the appendix of INTEGRATION.md: one row per entry point of include/ocn_mi355x.h with the Julia `ccall` type tuple a binder
writes (python tools/gen_integration_table.py). The prose sections above the marker line are kept."""
import os
import re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARK = "## 6. Every entry point and its `ccall` signature"
hdr = open(os.path.join(ROOT, "include", "ocn_mi355x.h")).read()
body = hdr[hdr.index("extern \"C\""):]
body = re.sub(r"/\*.*?\*/", lambda m: " " * len(m.group(0)) if "\n" not in m.group(0) else "\n" * m.group(0).count("\n"), body, flags=re.S)
decls = re.findall(r"((?:const\s+)?(?:int|void|char|double)\s*\*?\s*)(ocn_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", body, flags=re.S)


def jl_type(t):
    t = " ".join(t.replace("const", " ").split())
    name = re.sub(r"\[\d*\]", "", t.split()[-1].lstrip("*")) if t else ""
    core = t
    if "(*" in core:                                   # const int (*locs)[3]
        return "Ptr{Cint}"
    ptr = core.count("*") + core.count("[")
    base = core.split()[0] if core else ""
    if base in ("ocn_grid_t", "ocn_poisson_t", "ocn_model_t", "ocn_dist_t", "ocn_dist_poisson_t"):
        return "Ptr{Ptr{Cvoid}}" if ptr else "Ptr{Cvoid}"
    if base == "ocn_bc_t":
        return "Ptr{OcnBC}"
    if base == "ocn_transport_t":
        return "Ptr{OcnTransport}"
    if base == "char":
        return "Cstring"
    if base == "void":
        return "Ptr{Ptr{Cvoid}}" if ptr > 1 else "Ptr{Cvoid}"
    scalar = {"int": "Cint", "double": "Cdouble", "size_t": "Csize_t", "long": "Clong", "int64_t": "Int64", "unsigned": "Cuint"}.get(base, base)
    if ptr >= 2:
        return "Ptr{Ptr{%s}}" % scalar
    if ptr == 1:
        return "Ptr{%s}" % scalar
    return scalar


rows = []
for ret, name, args in decls:
    args = " ".join(args.split())
    types = [] if args in ("", "void") else [jl_type(a.strip()) for a in re.split(r",(?![^()]*\))", args)]
    r = " ".join(ret.split())
    rt = "Cstring" if "char" in r else ("Ptr{Cvoid}" if "void" in r and "*" in r else ("Cvoid" if r.startswith("void") else "Cint"))
    tup = "(" + ", ".join(types) + ("," if len(types) == 1 else "") + ")"
    rows.append(f"| `{name}` | `ccall((:{name}, libocn), {rt}, {tup}, …)` |")
table = [MARK, "",
         "Generated from `include/ocn_mi355x.h` by `tools/gen_integration_table.py` (%d entry points; the header's comment on each one cites the" % len(rows),
         "reference function it replaces). Handles are `Ptr{Cvoid}`, device arrays are passed as `Ptr{Cdouble}` obtained from",
         "`pointer(parent(field.data))`, arrays of device pointers as `Ptr{Ptr{Cdouble}}` built on the host; `OcnBC` / `OcnTransport` mirror",
         "`ocn_bc_t` / `ocn_transport_t` field by field (`struct OcnBC; kind::Cint; value::Cdouble; array::Ptr{Cdouble}; end`). Every `Cint`",
         "return is a status: `check(rc)` of §1.", "", "| entry point | binding |", "|---|---|"] + rows
p = os.path.join(ROOT, "INTEGRATION.md")
s = open(p).read()
if MARK in s:
    s = s[:s.index(MARK)].rstrip() + "\n\n"
else:
    s = s.rstrip() + "\n\n"
open(p, "w").write(s + "\n".join(table) + "\n")
print(len(rows), "entry points")
