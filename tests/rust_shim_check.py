"""A small structural checker for rust/triple_accel_amd/src/lib.rs (test infrastructure; there is no rustc in the build image).

It does what a compiler's name resolution and an FFI lint would do for the `extern "C"` boundary of the shim:

* `extern_fns(rs)`        every item of the `extern "C"` block: name, [(param, type)], return type;
* `defined_types(rs)`     every `struct` / `enum` / `type` the file defines, plus what its `use` lines import;
* `repr_c_structs(rs)`    `#[repr(C)]` structs with their fields in order;
* `header_fns(h)`, `header_structs(h)`   the same from include/triple_accel_amd.h;
* `c_to_rust(ctype)`      the Rust spelling an FFI declaration of that C type must have.
"""
import re

PRIMITIVES = {"u8", "u16", "u32", "u64", "usize", "i8", "i16", "i32", "i64", "isize", "bool", "f32", "f64", "char", "str"}

C_SCALARS = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "int": "c_int",
             "void": "c_void", "char": "c_char"}
# header struct -> the shim's #[repr(C)] struct; opaque handles travel as c_void
C_STRUCTS = {"ta_edit_costs": "TaEditCosts", "ta_match": "TaMatch", "ta_edit": "TaEdit", "ta_strings": "TaStrings",
             "ta_queue": "c_void", "ta_lev_select": "TaLevSelect", "ta_launch_info": "TaLaunchInfo"}


def strip_rust_comments(rs):
    rs = re.sub(r"/\*.*?\*/", "", rs, flags=re.S)
    return re.sub(r"//[^\n]*", "", rs)


def strip_c_comments(h):
    return re.sub(r"/\*.*?\*/", "", h, flags=re.S)


def _split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[{":
            depth += 1
        elif ch in ")>]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip()]


def extern_block(rs):
    rs = strip_rust_comments(rs)
    i = rs.index('extern "C" {')
    depth, j = 0, i + len('extern "C" ')
    while True:
        if rs[j] == "{":
            depth += 1
        elif rs[j] == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    return rs[i + len('extern "C" {'):j]


def norm_type(t):
    return re.sub(r"\s+", " ", t.strip())


def extern_fns(rs):
    fns = {}
    for m in re.finditer(r"\bfn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", extern_block(rs), re.S):
        params = []
        for p in _split_top(m.group(2)):
            name, ty = p.split(":", 1)
            params.append((name.strip(), norm_type(ty)))
        fns[m.group(1)] = (params, norm_type(m.group(3)) if m.group(3) else None)
    return fns


def type_idents(t):
    """identifiers a type expression names (pointer / reference sigils, `const`, `mut`, `dyn` dropped)"""
    ids = re.findall(r"[A-Za-z_][A-Za-z_0-9]*", t)
    return [i for i in ids if i not in ("const", "mut", "dyn")]


def defined_types(rs):
    rs = strip_rust_comments(rs)
    names = set(re.findall(r"\b(?:struct|enum|type|union)\s+([A-Za-z_]\w*)", rs))
    for m in re.finditer(r"\buse\s+([^;]+);", rs):
        body = m.group(1).strip()
        if body.startswith(("super::", "crate::", "self::")):
            continue                                       # a path into this crate defines nothing: its target must exist
        g = re.search(r"\{([^}]*)\}", body)
        leaves = _split_top(g.group(1)) if g else [body.split("::")[-1]]
        for leaf in leaves:
            leaf = leaf.split(" as ")[-1].strip()
            if leaf not in ("*", "self"):
                names.add(leaf.split("::")[-1])
    return names


def repr_c_structs(rs):
    rs = strip_rust_comments(rs)
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub\s+struct\s+(\w+)\s*\{([^}]*)\}", rs, re.S):
        fields = []
        for f in _split_top(m.group(2)):
            name, ty = re.sub(r"^pub(\([a-z]+\))?\s+", "", f).split(":", 1)
            fields.append((name.strip(), norm_type(ty)))
        out[m.group(1)] = fields
    return out


def c_to_rust(ctype):
    """`const uint8_t *` -> `*const u8`, `ta_edit **` -> `*mut *mut TaEdit`, `const uint32_t **` -> `*mut *const u32`"""
    t = ctype.replace("*", " * ").split()
    const_base = False
    if t and t[0] == "const":
        const_base, t = True, t[1:]
    if t and t[0] in ("struct", "enum"):
        t = t[1:]
    base, stars = t[0], t[1:]
    assert all(s in ("*", "const") for s in stars), ctype
    r = C_SCALARS.get(base) or C_STRUCTS.get(base)
    assert r, "no Rust spelling for C type %r" % ctype
    n = stars.count("*")
    for lvl in range(n):
        r = ("*const " if (lvl == 0 and const_base) else "*mut ") + r
    return r


def _c_params(args):
    args = args.strip()
    if args in ("", "void"):
        return []
    out = []
    for a in _split_top(args):
        m = re.match(r"(.*?)(\w+)\s*$", a, re.S)              # the last identifier is the parameter's name
        out.append((m.group(2), re.sub(r"\s+", " ", m.group(1).strip())))
    return out


def header_fns(h):
    h = strip_c_comments(h)
    fns = {}
    for m in re.finditer(r"^\s*([A-Za-z_][\w\s\*]*?)\b(ta_\w+)\s*\(([^;{]*?)\)\s*;", h, re.M | re.S):
        fns[m.group(2)] = (_c_params(m.group(3)), re.sub(r"\s+", " ", m.group(1).strip()))
    return fns


def header_structs(h):
    h = strip_c_comments(h)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s*\{([^}]*)\}\s*(\w+)\s*;", h, re.S):
        fields = []
        for f in m.group(1).split(";"):
            f = f.strip()
            if f:
                mm = re.match(r"(.*?)(\w+)\s*$", f, re.S)
                fields.append((mm.group(2), re.sub(r"\s+", " ", mm.group(1).strip())))
        out[m.group(2)] = fields
    return out
