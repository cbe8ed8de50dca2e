#!/usr/bin/env python3
"""Extract the reference's known-answer tests into data (tests/golden/kats.json).

Reads (never copies) the reference's own test inputs and expected outputs:
  * /root/reference/tests/basic_tests.rs      -- 19 #[test] fns
  * doc-test examples in /root/reference/src/{lib,hamming,levenshtein}.rs (Appendix B.2)
and writes one JSON record per call: the function name, its argument VALUES and the
asserted result VALUES.  Only data leaves the reference -- no source text.

Run in the build container (the GPU box has no /root/reference):
    python tests/golden/extract_kats.py
"""
import json
import os
import re
import sys

REF = os.environ.get("TA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kats.json")


# ----------------------------------------------------------------------------- lexing helpers
def unescape_bytes(body: str) -> bytes:
    out = bytearray()
    i = 0
    while i < len(body):
        c = body[i]
        if c == "\\":
            n = body[i + 1]
            if n == "0":
                out.append(0); i += 2
            elif n == "n":
                out.append(10); i += 2
            elif n == "t":
                out.append(9); i += 2
            elif n == "r":
                out.append(13); i += 2
            elif n == "\\":
                out.append(92); i += 2
            elif n == '"':
                out.append(34); i += 2
            elif n == "'":
                out.append(39); i += 2
            elif n == "x":
                out.append(int(body[i + 2:i + 4], 16)); i += 4
            else:
                raise ValueError("unknown escape \\" + n)
        else:
            out.append(ord(c)); i += 1
    return bytes(out)


def split_top(s: str, sep: str):
    """Split s on sep at bracket depth 0, outside string literals."""
    parts, depth, cur, i = [], 0, [], 0
    in_str = False
    while i < len(s):
        c = s[i]
        if in_str:
            cur.append(c)
            if c == "\\":
                cur.append(s[i + 1]); i += 2; continue
            if c == '"':
                in_str = False
        else:
            if c == '"':
                in_str = True; cur.append(c)
            elif c in "([{":
                depth += 1; cur.append(c)
            elif c in ")]}":
                depth -= 1; cur.append(c)
            elif c == sep and depth == 0:
                parts.append("".join(cur)); cur = []
            else:
                cur.append(c)
        i += 1
    if "".join(cur).strip():
        parts.append("".join(cur))
    return [p.strip() for p in parts]


def find_call(expr: str):
    """Return (fname, argstr, tail) for the first `ident(` call in expr, else None."""
    m = re.search(r"([A-Za-z_][A-Za-z0-9_:]*)\(", expr)
    if not m:
        return None
    start = m.end()
    depth, i, in_str = 1, start, False
    while i < len(expr):
        c = expr[i]
        if in_str:
            if c == "\\":
                i += 2; continue
            if c == '"':
                in_str = False
        elif c == '"':
            in_str = True
        elif c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    return m.group(1), expr[start:i], expr[i + 1:]


# ----------------------------------------------------------------------------- value parsing
COSTS = {
    "LEVENSHTEIN_COSTS": {"mismatch": 1, "gap": 1, "start_gap": 0, "transpose": None},
    "RDAMERAU_COSTS": {"mismatch": 1, "gap": 1, "start_gap": 0, "transpose": 1},
}


def parse_value(tok: str, env: dict):
    tok = tok.strip()
    if tok.startswith("&mut "):
        tok = tok[5:].strip()
    if tok.startswith("&"):
        tok = tok[1:].strip()
    if tok.startswith('b"') and tok.endswith('"'):
        return {"bytes": unescape_bytes(tok[2:-1])}
    if tok.startswith('"') and tok.endswith('"'):
        return {"str": tok[1:-1]}
    if re.fullmatch(r"\d+(u8|u32|usize)?", tok):
        return int(re.match(r"\d+", tok).group(0))
    if tok in ("true", "false"):
        return tok == "true"
    if tok in COSTS:
        return {"costs": COSTS[tok]}
    if tok.startswith("EditCosts::new("):
        a = split_top(tok[len("EditCosts::new("):-1], ",")
        tr = a[3].strip()
        trv = None if tr == "None" else int(re.match(r"Some\((\d+)\)", tr).group(1))
        return {"costs": {"mismatch": int(a[0]), "gap": int(a[1]), "start_gap": int(a[2]), "transpose": trv}}
    if tok.startswith("SearchType::"):
        return {"search_type": tok.split("::")[1]}
    if tok in env:
        return env[tok]
    raise KeyError("cannot evaluate " + tok)


def parse_match(tok: str):
    m = re.fullmatch(r"Match\s*\{\s*start:\s*(\d+),\s*end:\s*(\d+),\s*k:\s*(\d+)\s*\}", tok.strip())
    return {"start": int(m.group(1)), "end": int(m.group(2)), "k": int(m.group(3))}


def parse_vec(tok: str, item):
    tok = tok.strip()
    assert tok.startswith("vec![") and tok.endswith("]"), tok
    inner = tok[5:-1].strip()
    return [item(x) for x in split_top(inner, ",")] if inner else []


def parse_edit(tok: str):
    m = re.fullmatch(r"Edit\s*\{\s*edit:\s*EditType::(\w+),\s*count:\s*(\d+)\s*\}", tok.strip())
    return {"edit": m.group(1), "count": int(m.group(2))}


def parse_rhs(rhs: str):
    rhs = rhs.strip()
    if re.fullmatch(r"\d+", rhs):
        return {"value": int(rhs)}
    if rhs == "vec![]" or re.match(r"vec!\[\s*Match\s*\{", rhs):
        return {"matches": parse_vec(rhs, parse_match)}
    if re.match(r"vec!\[\s*Edit\s*\{", rhs):
        return {"trace": parse_vec(rhs, parse_edit)}
    if rhs.startswith("Match"):
        return {"first_match": parse_match(rhs)}
    m = re.fullmatch(r"\((\d+),\s*Some\((vec!\[.*\])\)\)", rhs, re.S)
    if m:
        return {"value": int(m.group(1)), "trace": parse_vec(m.group(2), parse_edit)}
    raise ValueError("unparsed rhs: " + rhs)


# ----------------------------------------------------------------------------- statement interpreter
SKIP_FNS = {"alloc_str", "fill_str", "levenstein_naive_str", "levenshtein_simd_k_str", "collect", "unwrap", "vec", "assert"}


def jsonable(v):
    if isinstance(v, dict) and "bytes" in v:
        b = v["bytes"]
        return {"hex": b.hex(), "repr": b.decode("latin-1").encode("unicode_escape").decode("ascii")}
    if isinstance(v, dict) and "costs" in v:
        return v["costs"]
    if isinstance(v, dict) and "search_type" in v:
        return v["search_type"]
    if isinstance(v, dict) and "str" in v:
        return {"str": v["str"]}
    return v


def run_block(stmts, source, records):
    env, calls = {}, {}
    for lineno, st in stmts:
        st = st.strip()
        if not st or st.startswith("use ") or st.startswith("#"):
            continue
        if st.startswith("assert!("):
            cond = st[len("assert!("):-1].strip()
            m = re.fullmatch(r"(\w+)\.is_none\(\)", cond)
            if m:
                rec = calls.get(m.group(1))
                if rec is not None:
                    rec["expect"]["none"] = True
                continue
            if "==" not in cond:
                continue
            lhs, rhs = [x.strip() for x in cond.split("==", 1)]
            root = re.match(r"\w+", lhs).group(0)
            rec = calls.get(root)
            if rec is None:
                continue
            access = lhs[len(root):]
            post = rec["_post"] + access
            if access == ".1.is_none()":
                continue
            try:
                exp = parse_rhs(rhs)
            except (ValueError, AssertionError):
                continue
            rec["expect"].update(exp)
            rec["_lines"].append(lineno)
            continue
        m = re.fullmatch(r"assert!\((\w+)\.1\.is_none\(\)\)", st)
        if m:
            continue
        m = re.match(r"(?:let\s+(?:mut\s+)?)?(\w+)(?:\s*:\s*[\w<>]+)?\s*=\s*(.*)$", st, re.S)
        if m:
            name, expr = m.group(1), m.group(2).strip()
            call = find_call(expr) if not expr.startswith(('b"', '"')) else None
            if call and call[0] not in SKIP_FNS and not call[0].startswith(("EditCosts", "Some", "Vec")):
                fname, argstr, tail = call
                try:
                    args = [parse_value(a, env) for a in split_top(argstr, ",")]
                except KeyError:
                    continue
                rec = {"source": source, "fn": fname, "args": [jsonable(a) for a in args],
                       "expect": {}, "_post": tail.strip(), "_lines": [lineno]}
                if ".next()" in tail:
                    rec["first_only"] = True
                calls[name] = rec
                records.append(rec)
            elif call and call[0] == "alloc_str":
                env[name] = {"bytes": b""}
            else:
                try:
                    env[name] = parse_value(expr, env)
                except (KeyError, AttributeError, ValueError):
                    pass
            continue
        call = find_call(st)
        if call and call[0] == "fill_str":
            dst, src = split_top(call[1], ",")
            try:
                env[dst.replace("&mut", "").strip()] = parse_value(src, env)
            except KeyError:
                pass


def statements_with_lines(body: str, first_line: int):
    """Split a fn body on ';' at depth 0, tracking 1-based line numbers."""
    out, depth, cur, line, start_line, in_str, i = [], 0, [], first_line, first_line, False, 0
    while i < len(body):
        c = body[i]
        if c == "\n":
            line += 1
        if in_str:
            cur.append(c)
            if c == "\\":
                cur.append(body[i + 1]); i += 2; continue
            if c == '"':
                in_str = False
        else:
            if not "".join(cur).strip():
                start_line = line
            if c == '"':
                in_str = True; cur.append(c)
            elif c in "([{":
                depth += 1; cur.append(c)
            elif c in ")]}":
                depth -= 1; cur.append(c)
            elif c == ";" and depth == 0:
                out.append((start_line, "".join(cur))); cur = []
            else:
                cur.append(c)
        i += 1
    if "".join(cur).strip():
        out.append((start_line, "".join(cur)))
    return out


def strip_line_comments(text: str) -> str:
    return "\n".join(re.sub(r"(?<![:\"])//(?!/).*$", "", ln) if '"' not in ln.split("//")[0] or True else ln
                     for ln in text.split("\n"))


def extract_integration_tests(records):
    path = os.path.join(REF, "tests", "basic_tests.rs")
    text = open(path).read()
    for m in re.finditer(r"#\[test\]\s*fn\s+(\w+)\(\)\s*\{", text):
        name = m.group(1)
        start = m.end()
        depth, i = 1, start
        while depth:
            c = text[i]
            if c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
            i += 1
        body = text[start:i - 1]
        first_line = text.count("\n", 0, start) + 1
        run_block(statements_with_lines(body, first_line), "tests/basic_tests.rs::" + name, records)


def extract_doc_tests(records):
    for rel in ("src/lib.rs", "src/hamming.rs", "src/levenshtein.rs"):
        lines = open(os.path.join(REF, rel)).read().split("\n")
        i = 0
        while i < len(lines):
            s = lines[i].strip()
            if re.fullmatch(r"//[/!] ```", s):
                j = i + 1
                block = []
                while j < len(lines) and not re.fullmatch(r"//[/!] ```", lines[j].strip()):
                    block.append((j + 1, re.sub(r"^\s*//[/!] ?", "", lines[j])))
                    j += 1
                body = "\n".join(re.sub(r"^# ", "", t) for _, t in block)
                body = "\n".join(re.sub(r"(^|\s)//.*$", "", ln) for ln in body.split("\n"))
                run_block(statements_with_lines(body, i + 2), "%s:%d(doc)" % (rel, i + 1), records)
                i = j + 1
            else:
                i += 1


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present at %s (run in the build container)" % REF)
    records = []
    extract_integration_tests(records)
    extract_doc_tests(records)
    out = []
    for r in records:
        if not r["expect"]:
            continue
        r["source"] = "%s@L%s" % (r["source"], ",".join(str(x) for x in sorted(set(r["_lines"]))))
        r.pop("_post"); r.pop("_lines")
        out.append(r)
    with open(OUT, "w") as f:
        json.dump({"generator": "tests/golden/extract_kats.py", "reference": "triple_accel v0.4.0", "kats": out}, f, indent=1)
    by = {}
    for r in out:
        by[r["fn"]] = by.get(r["fn"], 0) + 1
    print("wrote %d KATs to %s" % (len(out), OUT))
    for k in sorted(by):
        print("  %-45s %d" % (k, by[k]))


if __name__ == "__main__":
    main()
