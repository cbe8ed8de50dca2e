"""Runs the reference's known-answer tests (tests/golden/kats.json) against a backend.

A backend is any object exposing the reference's function names with Python
values (bytes in, ints / tuples / lists out):
    hamming(a, b) -> int
    hamming_search_with_opts(needle, haystack, k, search_type) -> [(start, end, k)]
    levenshtein_full(a, b, trace_on, costs) -> (dist, trace|None)          # unbounded distance
    levenshtein_k_with_opts(a, b, k, trace_on, costs) -> (dist|None, trace|None)
    levenshtein(a, b), rdamerau(a, b), levenshtein_exp(a, b), rdamerau_exp(a, b) -> int
    levenshtein_exp_with_opts(a, b, trace_on, costs) -> (dist, trace|None)
    levenshtein_search_with_opts(needle, haystack, k, search_type, costs, anchored) -> [(start, end, k)]
    default_search_k(n) -> int
and an attribute `supports_trace` (bool).
search_type: 0 = All, 1 = Best.  costs: (mismatch, gap, start_gap, transpose|None).
"""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LEV = (1, 1, 0, None)
ST = {"All": 0, "Best": 1}


def load_kats():
    with open(os.path.join(_HERE, "golden", "kats.json")) as f:
        return json.load(f)["kats"]


def _b(x):
    return bytes.fromhex(x["hex"])


def _c(x):
    return (x["mismatch"], x["gap"], x["start_gap"], x["transpose"])


def needs_trace(kat):
    return "trace" in kat["expect"] or any(a is True for a in kat["args"][2:5] if isinstance(a, bool))


def run_kat(be, kat):
    """Returns (got, want) in a comparable normal form."""
    fn, args, exp = kat["fn"], kat["args"], kat["expect"]
    want_trace = [(e["edit"], e["count"]) for e in exp["trace"]] if "trace" in exp else None

    def dist_result(d, tr):
        got = {"value": d}
        want = {"value": None if exp.get("none") else exp.get("value")}
        if want_trace is not None:
            got["trace"] = tr
            want["trace"] = want_trace
        return got, want

    def match_result(ms):
        if "first_match" in exp:
            m = exp["first_match"]
            return ms[0], (m["start"], m["end"], m["k"])
        return ms, [(m["start"], m["end"], m["k"]) for m in exp["matches"]]

    if fn in ("hamming", "hamming_naive", "hamming_simd_movemask", "hamming_simd_parallel",
              "hamming_words_64", "hamming_words_128"):
        return {"value": be.hamming(_b(args[0]), _b(args[1]))}, {"value": exp["value"]}
    if fn in ("hamming_search", "hamming_search_naive", "hamming_search_simd"):
        n, h = _b(args[0]), _b(args[1])
        return match_result(be.hamming_search_with_opts(n, h, be.default_search_k(len(n)), 1))
    if fn in ("hamming_search_naive_with_opts", "hamming_search_simd_with_opts"):
        return match_result(be.hamming_search_with_opts(_b(args[0]), _b(args[1]), args[2], ST[args[3]]))
    if fn == "levenshtein_naive":
        return dist_result(*be.levenshtein_full(_b(args[0]), _b(args[1]), False, LEV))
    if fn == "levenshtein_naive_with_opts":
        return dist_result(*be.levenshtein_full(_b(args[0]), _b(args[1]), args[2], _c(args[3])))
    if fn in ("levenshtein_naive_k", "levenshtein_simd_k"):
        return dist_result(*be.levenshtein_k_with_opts(_b(args[0]), _b(args[1]), args[2], False, LEV))
    if fn in ("levenshtein_naive_k_with_opts", "levenshtein_simd_k_with_opts"):
        return dist_result(*be.levenshtein_k_with_opts(_b(args[0]), _b(args[1]), args[2], args[3], _c(args[4])))
    if fn in ("levenshtein", "rdamerau", "levenshtein_exp", "rdamerau_exp"):
        return {"value": getattr(be, fn)(_b(args[0]), _b(args[1]))}, {"value": exp["value"]}
    if fn == "levenshtein_exp_with_opts":
        return dist_result(*be.levenshtein_exp_with_opts(_b(args[0]), _b(args[1]), args[2], _c(args[3])))
    if fn in ("levenshtein_search", "levenshtein_search_naive", "levenshtein_search_simd"):
        n, h = _b(args[0]), _b(args[1])
        return match_result(be.levenshtein_search_with_opts(n, h, be.default_search_k(len(n)), 1, LEV, False))
    if fn in ("levenshtein_search_naive_with_opts", "levenshtein_search_simd_with_opts"):
        return match_result(be.levenshtein_search_with_opts(_b(args[0]), _b(args[1]), args[2], ST[args[3]],
                                                            _c(args[4]), args[5]))
    raise KeyError(fn)
