#!/usr/bin/env python3
"""profiles/<round>/bench_*.json (+ _pmc.json) -> the compact table README.md / DESIGN.md section 5 quote (one row per workload)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
RND = sys.argv[1] if len(sys.argv) > 1 else "r05"
D = os.path.join(ROOT, "profiles", RND)
ROWS = [("cfg2", "cfg2 1M×256 B, k=32 (the headline)"), ("cfg4", "cfg4 1M×128 B, k=8, RDAMERAU"), ("cfg3", "cfg3 100K×4 KiB `levenshtein_exp`, random pairs"),
        ("cfg3_mutated", "cfg3's shape on similar strings (mutated pairs): the `levenshtein_exp` rounds, device-driven"),
        ("cfg5", "cfg5 32 B needle, 1 GiB shard, k=16, Best"), ("cfg5w_231", "cfg5's geometry under `EditCosts(2,3,1,None)` (superset filter)"),
        ("cfg5w_2213", "cfg5's geometry under `EditCosts(2,2,1,Some(3))` (superset filter)"), ("cfg5w_231_nofilter", "the same without the filter (round 4's route, `TA_SEARCH_NOWFILTER=1`: an A/B row)"),
        ("cfg1", "cfg1 10K×1 KiB hamming (GPU batch)"), ("cfg2w", "cfg2w 1M×256 B, k=32, `EditCosts(2,3,1,None)`"),
        ("cfg2w_prefilter", "cfg2w with `TA_OPT_UNIT_PREFILTER` (random pairs: an option, not a headline)"), ("cfg2w_mutated_prefilter", "cfg2w with `TA_OPT_UNIT_PREFILTER` on mutated pairs (every pair survives the pre-pass)"),
        ("cfg4w", "cfg4w 1M×128 B, k=8, `EditCosts(2,2,1,Some(3))`"), ("cfg2l", "cfg2l 1M×256 B, k=32, `EditCosts(2,3,0,None)`"),
        ("cfg2s", "cfg2s 1M×256 B, k=32, `EditCosts(2,2,0,None)` = unit × 2"), ("cfg2t", "cfg2t 1M×256 B mutated pairs, k=32, `trace_on` for every pair (checkpoints + recomputation)"),
        ("cfg2t_own_sweep", "the same with the trace kernel's own forward sweep (`TA_TRACE_OWN_SWEEP=1`: an A/B row)"),
        ("cfg2t_dp", "cfg2t through the DP kernel's per-cell records (round 4's route, `TA_TRACE_NO_BITS=1`: an A/B row)"),
        ("cfg2_ragged", "cfg2 ragged: 1M pairs, lengths uniform on 32..256, k=32 (CSR)"), ("cfg2_dna", "cfg2 on DNA: 1M×256 B over A C G T, k=32"),
        ("cfg2_dna5", "cfg2 over A C G T N (5 symbols), k=32"), ("hsearch8", "hamming_search, 8 B needle over 1 GiB, k=2"), ("hsearch16", "hamming_search, 16 B needle over 1 GiB, k=4"),
        ("hsearch32", "hamming_search, 32 B needle over 1 GiB, k=8 (2 phases × 16 positions)"), ("hsearch32_r04", "the same on round 4's routing (`TA_HAMMING_SEARCH_NO_PHASE=1`: an A/B row)"),
        ("hsearch64", "hamming_search, 64 B needle over 1 GiB, k=16 (filter on 32 positions)"), ("hsearch64_r04", "the same on round 4's routing (an A/B row)")]


def J(n):
    try:
        return json.load(open(os.path.join(D, n)))
    except Exception:
        return None


print("| config | ms / pass | TCUPS credited / evaluated | % of 8 TB/s (algorithmic bytes) | fabric-side / algorithmic bytes | VALU instr / launch | cycles per VALU instr |")
print("|---|---|---|---|---|---|---|")
for tag, desc in ROWS:
    b = J("bench_%s.json" % tag)
    if not b:
        continue
    p = J("bench_%s_pmc.json" % tag)
    r = b["roofline"]
    ev = b.get("value_evaluated_cells")
    tr = (p or {}).get("_traffic", {}).get("bytes_per_pass")
    cyc = insts = None
    if p and "SQ_INSTS_VALU" in p and "GRBM_GUI_ACTIVE" in p:
        insts = p["SQ_INSTS_VALU"]["mean_per_launch"]
        cyc = 1024.0 * p["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0 / insts
    print("| %s | %.4f | **%.1f**%s | %.1f %% | %s | %s | %s |" % (desc, b["ms_per_step"], b["value"] / 1e3, " / %.1f" % (ev / 1e3) if ev else "", 100 * r["frac"],
          "%.2f" % (tr / r["algorithmic_bytes_per_pass"]) if tr else "—", "%.3g" % insts if insts else "—", "%.2f" % cyc if cyc else "—"))
