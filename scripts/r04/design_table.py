"""profiles/r04/*.json -> the rows of DESIGN.md section 5's table (printed as markdown; pasted by hand between the markers in DESIGN.md)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = os.path.join(ROOT, "profiles", "r04")
def J(n):
    try: return json.load(open(os.path.join(D, n)))
    except Exception: return None
try:
    ub = open(os.path.join(D, "ubench_mix.txt")).read(); sect = ub[ub.index("4 wave(s) per SIMD"):]
    MIXED = float(sect[sect.index("xor/perm strictly alternating"):].split("ms")[1].split("cycles")[0])
except Exception:
    MIXED = 4.05
ROWS = [("cfg2", "cfg2 1M×256 B, k=32 (the headline)", "bit-parallel band, stride-8 window of 33 diagonals, line-form fetch, 16 waves / CU"),
        ("cfg4", "cfg4 1M×128 B, k=8, RDAMERAU", "two pairs per lane, stride-8 window (§3.2c)"),
        ("cfg3", "cfg3 100K×4 KiB `levenshtein_exp`", "bag lower bound + row-blocked bit-parallel"),
        ("cfg5", "cfg5 32 B needle, 1 GiB shard, k=16, Best", "filter + one wavefront per flagged block incl. the Best selection; one fill, one synchronisation"),
        ("cfg1", "cfg1 10K×1 KiB hamming (GPU batch)", "hamming batch (launch-bound at 20 MB; the K passes of the timed region are one hipGraph)"),
        ("cfg2w", "cfg2w 1M×256 B, k=32, `EditCosts(2,3,1,None)`", "DP band-wavefront, 12 diagonals in one lane, affine gaps, score form"),
        ("cfg4w", "cfg4w 1M×128 B, k=8, `EditCosts(2,2,1,Some(3))`", "DP band-wavefront, 6 diagonals in one lane, affine + transposition, score form"),
        ("cfg2l", "cfg2l 1M×256 B, k=32, `EditCosts(2,3,0,None)`", "DP band-wavefront, linear gaps, score form"),
        ("cfg2s", "cfg2s 1M×256 B, k=32, `EditCosts(2,2,0,None)` = unit × 2", "bit-parallel band with k / 2 (17 diagonals) + the scaling kernel (§3.4c)"),
        ("cfg2t", "cfg2t 1M×256 B mutated pairs, k=32, `trace_on` for every pair", "DP band trace kernel (34 diagonals in one lane) + the walk kernel (§3.4b)"),
        ("cfg2_ragged", "cfg2 ragged: 1M pairs, lengths uniform on 32..256, k=32 (CSR)", "counting sort, longest first + stride-8 window, chunk-form fetch"),
        ("cfg2_ragged_vline", "the same through the VLINE fetch form (`TA_BITS_VLINE=1`, an A/B row)", "counting sort on exact lengths + stride-8 window, VLINE fetch"),
        ("cfg2_dna", "cfg2 on DNA: 1M×256 B over A C G T, k=32", "small-alphabet kernel (§3.2d)"),
        ("cfg2_dna5", "cfg2 over A C G T N (5 symbols), k=32", "5-bit-code small-alphabet kernel (§3.2d; the default for ≤ 4 groups of codes at ≥ 16 diagonals)"),
        ("cfg2_protein_table", "cfg2 over the 20 amino acids, k=32, FORCED through the 5-bit-code kernel (`TA_BITSQ_WIDE=1`, an A/B row: the default is the byte test, 0.290)", "5-bit-code small-alphabet kernel"),
        ("hsearch8", "hamming_search, 8 B needle over 1 GiB, k=2", "SWAR, 16 offsets per lane, NUL scan fused"),
        ("hsearch32", "hamming_search, 32 B needle over 1 GiB, k=8", "bit-sliced counters (4 planes), NUL scan fused"),
        ("hsearch64", "hamming_search, 64 B needle over 1 GiB, k=16", "SWAR, 16 offsets per lane")]
README = "--readme" in sys.argv
if README:
    print("| config | ms / pass | TCUPS credited / evaluated | % of 8 TB/s (algorithmic bytes) | fabric-side / algorithmic bytes | cycles per VALU instr |")
    print("|---|---|---|---|---|---|")
else:
  print("| Config | Kernel(s) | ms / pass (wall, driver protocol) | TCUPS credited / evaluated | Algorithmic GB/s (% of 8 TB/s) | Fabric-side bytes / algorithmic | VALU instr / launch | Cycles / VALU instr (of the 2-cycle ceiling; of the mixed-stream rate) |")
  print("|---|---|---|---|---|---|---|---|")
for tag, name, kern in ROWS:
    b, p = J("bench_%s.json" % tag), J("bench_%s_pmc.json" % tag)
    if not b: continue
    r = b["roofline"]
    ev = b.get("value_evaluated_cells")
    tr = (p or {}).get("_traffic", {}).get("bytes_per_pass")
    v = ""
    insts = "—"
    if p and "SQ_INSTS_VALU" in p and "GRBM_GUI_ACTIVE" in p:
        i_, busy = p["SQ_INSTS_VALU"]["mean_per_launch"], p["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
        cyc = 1024.0 * busy / i_
        v = "%.2f (%.2f; %.2f)" % (cyc, 2.0 / cyc, MIXED / cyc)
        insts = "%.3g" % i_
    if README:
        print("| %s | %.4f | **%.1f**%s | %.1f %% | %s | %s |" % (name, b["ms_per_step"], b["value"] / 1e3, (" / %.1f" % (ev / 1e3)) if ev else "", 100 * r["frac"],
              ("%.2f" % (tr / r["algorithmic_bytes_per_pass"])) if tr else "—", v.split(" ")[0] if v else "—"))
        continue
    print("| %s | %s | **%.4f** | **%.1f**%s | %.0f (%.1f %%) | %s | %s | %s |" % (
        name, kern, b["ms_per_step"], b["value"] / 1e3, (" / %.1f" % (ev / 1e3)) if ev else "", r["achieved"], 100 * r["frac"],
        ("%.2f" % (tr / r["algorithmic_bytes_per_pass"])) if tr else "—", insts, v or "—"))
