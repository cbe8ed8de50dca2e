#!/usr/bin/env python3
"""profiles/<round>/*.json (written by scripts/gpu_profiles.sh on one GPU box) -> profiles/<round>/SUMMARY.md: the tables README.md and
DESIGN.md quote.  Nothing is typed by hand: every figure is a field of a committed file, named in the table's last column."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r06"
D = os.path.join(ROOT, "profiles", RND)


def J(name):
    try:
        return json.load(open(os.path.join(D, name)))
    except Exception:
        return None


def kernel_us(wl, needle):
    """average duration (us) and calls of the kernel whose name contains `needle`, from rocprofv3 --kernel-trace --stats"""
    try:
        rows = list(csv.DictReader(open(os.path.join(D, "bench_%s_kernel_stats.csv" % wl))))
    except Exception:
        return None
    for r in rows:
        if needle in r.get("Name", ""):
            return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    return None


def mixed_rate():
    """cycles per VALU instruction of a strictly alternating full-rate / half-rate stream at 4 waves per SIMD (ubench_mix.txt)"""
    try:
        ub = open(os.path.join(D, "ubench_mix.txt")).read()
        sect = ub[ub.index("4 wave(s) per SIMD"):]
        return float(sect[sect.index("xor/perm strictly alternating"):].split("ms")[1].split("cycles")[0])
    except Exception:
        return None


MIXED = mixed_rate()


def main():
    out = ["# profiles/%s — measured on one MI355X box by `scripts/gpu_profiles.sh`, tabulated by `scripts/make_tables.py`" % RND, ""]
    out += ["Credited cells = the cells the reference's scalar band visits (SURVEY.md 8d); evaluated = the cells inside the half band the kernels "
            "compute (DESIGN.md 3.1).  HBM fraction = algorithmic bytes / device time / 8 TB/s.  Fabric-side bytes = 128 x `TCC_EA0_RDREQ_128B` + "
            "64 x `TCC_EA0_RDREQ_64B` + `WRITE_SIZE`, every kernel of one pass.  VALU cycles per instruction = 1024 SIMDs x `GRBM_GUI_ACTIVE`/8 / "
            "`SQ_INSTS_VALU` of the dominant kernel.", ""]
    out += ["| config | GCUPS credited | GCUPS evaluated | ms / pass (wall, driver protocol) | device ms / pass | algorithmic GB/s | % of 8 TB/s | fabric-side bytes / algorithmic | "
            "VALU instr / launch | cycles / VALU instr | of the 2-cycle ceiling | of the measured mixed-stream rate | files |", "|" + "---|" * 13]
    for wl in ("cfg2", "cfg2_mutated", "cfg4", "cfg4_mutated", "cfg3", "cfg3_mutated", "cfg3_mutated_host_rounds", "cfg5", "cfg5w_220", "cfg5w_231", "cfg5w_2213", "cfg5w_1101",
               "cfg5w_231_nofilter", "cfg1", "cfg2w", "cfg2w_prefilter", "cfg2w_mutated", "cfg2w_mutated_prefilter", "cfg4w", "cfg4w_prefilter", "cfg2l", "cfg2s", "cfg2t", "cfg2tp", "cfg2tp_stile128", "cfg2t_220", "cfg2tp_220", "cfg2t_231", "cfg2t_2213", "cfg2t_own_sweep", "cfg2t_dp",
               "cfg2_ragged", "cfg2_ragged_vline", "cfg2_dna",
               "cfg2_dna5", "cfg2_protein_table", "hsearch8", "hsearch16", "hsearch32", "hsearch32_r04", "hsearch64", "hsearch64_r04", "cfg2_early_out", "cfg2_2m"):
        b = J("bench_%s.json" % wl)
        if not b:
            continue
        base = wl
        p = J("bench_%s_pmc.json" % wl)
        r = b["roofline"]
        ev = b.get("value_evaluated_cells")
        tr = (p or {}).get("_traffic", {}).get("bytes_per_pass")
        v = {}
        if p and "SQ_INSTS_VALU" in p and "GRBM_GUI_ACTIVE" in p:
            insts, busy = p["SQ_INSTS_VALU"]["mean_per_launch"], p["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
            cyc = 1024.0 * busy / insts
            v = {"insts": insts, "cyc": cyc, "f2": 2.0 / cyc, "fm": MIXED / cyc if MIXED else None, "clock": busy / (kernel_us(base, p["_dominant"].split("::")[-1].split("<")[0]) or (1e9, 0))[0] / 1e3}
        out.append("| %s | %.0f | %s | %.4f | %.4f | %.0f | %.1f %% | %s | %s | %s | %s | %s | `bench_%s.json`%s |" % (
            wl, b["value"], "%.0f" % ev if ev else "—", b["ms_per_step"], r["device_ms_per_pass"], r["achieved"], 100 * r["frac"],
            "%.2f" % (tr / r["algorithmic_bytes_per_pass"]) if tr else "—",
            "%.3g" % v["insts"] if v else "—", "%.2f" % v["cyc"] if v else "—",
            "%.2f" % v["f2"] if v else "—", "%.2f" % v["fm"] if v and v["fm"] else "—",
            wl, ", `bench_%s_pmc.json`" % base if p else ""))
    out += ["", "(cfg2w / cfg4w: the cfg2 / cfg4 geometry under EditCosts(2,3,1,None) k=32 / (2,2,1,Some(3)) k=8 -- the DP band-wavefront kernel; "
            "cfg2_ragged: CSR batch, lengths uniform on 32..256, taken in length order on the device, cells credited pair by pair; cfg2_dna: strings "
            "over A C G T through the small-alphabet kernel; cfg2_early_out: `ta_set_option(TA_OPT_EARLY_OUT)` on the random batch -- same answers, "
            "data-dependent work, NOT a headline figure; cfg2_2m: 2M pairs per pass; cfg2l / cfg2s: cfg2's geometry under EditCosts(2,3,0,None) -- DP band kernel, linear "
            "gaps -- and (2,2,0,None) = unit costs x 2 on the bit-parallel kernel; cfg2t: every pair traced, ta_levenshtein_trace_batch (bytes: strings + records are "
            "NOT counted, only strings, distances and the scripts written); cfg2_ragged_vline: the ragged batch through the VLINE fetch form, an A/B row; cfg2_dna5: strings over A C G T N -- the 5-bit-code small-alphabet kernel, the default there; cfg2_protein_table: the 20 amino acids FORCED through that kernel (TA_BITSQ_WIDE=1), an A/B row -- the default runs the byte test at cfg2's rate, ab_alphabet.md; "
            "hsearchN: hamming_search of an N-byte needle over 1 GiB, k = N / 4, cells = N byte compares per offset.  Round 5: cfg3_mutated / cfg4_mutated: similar strings "
            "(cfg3_mutated: the levenshtein_exp rounds, device-driven; cfg3_mutated_host_rounds: TA_EXP_HOST_ROUNDS=1, an A/B row); cfg5w_<costs>: levenshtein_search of cfg5's "
            "geometry under EditCosts(2,2,0,None) / (2,3,1,None) / (2,2,1,Some(3)) / RDAMERAU_COSTS through the superset filter, cfg5w_231_nofilter: round 4's route "
            "(TA_SEARCH_NOWFILTER=1, an A/B row); cfgNw_prefilter / cfg2w_mutated_prefilter: ta_set_option(TA_OPT_UNIT_PREFILTER) -- same answers, data-dependent work, NOT a "
            "headline figure; cfg2t: the checkpoint-and-recompute kernel, its forward sweep done by the distance pass (cfg2t_own_sweep: the trace kernel's own forward sweep, TA_TRACE_OWN_SWEEP=1; cfg2t_dp: the DP kernel's records, TA_TRACE_NO_BITS=1: A/B rows); hsearchN_r04: round 4's routing "
            "(TA_HAMMING_SEARCH_NO_PHASE=1, A/B rows).)", ""]
    out += ["(Round 6: cfg2tp: cfg2t with PACKED records (ta_levenshtein_trace_batch_packed: 4-byte runs written in place by the walk); cfg2tp_stile128: the same with 128 columns per "
            "string fetch (TA_TRACE_STILE=128, an A/B row); cfg2t_220 / cfg2tp_220: EditCosts(2,2,0,None) -- unit costs x 2 on the checkpoint route with k / 2; cfg2t_231 / cfg2t_2213: "
            "EditCosts(2,3,1,None) / (2,2,1,Some(3)), k = 32 -- the DP band kernel's records + the walk kernel; cfg2t_dp: unit costs forced through that route; cfg3_mutated: the first "
            "threshold is the stride-8 window's k = 32.)", ""]
    b2 = J("bench_cfg2.json")
    if b2 and b2.get("all_configs"):
        out += ["## The driver's one line: every BASELINE config (`bench_cfg2.json` `all_configs`; each leg its own bench.py process behind its own parity gate, %s s in all)" % b2.get("all_configs_seconds"), "",
                "| workload | ms / pass | GCUPS | % of 8 TB/s | cycles per VALU instr | fabric-side / algorithmic | parity: answers that were numbers | kernel |", "|" + "---|" * 8]
        for r in b2["all_configs"]:
            if "error" in r:
                out.append("| %s | FAILED: %s |" % (r["workload"], r["error"][:120]))
                continue
            f = lambda v, fmt: (fmt % v) if v is not None else "—"
            out.append("| %s | %.4f | %.0f | %.1f %% | %s | %s | %s | `%s`%s |" % (r["workload"][:80], r["ms_per_step"], r["value"], 100 * r["roofline_frac"], f(r.get("cycles_per_valu_inst"), "%.2f"),
                                                                                 f(r.get("traffic_ratio"), "%.2f"), r.get("parity_checked_some"), r.get("kernel_name"), " — " + r["note"] if r.get("note") else ""))
        out.append("")
    ov = [(wl, J("bench_%s.json" % wl)) for wl in ("cfg2", "cfg4", "cfg2w", "cfg4w", "cfg2l", "cfg2s")]
    if any(b and b.get("overlapped_passes") for _, b in ov):
        out += ["## Consecutive passes on two streams (`overlapped_passes`: disclosed, never the headline -- DESIGN.md section 8, round 6 item 7)", "",
                "| config | device ms / pass, one stream | two streams | GCUPS credited, two streams |", "|---|---|---|---|"]
        for wl, b in ov:
            if b and b.get("overlapped_passes"):
                o = b["overlapped_passes"]
                out.append("| %s | %.4f | %.4f | %.0f |" % (wl, b["roofline"]["device_ms_per_pass"], o["device_ms_per_pass"], o["value"]))
        out.append("")
    sp = [(n, sc, J("bench_sp_cfg2_%s_n%d.json" % (sc, n))) for sc in ("weak", "strong") for n in (1, 2, 8)]
    if any(b for _, _, b in sp):
        out += ["## The device set: one process, `bench.py --gpus N --single-process` (THIS box has one GPU: device 0 listed N times -- N workers share it and its one PCIe link)", "",
                "| workload | N | scaling | pairs | ms / pass (wall) | slowest shard's device ms | GCUPS credited | end to end ms (host strings in, answers out) |", "|" + "---|" * 8]
        for n, sc, b in sp:
            if b:
                out.append("| cfg2 | %d | %s | %d | %.4f | %.4f | %.0f | %.2f |" % (n, sc, b["config"]["units_total"], b["ms_per_step"], b["roofline"]["device_ms_per_pass"], b["value"], b["end_to_end_ms"]))
        for n in (1, 8):
            b = J("bench_sp_cfg5_n%d.json" % n)
            if b:
                out.append("| cfg5 (resident sharded haystack, Best) | %d | weak | %d | %.4f | — | %.0f | %s |" % (n, b["config"]["units_total"], b["ms_per_step"], b["value"], "%.2f" % b["end_to_end_ms"] if b.get("end_to_end_ms") else "—"))
        out.append("")
    e2e = [(wl, J("bench_%s.json" % wl)) for wl in ("cfg2", "cfg4", "cfg5", "cfg1", "cfg2_ragged")]
    out += ["## Host buffers in, answers out (`end_to_end_ms`: pinned H2D of the batch + the pass + D2H; never the headline)", "",
            "| config | ms per pass, inputs resident | ms end to end | untimed ramp passes before the timed region |", "|---|---|---|---|"]
    for wl, b in e2e:
        if b and b.get("end_to_end_ms"):
            out.append("| %s | %.4f | %.2f | %s |" % (wl, b["ms_per_step"], b["end_to_end_ms"], b.get("prewarm_passes")))
    out.append("")
    b2 = J("bench_cfg2.json")
    if b2 and b2.get("cpu_baseline"):
        c = b2["cpu_baseline"]
        out += ["## CPU baseline of the same run (cfg2, `bench_cfg2.json`)", "",
                "%.1f GCUPS, effective cores %.1f, threads used %s; host: %s" % (c["value"], c["cores"], c.get("threads_used"), json.dumps(c.get("host"))), "",
                "| restatement | GCUPS all threads | GCUPS one thread |", "|---|---|---|"]
        for nm, (va, v1) in c.get("variants_gcups_all_threads_and_one_thread", {}).items():
            out.append("| %s | %.2f | %.3f |" % (nm, va, v1))
        out += ["", "thread scaling of the best variant (threads, GCUPS): " + ", ".join("%d: %.1f" % (t, g) for t, g in c.get("thread_scaling_gcups", [])), ""]
    out += ["## Dominant kernels (rocprofv3 --kernel-trace --stats, `bench_cfgN_kernel_stats.csv`)", "", "| config | kernel | calls | average us |", "|---|---|---|---|"]
    for wl, needle in (("cfg2", "lev_bits_"), ("cfg4", "lev_bits"), ("cfg3", "lev_widebits_kernel"), ("cfg3", "bag_bound"), ("cfg5", "lev_filter_kernel"),
                       ("cfg5", "lev_search_wave_kernel"), ("cfg1", "hamming"), ("cfg2w", "lev_band_kernel"), ("cfg4w", "lev_band_kernel"),
                       ("cfg2_ragged", "lev_bits_"), ("cfg2_ragged", "len_hist"), ("cfg2_ragged", "len_scan"), ("cfg2_ragged", "len_scatter"),
                       ("cfg2_dna", "lev_bitsq_kernel"), ("cfg2_dna", "lev_bits_s8"), ("cfg2_dna5", "lev_bitsqw_kernel"), ("cfg2_protein_table", "lev_bitsqw_kernel"), ("cfg2_ragged_vline", "lev_bits_s8v"), ("cfg2_ragged_vline", "len_scatter"),
                       ("cfg2l", "lev_band_score"), ("cfg2s", "lev_bits_line"), ("cfg2s", "scale_results"), ("cfg2t", "lev_band_trace_kernel"), ("cfg2t", "lev_trace_walk"),
                       ("cfg2t", "lev_bits_trace_kernel"), ("cfg2t", "lev_bits_s8_ckpt_kernel"), ("cfg2tp", "lev_bits_trace_kernel"), ("cfg2tp", "lev_bits_s8_ckpt_kernel"), ("cfg2t_231", "lev_band_trace_kernel"), ("cfg2t_231", "lev_trace_walk"), ("cfg2t_dp", "lev_band_trace_kernel"), ("cfg2t_dp", "lev_trace_walk"), ("cfg3_mutated", "lev_bits_"), ("cfg3_mutated", "lev_widebits_kernel"), ("cfg3_mutated", "compact_"),
                       ("cfg3_mutated", "bag_bound"), ("cfg4_mutated", "lev_bits2"), ("cfg5w_231", "lev_filter_kernel"), ("cfg5w_231", "lev_search_wave_kernel"),
                       ("cfg5w_2213", "lev_filter_kernel"), ("cfg5w_2213", "lev_search_wave_kernel"),
                       ("cfg2w_prefilter", "lev_bits_"), ("cfg2w_prefilter", "compact_some"), ("cfg2w_prefilter", "lev_band_score"),
                       ("cfg2w_mutated_prefilter", "lev_bits_"), ("cfg2w_mutated_prefilter", "lev_band_score"),
                       ("hsearch8", "hamming_search"), ("hsearch16", "hamming_search"), ("hsearch32", "hamming_search"), ("hsearch64", "hamming_search")):
        k = kernel_us(wl, needle)
        if k:
            out.append("| %s | `%s` | %d | %.1f |" % (wl, needle, k[1], k[0]))
    mix = J("isa_mix.json")
    if mix:
        out += ["", "## Inner loops (`isa_mix.json`, `inner_loop_cfgN.s`; scripts/isa_mix.py)", "",
                "| config | kernel | instructions | VALU | full-rate / bitop3 / half-rate | s_nop | LDS | modelled cycles per VALU instr (per-opcode costs) | VGPRs |", "|" + "---|" * 9]
        for wl, m in mix.items():
            bc = m["by_class"]
            out.append("| %s | `%s` | %d | %d | %d / %d / %d | %d | %d | %.2f | %d |" % (wl, m["kernel"][:70], m["instructions"], m["valu"], bc.get("full", 0),
                                                                                         bc.get("bitop3", 0), bc.get("half", 0), m["s_nop"], m["lds"],
                                                                                         m["modelled_cycles_per_valu_inst"], m.get("vgprs", 0)))
    open(os.path.join(D, "SUMMARY.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
