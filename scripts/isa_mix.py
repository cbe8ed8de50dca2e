#!/usr/bin/env python3
"""Opcode histogram and issue-cycle model of a kernel's inner loop, from hipcc's own assembly.

    python scripts/isa_mix.py [--json out.json] [--dump-dir DIR]

For each BASELINE configuration's dominant kernel: compile its translation unit to gfx950 assembly (-S), take the
innermost loop with the most VALU instructions (all its basic blocks, minus the ragged-tail variants),
count its instructions by opcode, and price every VALU opcode with the issue cost MEASURED on MI355X by
scripts/ubench_valu*.hip (profiles/r01/ubench_valu_issue_rates*.txt, cycles per wave-instruction per SIMD at >= 2
waves/SIMD):
    full-rate class  2.3  v_add/sub/and/or/xor/mov/not/lshrrev (plain VOP1/VOP2 integer ops)
    v_bitop3_b32     2.8
    half-rate class  4.2  everything else measured: v_perm, v_dot4*, v_alignbit/byte, v_lshlrev, v_bfe, v_bcnt, v_min/max,
                          v_cmp, v_cndmask (sgpr mask), v_add_co/v_addc_co, v_and_or, v_lshl_or/add, DPP/SDWA forms, v_pk_*16
The modelled cycles per iteration = sum(count x class cost) is the MIX-SPECIFIC issue ceiling of that loop; the guide's
hard ceiling is 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md, SIMD-32).  bench.py reports the measured
cycles per VALU instruction (SQ_INSTS_VALU, GRBM_GUI_ACTIVE) against both.
"""
import argparse
import json
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "triple_accel_amd", "csrc")

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_not_b32", "v_lshrrev_b32",
        "v_add_f32", "v_mul_f32", "v_min_u16", "v_add_u16", "v_xnor_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_nop"}
SPECIAL = {"v_bitop3_b32": 2.8, "v_fma_f32": 2.9, "ds_bpermute_b32": 24.0}
HALF = 4.2

# (config, translation unit, mangled-name regex of the dominant kernel, columns / steps one iteration of the block advances)
KERNELS = [
    ("cfg2", "lev_bits.hip", r"_ZN2ta18lev_bits_s8_kernelILb0ELb1ELb0E\w*", "8 columns of 64 pairs (33-diagonal band, stride-8 window, line form)", "all_live"),
    ("cfg4", "lev_bits.hip", r"_ZN2ta16lev_bits2_kernelILb1ELb0E\w*", "the span loop: 16 columns of 128 pairs with their commits (11-diagonal band + transposition, two pairs per lane, stride-8 window)"),
    ("cfg2w", "lev_band_score.hip", r"_ZN2ta21lev_band_score_kernelILi12ELb1ELi0ELb1E\w*", "four iterations = eight anti-diagonals = four columns of 64 pairs x 12 diagonals (affine gaps, score form)"),
    ("cfg4w", "lev_band_score.hip", r"_ZN2ta21lev_band_score_kernelILi6ELb1ELi1ELb1E\w*", "four iterations = eight anti-diagonals = four columns of 64 pairs x 6 diagonals (affine gaps, transposition)"),
    ("cfg2_dna", "lev_bitsq.hip", r"_ZN2ta16lev_bitsq_kernelILb0E\w*", "the span loop, full and cut-short copies: 2 x 16 columns of 64 pairs with their conversions (33-diagonal band, match vectors from per-symbol tables)"),
    ("cfg3", "lev_widebits.hip", r"_ZN2ta19lev_widebits_kernelILi2ELb0E\w*", "two steps of the sweep: 2 x (64 lanes x 64 rows) cells of one pair (one pair per wavefront)", "first_dpp"),
    ("cfg5", "lev_search.hip", r"_ZN2ta17lev_filter_kernelILb0ELb1ELb0E\w*", "haystack bytes per lane (bit-parallel filter scan)"),
]


def cost(op):
    base = re.sub(r"_(e32|e64)$", "", op)
    if base.endswith("_dpp") or base.endswith("_sdwa"):
        return HALF, "half"
    if base in SPECIAL:
        return SPECIAL[base], "bitop3" if base == "v_bitop3_b32" else base
    if base in FULL:
        return 2.3, "full"
    if base.startswith("v_"):
        return HALF, "half"
    return 0.0, "other"            # SALU / LDS / VMEM / waitcnt issue on other ports


def kernel_body(asm, name_re):
    m = re.search(r"^(%s):" % name_re, asm, re.M)
    if not m:
        return None, None
    i = m.start()
    j = asm.index(".Lfunc_end", i)
    return m.group(1), asm[i:j]


def blocks_of(body):
    body = re.sub(r"\n; %bb\.(\d+):", lambda m: "\n.LBB0_%s:" % m.group(1), body)     # fall-through blocks carry no label of their own
    parts = re.split(r"\n(\.LBB\d+_\d+):", body)
    out = []
    for k in range(1, len(parts), 2):
        ins = []
        for l in parts[k + 1].split("\n"):
            if not l.startswith("\t"):
                continue
            t = l.split()
            if not t or t[0].startswith((";", ".")):
                continue
            ins.append(t[0])
        depth = 0
        md = re.findall(r"Depth=(\d+)", parts[k + 1][:400])
        if md:
            depth = max(int(x) for x in md)
        out.append((parts[k], ins, depth, parts[k + 1]))
    return out


def hot_loop(body, pick="largest"):
    """The innermost loop with the most VALU work: all blocks LLVM annotates with the same deepest `Header=`, plus that header.
    Blocks of the loop that carry the per-column liveness select of the ragged tail (a v_cndmask_b32 fed by a v_cmp -- only the
    last chunk of a batch takes them) are left out, so the histogram is the path a full chunk runs."""
    bl = blocks_of(body)
    groups = {}
    depth_of = {}
    for lab, ins, depth, text in bl:
        m = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", text[:600])
        hdr = None
        if "This Inner Loop Header" in text[:600]:
            hdr = lab.lstrip(".L")
        elif m:
            hdr = m.group(1)
        if hdr:
            groups.setdefault(hdr, []).append((lab, ins, text))
            dm = re.search(r"Depth=(\d+)", text[:600])
            if dm and "Inner Loop Header" in text[:600] or (m and lab.lstrip(".L") != hdr):
                depth_of[hdr] = max(depth_of.get(hdr, 0), int((dm or m).group(1) if dm else m.group(2)))
    def valu(g):      # blocks with next to no VALU work (switch arms, loop latches) do not make a loop "hot"
        per = [sum(1 for x in ins if x.startswith("v_")) for _, ins, _ in g]
        return sum(v for v in per if v >= 12)
    hdr = max(groups, key=lambda h: valu(groups[h]))
    if pick == "deepest":
        # a loop nest whose outer loops carry more code than the innermost one: the innermost (deepest) loop with the most VALU work
        dmax = max(depth_of.get(h, 0) for h in groups)
        hdr = max((h for h in groups if depth_of.get(h, 0) == dmax), key=lambda h: sum(1 for _, ins, _ in groups[h] for x in ins if x.startswith("v_")))
    if pick == "first_dpp":
        # lev_widebits: six instantiations of the sweep loop (one per role of the stripe) next to the table-building loop, all at the
        # same depth; a single-stripe pair (cfg3) runs the first sweep loop in program order -- the first loop with a DPP hand-off
        dmax = max(depth_of.get(h, 0) for h in groups)
        hdr = next(h for h in groups if depth_of.get(h, 0) == dmax and any(x.endswith("_dpp") for _, ins, _ in groups[h] for x in ins))
    if pick == "all_live":
        # the kernel keeps its whole blocks (every pair live) and its capped / cut-short blocks in separate loops of about the same
        # size: the BASELINE configuration runs the former -- the one with the fewest selects (the capped loop has one more
        # v_cndmask per column, the liveness select)
        top = valu(groups[hdr])
        def selects(g):
            return sum(1 for _, ins, _ in g for x in ins if x.startswith("v_cndmask"))
        hdr = min((h for h in groups if valu(groups[h]) >= 0.8 * top), key=lambda h: selects(groups[h]))
    if pick == "smallest_hot":
        # the kernel holds several copies of its inner loop (chunk form, line form specialised by the answer's word): the BASELINE
        # configuration runs the specialised line-form copy -- the leanest of the hot loops
        top = valu(groups[hdr])
        hdr = min((h for h in groups if valu(groups[h]) >= 0.8 * top), key=lambda h: valu(groups[h]))
    used, skipped, all_ins, texts = [], [], [], []
    def nvalu(ins):
        return sum(1 for x in ins if x.startswith("v_"))
    def ragged(ins):
        return any(x.startswith("v_cndmask") for x in ins) and any(x.startswith("v_cmp") for x in ins)
    plain = [nvalu(ins) for _, ins, _ in groups[hdr] if not ragged(ins)]
    # twins without a visible select (the compiler folded it into a bitop): the loop holds every column block twice, in the same
    # order -- second half = the ragged-tail copies of the first.  Recognised by the big blocks pairing up in size.
    big = [(lab, nvalu(ins)) for lab, ins, _ in groups[hdr] if nvalu(ins) >= 30]
    twin_skip = set()
    if len(big) >= 8 and len(big) % 2 == 0 and not any(ragged(ins) for _, ins, _ in groups[hdr]):
        sizes = sorted(v for _, v in big)
        if all(abs(sizes[i] - sizes[i + 1]) <= 3 for i in range(0, len(sizes), 2)):
            seen = {}
            for lab, v in big:                        # keep the first block of every size class pair, drop its twin
                key = min(seen, key=lambda k: abs(k - v)) if seen and min(abs(k - v) for k in seen) <= 3 else None
                if key is not None and seen[key] > 0:
                    seen[key] -= 1
                    twin_skip.add(lab)
                else:
                    seen[v] = seen.get(v, 0) + 1
    for lab, ins, text in groups[hdr]:
        # a block with the liveness select is left out only if the loop also holds its plain twin (a block of comparable size without it)
        if (ragged(ins) and any(v >= 0.5 * nvalu(ins) for v in plain)) or lab in twin_skip:
            skipped.append(lab)
            continue
        used.append(lab)
        all_ins += ins
        texts.append("%s:%s" % (lab, text))
    return "+".join(used), all_ins, skipped, "\n".join(texts)


def analyse(label, ins):
    c = Counter(re.sub(r"_(e32|e64)$", "", x) for x in ins)
    classes = Counter()
    cycles = 0.0
    nvalu = 0
    for op, n in c.items():
        w, cl = cost(op)
        if op.startswith("v_") or op == "ds_bpermute_b32":
            classes[cl] += n
            cycles += w * n
            if op.startswith("v_"):
                nvalu += n
    return {"block": label, "instructions": len(ins), "valu": nvalu, "by_class": dict(classes),
            "s_nop": c.get("s_nop", 0), "lds": sum(n for op, n in c.items() if op.startswith("ds_")),
            "vmem": sum(n for op, n in c.items() if op.startswith(("global_", "buffer_", "flat_"))),
            "salu": sum(n for op, n in c.items() if op.startswith("s_") and op != "s_nop"),
            "modelled_issue_cycles": round(cycles, 1),
            "modelled_cycles_per_valu_inst": round(cycles / max(nvalu, 1), 3),
            "histogram": dict(c.most_common())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--dump-dir", help="write each kernel's hot block (assembly text) here")
    args = ap.parse_args()
    asm_cache = {}
    result = {}
    for entry in KERNELS:
        cfg, tu, name_re, what = entry[:4]
        pick = entry[4] if len(entry) > 4 else "largest"
        if tu not in asm_cache:
            out = "/tmp/isa_mix_%s.s" % tu.replace(".hip", "")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                                   "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", os.path.join(CSRC, tu), "-o", out],
                                  cwd=CSRC)
            asm_cache[tu] = open(out).read()
        name, body = kernel_body(asm_cache[tu], name_re)
        if body is None:
            print("%s: kernel %s not found in %s" % (cfg, name_re, tu), file=sys.stderr)
            continue
        lab, ins, skipped, text = hot_loop(body, pick)
        r = analyse(lab, ins)
        r["blocks_left_out_ragged_tail"] = skipped
        r["kernel"] = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        r["iteration"] = what
        r["vgprs"] = int((re.findall(r"; NumVgprs: (\d+)", asm_cache[tu][asm_cache[tu].index(name + ":"):]) or ["0"])[0])
        result[cfg] = r
        print("%s  %s  block %s: %d instr, %d VALU (%s), %d s_nop, %d LDS, %d VMEM -> %.0f modelled issue cycles, %.2f per VALU instr"
              % (cfg, r["kernel"][:60], lab, r["instructions"], r["valu"],
                 ", ".join("%s %d" % kv for kv in sorted(r["by_class"].items())), r["s_nop"], r["lds"], r["vmem"],
                 r["modelled_issue_cycles"], r["modelled_cycles_per_valu_inst"]))
        print("      " + ", ".join("%d %s" % (n, op) for op, n in list(r["histogram"].items())[:14]))
        if args.dump_dir:
            os.makedirs(args.dump_dir, exist_ok=True)
            with open(os.path.join(args.dump_dir, "inner_loop_%s.s" % cfg), "w") as f:
                f.write("; %s\n; inner loop (blocks %s) of %s (hipcc -O3 --offload-arch=gfx950 -S)\n%s\n" % (r["kernel"], lab, tu, text))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
