"""Static VALU issue-cycle estimate of a kernel's hottest loop from hipcc's .s, using the per-instruction issue
costs measured by scripts/ubench_valu*.hip on MI355X (cycles per wave-instruction per SIMD)."""
import re
import sys
from collections import Counter

COST2 = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32",
         "v_add_f32", "v_mul_f32", "v_min_u16", "v_add_u16", "v_not_b32", "v_lshlrev_b32", "v_ashrrev_i32"}
SPECIAL = {"v_bitop3_b32": 2.8, "v_fma_f32": 2.9, "ds_bpermute_b32": 24.0}


def cost(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return 4.2
    if base in SPECIAL:
        return SPECIAL[base]
    if base in COST2:
        return 2.3
    if op.startswith("v_"):
        return 4.2
    return 0.0   # SALU / LDS / VMEM issue on other ports


def hot_loop(s, name):
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    blocks = re.split(r"\n(\.LBB\d+_\d+):", s[i:j])
    best = None
    for k in range(1, len(blocks), 2):
        txt = blocks[k + 1].split("s_cbranch")[0]
        ins = [l.split()[0] for l in txt.split("\n") if l.startswith("\t") and l.split() and not l.strip().startswith((";", "."))]
        n = sum(1 for x in ins if x.startswith(("v_min3", "v_pk_min", "v_min_u16")))
        if best is None or n > best[2]:
            best = (blocks[k], ins, n)
    return best


if __name__ == "__main__":
    s = open(sys.argv[1]).read()
    for name in sys.argv[2:]:
        lab, ins, n = hot_loop(s, name)
        c = Counter(ins)
        total = sum(cost(op) * k for op, k in c.items())
        print("%s %s: %d instrs, %.0f VALU issue cycles/iteration" % (name[:50], lab, len(ins), total))
        print("   " + ", ".join("%d %s" % (k, op) for op, k in c.most_common()))
