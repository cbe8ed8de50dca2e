"""not-gpu: invariants of the band kernel's launch planner (lev_plan.h, shared by the C ABI and the emulation)."""
import numpy as np

import emu_lib as E
import oracle_lib as O


def plan_for(k, max_len, costs=(1, 1, 0, None), force_D=0, force_L=0):
    a = [b"x" * max_len]
    b = [b"y" * max_len]
    try:
        _, pl = E.lev_band(a, b, k, costs, force_D=force_D, force_L=force_L)
    except RuntimeError:
        return None
    return pl


def batch_unit_k(k, costs, max_len):
    """lev_plan.h lev_batch_unit_k: the dispatcher's max_k clamp (src/levenshtein.rs:734-757) taken over every pair
    with both lengths <= max_len, then unit_k = (K - sg) / gc, capped by the 2 max_len diagonals a matrix has."""
    mc, gc, sg, _ = costs
    K = min(k, 2 * max_len * gc + 2 * sg, max_len * max(mc, gc) + sg)
    return min(max(0, K - sg) // gc, 2 * max_len)


def test_plan_invariants():
    for max_len in (1, 5, 64, 256, 1000):
        for k in (0, 1, 8, 31, 32, 33, 100, 254, 255, 1000, 0xFFFFFFFF):
            for costs in [(1, 1, 0, None), (2, 3, 1, None), (1, 255, 0, None), (5, 1, 4, 1)]:
                pl = plan_for(k, max_len, costs)
                u = batch_unit_k(k, costs, max_len)
                if pl is None:
                    assert u + 2 > 64 * 66
                    continue
                assert pl["u"] == u and pl["o"] == ((u >> 1) | 1)
                # every pair's band fits: the widest one holds u + 1 diagonals plus one slot of parity padding
                assert pl["D"] % 2 == 0 and pl["D"] * pl["L"] >= u + 2
                assert 1 <= pl["L"] <= 64 and pl["PW"] == 64 // pl["L"] and pl["PW"] * pl["L"] <= 64
                # the batch bound never falls below a pair's own unit_k (ta_levenshtein_select) for |delta| <= u
                for la, lb in ((max_len, max_len), (max_len, max_len // 2), (1, max_len)):
                    assert min(O.levenshtein_select(la, lb, k, costs)[1], 2 * max_len) <= max(u, 0) or abs(la - lb) > u


def test_plan_band_is_half_of_the_reference_band_on_baseline_configs():
    # cfg2: unit_k = 32: reference band 65 diagonals, ours 33 (+1 parity slot); cfg4: unit_k = 8: 17 vs 9 (+1)
    assert O.levenshtein_select(256, 256, 32)[1] == 32
    pl = plan_for(32, 256)
    assert pl["u"] == 32 and pl["o"] == 17 and pl["D"] * pl["L"] >= 34 and pl["D"] * pl["L"] < 66
    pl = plan_for(8, 128, (1, 1, 0, 1))
    assert pl["u"] == 8 and pl["o"] == 5 and pl["D"] * pl["L"] >= 10 and pl["D"] * pl["L"] < 18


def test_sliced_plan_geometry():
    """lev_sliced_make_plan: the strips cover the band of 3.1, the answer cell is row a_len of column b_len, S is odd."""
    seen_ok = 0
    for a_len in (1, 40, 64, 127, 256, 300, 512, 600):
        for d in (-40, -10, -3, 0, 4, 11, 45):
            b_len = a_len + d
            if b_len < 1:
                continue
            for uk in (0, 8, 23, 24, 25, 32, 33, 40, 41, 44, 45, 60):
                p = E.sliced_plan(a_len, b_len, uk)
                if abs(d) > uk or b_len > 512:
                    assert not p["ok"]
                    continue
                t = (uk - abs(d)) // 2
                W = abs(d) + 2 * t + 1
                assert p["S"] % 2 == 1 and 3 * p["S"] >= W and 3 * (p["S"] - 2) < W
                assert p["ok"] == (9 <= p["S"] <= 15)
                assert p["dhi"] == max(0, d) + t and p["dabs"] == abs(d)
                w_ans = 3 * p["c_ans"] + p["e_ans"]
                assert 0 <= w_ans < W and p["e_ans"] < 3
                assert b_len - p["dhi"] + w_ans == a_len                  # row of window cell w_ans at column b_len
                assert p["steps"] % 64 == 0 and p["steps"] >= 2 * b_len + p["S"]
                seen_ok += p["ok"]
    assert seen_ok > 50
    assert E.sliced_plan(256, 256, 32) == dict(ok=1, S=11, dhi=16, c_ans=5, e_ans=1, dabs=0, steps=576)
