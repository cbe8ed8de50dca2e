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


def test_plan_invariants():
    for max_len in (1, 5, 64, 256, 1000):
        for k in (0, 1, 8, 31, 32, 33, 100, 254, 255, 1000, 0xFFFFFFFF):
            for costs in [(1, 1, 0, None), (2, 3, 1, None), (1, 255, 0, None), (5, 1, 4, 1)]:
                pl = plan_for(k, max_len, costs)
                mc, gc, sg, tc = costs
                u = min(max(0, min(k, 0xFFFFFFFF) - sg) // gc, max_len)
                if pl is None:
                    assert (u | 1) + u + 1 > 64 * 66
                    continue
                assert pl["u"] == u and pl["o"] == (u | 1)
                need = pl["o"] + pl["u"] + 1
                assert pl["D"] % 2 == 0 and pl["D"] * pl["L"] >= need            # every band diagonal has a register
                assert 1 <= pl["L"] <= 64 and pl["PW"] == 64 // pl["L"] and pl["PW"] * pl["L"] <= 64


def test_plan_matches_reference_band_on_baseline_configs():
    # cfg2: unit_k = 32 -> 66 diagonals; cfg4: unit_k = 8 -> 18 diagonals
    assert O.levenshtein_select(256, 256, 32)[1] == 32
    pl = plan_for(32, 256)
    assert pl["u"] == 32 and pl["o"] == 33 and pl["D"] * pl["L"] >= 66
    pl = plan_for(8, 128, (1, 1, 0, 1))
    assert pl["u"] == 8 and pl["o"] == 9 and pl["D"] * pl["L"] >= 18
