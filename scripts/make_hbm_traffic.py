"""profiles/r01/traffic_raw.json (per-kernel FETCH_SIZE / WRITE_SIZE means from scripts/gpu_final_r01.sh) ->
profiles/hbm_traffic.json: bytes per bench step and config, as bench.py's roofline.traffic reads them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r01", "traffic_raw.json")))
old = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
ALG = {"cfg1": 20520000, "cfg2": 516000000, "cfg3": 819600000, "cfg4": 260000000, "cfg5": 1073741824}
out = {"_note": old["_note"]}
for wl, per in raw.items():
    # every kernel of the path runs once per bench step, except helpers launched per round (counted by their share)
    steps = max(per["FETCH_SIZE"].values(), key=lambda v: v[0])[1]   # launches of the heaviest kernel = bench steps incl. warm-up
    fetch = {k: v[0] * v[1] / steps for k, v in per["FETCH_SIZE"].items()}
    write = {k: v[0] * v[1] / steps for k, v in per["WRITE_SIZE"].items()}
    f, w = sum(fetch.values()), sum(write.values())
    out[wl] = {"bytes_per_launch": int(f * 1024 * 2 + w * 1024), "fetch_size_kib_raw": int(f), "write_size_kib": int(w),
               "algorithmic_bytes": ALG[wl], "kernels": {k: int(v) for k, v in fetch.items() if v >= 1}}
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
for wl in sorted(k for k in out if k != "_note"):
    print(wl, out[wl]["bytes_per_launch"], "x%.2f" % (out[wl]["bytes_per_launch"] / out[wl]["algorithmic_bytes"]))
