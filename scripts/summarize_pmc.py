#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, MI355X_MICROARCH.md §HBM) into per-kernel HBM
traffic per launch.  gfx950 correction: FETCH_SIZE reports exactly half of the bytes of a wide coalesced read -> doubled; WRITE_SIZE is
exact for 16-B-per-lane stores.  Units of both counters: KiB.  usage: summarize_pmc.py <fetch.csv> <write.csv> <out.json> [config [note]]
With a config name (C2 / C3 / C4) the summary is merged into <out.json> under that key -- the layout bench.py reads `roofline.traffic` from."""
import json, os, re, sys
import pandas as pd

def per_kernel(path, counter):
    d = pd.read_csv(path); d = d[d["Counter_Name"] == counter]
    d["k"] = d["Kernel_Name"].map(lambda s: (re.search(r"(k_[a-z_]+)", s) or [None, "other"])[1] if "k_" in s else "other")
    g = d.groupby("k")["Counter_Value"].agg(["sum", "count", "mean"])
    return {k: {"launches": int(r["count"]), "mean_KiB": float(r["mean"]), "sum_KiB": float(r["sum"])} for k, r in g.iterrows()}

f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    fetch = f.get(k, {"mean_KiB": 0, "launches": 0}); write = w.get(k, {"mean_KiB": 0, "launches": 0})
    out[k] = {"launches": fetch["launches"] or write["launches"], "fetch_bytes_per_launch_corrected": 2 * fetch["mean_KiB"] * 1024,
              "write_bytes_per_launch": write["mean_KiB"] * 1024,
              "hbm_bytes_per_launch": 2 * fetch["mean_KiB"] * 1024 + write["mean_KiB"] * 1024}
cfg = sys.argv[4] if len(sys.argv) > 4 else None
note = sys.argv[5] if len(sys.argv) > 5 else ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of a reduced-spp `bench.py --steps 1 --warmup 0` run of the "
                                             "same workload (same wavefront batches as the full job, so per-launch figures carry over); FETCH doubled per the gfx950 correction")
if cfg:
    doc = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
    doc[cfg] = {"note": note, "kernels": out}
else:
    doc = {"note": note, "kernels": out}
json.dump(doc, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k:16s} launches {v['launches']:4d}  fetch {v['fetch_bytes_per_launch_corrected']/1e6:9.1f} MB  write {v['write_bytes_per_launch']/1e6:9.1f} MB")
