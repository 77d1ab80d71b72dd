#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (see collect.sh) into the files kept in this directory:

  <tag>_kernel_stats.csv   per-kernel totals / averages of the --kernel-trace --stats run of bench.py
  <tag>_expand_by_geometry.csv   average duration of the one-pass kernel (and of the tile expansion) per launch geometry
  <tag>_pmc.csv            per kernel: launches, mean FETCH_SIZE and WRITE_SIZE (KB, as rocprofv3 reports them)
  traffic.json             HBM bytes per launch of the one-pass kernel (k_page_fused) = (2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950
                           correction for wide coalesced reads applied (FETCH_SIZE counts 128-byte requests as 64)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    return hits


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def main():
    out, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    # ---- kernel stats ----
    for f in find(os.path.join(out, "kt"), "kernel_stats.csv"):
        rows = list(csv.reader(open(f)))
        with open(os.path.join(here, tag + "_kernel_stats.csv"), "w", newline="") as g:
            csv.writer(g).writerows(rows)
        break
    agg = defaultdict(lambda: [0, 0.0])
    for f in find(os.path.join(out, "kt"), "kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "k_page_fused" not in r["Kernel_Name"] and "k_hybrid_expand" not in r["Kernel_Name"]:
                continue
            key = (short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""))
            e = agg[key]
            e[0] += 1
            e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open(os.path.join(here, tag + "_expand_by_geometry.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["kernel", "grid", "lds_bytes", "launches", "avg_us"])
        for k, (n, us) in sorted(agg.items()):
            w.writerow(list(k) + [n, round(us / n, 2)])
    # ---- counters ----
    pmc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for f in find(os.path.join(out, sub), "counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != ctr:
                    continue
                e = pmc[short(r["Kernel_Name"])][ctr]
                e[0] += 1
                e[1] += float(r["Counter_Value"])
    with open(os.path.join(here, tag + "_pmc.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["kernel", "launches", "mean_FETCH_SIZE_KB", "mean_WRITE_SIZE_KB", "hbm_bytes_per_launch(2*F+W)"])
        for k, d in sorted(pmc.items()):
            f_n, f_v = d.get("FETCH_SIZE", [0, 0.0])
            w_n, w_v = d.get("WRITE_SIZE", [0, 0.0])
            fm = f_v / f_n if f_n else 0.0
            wm = w_v / w_n if w_n else 0.0
            w.writerow([k, max(f_n, w_n), round(fm, 1), round(wm, 1), int((2 * fm + wm) * 1024)])
    # the 4-byte dictionary gather instantiations (dictionary in LDS or in global memory) taken together
    tot = {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]}
    for k, d in pmc.items():
        if "k_page_fused<unsigned int" in k:
            for c in tot:
                tot[c][0] += d.get(c, [0, 0.0])[0]
                tot[c][1] += d.get(c, [0, 0.0])[1]
    try:
        cfg = json.loads(open(os.path.join(out, "fetch.json")).read().strip().splitlines()[-1])["config"]
    except Exception:
        cfg = {}
    if tot["FETCH_SIZE"][0] and tot["WRITE_SIZE"][0]:
        fm = tot["FETCH_SIZE"][1] / tot["FETCH_SIZE"][0]
        wm = tot["WRITE_SIZE"][1] / tot["WRITE_SIZE"][0]
        json.dump({"traffic": int((2 * fm + wm) * 1024), "unit": "bytes per launch of k_page_fused, mean over the widths",
                   "fetch_size_kb": round(fm, 1), "write_size_kb": round(wm, 1), "launches": tot["FETCH_SIZE"][0],
                   "values_per_gpu": cfg.get("values_per_gpu"), "page_values": cfg.get("page_values"), "widths": cfg.get("widths"),
                   "source": tag + "_pmc.csv"}, open(os.path.join(here, "traffic.json"), "w"), indent=1)
    print("summaries written to", here)


if __name__ == "__main__":
    main()
