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
    # ---- serialized launches ----
    for f in find(os.path.join(out, "kt_serial"), "kernel_stats.csv"):
        rows = list(csv.reader(open(f)))
        with open(os.path.join(here, tag + "_kernel_stats_serial.csv"), "w", newline="") as g:
            csv.writer(g).writerows(rows)
        break
    # ---- counters of the other encodings: HBM bytes per decode call and per kernel ----
    extras = {}
    with open(os.path.join(here, tag + "_pmc_extras.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["workload", "kernel", "launches_per_call", "mean_FETCH_SIZE_KB", "mean_WRITE_SIZE_KB", "hbm_bytes_per_call(2*F+W)"])
        for d in sorted(glob.glob(os.path.join(out, "ex_*"))):
            if not os.path.isdir(d):
                continue
            name = os.path.basename(d)[3:]
            per = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
            for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
                for f in find(os.path.join(d, sub), "counter_collection.csv"):
                    for r in csv.DictReader(open(f)):
                        if r["Counter_Name"] == ctr:
                            e = per[short(r["Kernel_Name"])][ctr]
                            e[0] += 1
                            e[1] += float(r["Counter_Value"])
            calls = max(1, max((v["FETCH_SIZE"][0] for k, v in per.items() if "k_collect_status" in k), default=1))
            total = 0
            for k, v in sorted(per.items()):
                f_n, f_v = v.get("FETCH_SIZE", [0, 0.0])
                w_n, w_v = v.get("WRITE_SIZE", [0, 0.0])
                b = int((2 * f_v + w_v) * 1024 / calls)
                total += b
                w.writerow([name, k, round(max(f_n, w_n) / calls, 2), round(f_v / max(f_n, 1), 1), round(w_v / max(w_n, 1), 1), b])
            try:
                res = json.loads(open(os.path.join(out, "ex_%s.fetch.json" % name)).read().strip().splitlines()[-1])["extra"][name]
            except Exception:
                res = {}
            extras[name] = {"traffic": total, "calls": calls, "in_bytes": res.get("in_bytes"), "out_bytes": res.get("out_bytes"),
                            "workload": res.get("workload")}
    tj = os.path.join(here, "traffic.json")
    try:
        t = json.load(open(tj))
    except Exception:
        t = {}
    t["extras"] = extras
    t["extras_unit"] = "HBM bytes per decode call (2*FETCH_SIZE + WRITE_SIZE over all kernels of the call), " + tag + "_pmc_extras.csv"
    json.dump(t, open(tj, "w"), indent=1)
    # ---- the headline at 4 hardware queues ----
    try:
        r4 = json.loads(open(os.path.join(out, "bench_4_queues.json")).read().strip().splitlines()[-1])
        r16 = json.loads(open(os.path.join(out, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
        json.dump({"hw_queues_4": {"value": r4["value"], "ms_per_step": r4["ms_per_step"], "frac": r4["roofline"]["frac"]},
                   "hw_queues_16_under_rocprof": {"value": r16["value"], "ms_per_step": r16["ms_per_step"], "frac": r16["roofline"]["frac"]}},
                  open(os.path.join(here, tag + "_queues.json"), "w"), indent=1)
    except Exception as ex:
        print("no queue comparison:", ex)
    print("summaries written to", here)


if __name__ == "__main__":
    main()
