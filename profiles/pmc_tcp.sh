#!/bin/bash
# bash profiles/pmc_tcp.sh <width> <tag>: vector L1 (TCP) and address unit (TA) counters of the one-pass kernel for one bit width
W=${1:-16}; TAG=${2:-tcp}
R=$(pwd); OUT=$R/gpurun_out/$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu --steps 1 --warmup 0 --iso-reps 1 --handles 1 --widths $W --only-extra none"
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/a" -- python3 "$R/bench.py" $ARGS > "$OUT/a.json" 2> "$OUT/a.err" || { echo "pass a failed"; tail -5 "$OUT/a.err"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d "$OUT/b" -- python3 "$R/bench.py" $ARGS > "$OUT/b.json" 2> "$OUT/b.err" || { echo "pass b failed"; tail -5 "$OUT/b.err"; exit 1; }
cd "$R" && python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "fused" not in k: continue
        e = agg[k][r["Counter_Name"]]
        e[0] += 1; e[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print("   %-44s %16.0f  (mean of %d)" % (c, v / n, n))
PY
