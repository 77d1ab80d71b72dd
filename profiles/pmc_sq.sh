#!/bin/bash
# bash profiles/pmc_sq.sh <widths> <tag> [extra]: SQ issue / wait counters of the hybrid and delta kernels for the given bit widths (and one of the extras)
set -u
W=${1:-8}; TAG=${2:-sq}
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu --steps 1 --warmup 0 --iso-reps 1 --handles 1 --widths $W --only-extra ${3:-none}"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/a" -- python3 "$R/bench.py" $ARGS > "$OUT/a.json" 2> "$OUT/a.err" || { echo "pass a failed"; tail -5 "$OUT/a.err"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/b" -- python3 "$R/bench.py" $ARGS > "$OUT/b.json" 2> "$OUT/b.err" || { echo "pass b failed"; tail -5 "$OUT/b.err"; exit 1; }
cd "$R" && python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "expand" not in k and "prescan" not in k and "delta" not in k and "fused" not in k:
            continue
        e = agg[k][r["Counter_Name"]]
        e[0] += 1; e[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print("   %-24s %14.0f  (mean of %d)" % (c, v / n, n))
PY
