#!/bin/bash
# Runs on the GPU box from the repo root:  bash profiles/collect.sh [tag]
# 1. rocprofv3 --kernel-trace --stats of the default bench.py run   -> gpurun_out/prof_<tag>/kt
# 2. two counter passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) over a short bench.py run -> .../fetch, .../write
# profiles/summarize.py turns the CSVs into the small files kept under profiles/.
set -u
TAG=${1:-r1}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err" || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.err"; exit 1; }
echo "kernel-trace pass done"
SHORT="--no-cpu --no-extra --steps 1 --warmup 0 --iso-reps 1 --handles 1 --depth 1"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/bench.py" $SHORT > "$OUT/fetch.json" 2> "$OUT/fetch.err" || { echo "FETCH_SIZE pass failed"; tail -5 "$OUT/fetch.err"; exit 1; }
echo "FETCH_SIZE pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/bench.py" $SHORT > "$OUT/write.json" 2> "$OUT/write.err" || { echo "WRITE_SIZE pass failed"; tail -5 "$OUT/write.err"; exit 1; }
echo "WRITE_SIZE pass done"
cd "$R" && python3 profiles/summarize.py "$OUT" "$TAG"
# the box's profiles/ does not travel back; gpurun_out/ does
mkdir -p "$R/gpurun_out/profiles_$TAG" && cp "$R"/profiles/${TAG}_*.csv "$R"/profiles/traffic.json "$R/gpurun_out/profiles_$TAG/"
