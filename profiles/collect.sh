#!/bin/bash
# Runs on the GPU box from the repo root:  bash profiles/collect.sh [tag]
# 1. rocprofv3 --kernel-trace --stats of the default bench.py run   -> gpurun_out/prof_<tag>/kt
# 2. two counter passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) over a short bench.py run -> .../fetch, .../write
# 3. the kernel trace once more with launches serialized; 4. counter passes over the other encodings; 5. the headline at 4 hardware queues
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
# 3. the same bench with one handle and one step in flight: launches one after the other, so that average duration x launches
#    ties to the per-kernel durations bench.py reports as 'isolated'
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_serial" -- python3 "$R/bench.py" --no-cpu --handles 1 --depth 1 --steps 5 --warmup 1 > "$OUT/bench_serial_under_rocprof.json" 2> "$OUT/kt_serial.err" || { echo "serial kernel-trace pass failed"; tail -5 "$OUT/kt_serial.err"; exit 1; }
echo "serial kernel-trace pass done"
# 4. counter passes over the other encodings, one run per workload and counter
for X in delta_int64 delta_int32 delta_length_byte_array delta_byte_array list_int64; do
  EX="--no-cpu --steps 1 --warmup 0 --iso-reps 1 --handles 1 --depth 1 --widths none --only-extra $X"
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/ex_$X/fetch" -- python3 "$R/bench.py" $EX > "$OUT/ex_$X.fetch.json" 2> "$OUT/ex_$X.fetch.err" || { echo "FETCH_SIZE pass of $X failed"; tail -5 "$OUT/ex_$X.fetch.err"; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/ex_$X/write" -- python3 "$R/bench.py" $EX > "$OUT/ex_$X.write.json" 2> "$OUT/ex_$X.write.err" || { echo "WRITE_SIZE pass of $X failed"; tail -5 "$OUT/ex_$X.write.err"; exit 1; }
  echo "counter passes of $X done"
done
# 5. the headline at the HIP default of 4 hardware queues per process, next to the 16 the bench asks for
cd "$R" && PQD_HW_QUEUES=4 timeout -k 10 300 python3 bench.py --no-cpu --no-extra > "$OUT/bench_4_queues.json" 2> "$OUT/bench_4_queues.err" || { echo "4-queue run failed"; exit 1; }
cd "$R" && python3 profiles/summarize.py "$OUT" "$TAG"
# the box's profiles/ does not travel back; gpurun_out/ does
mkdir -p "$R/gpurun_out/profiles_$TAG" && cp "$R"/profiles/${TAG}_* "$R"/profiles/traffic.json "$R/gpurun_out/profiles_$TAG/"
