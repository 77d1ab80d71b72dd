#!/usr/bin/env python3
"""Benchmark of the page-decode path on MI355X.

Workload (BASELINE.json configs[1]): an INT32 RLE_DICTIONARY column chunk of 256 Mi values per GPU, decoded once
for each bit width in {1,2,4,8,12,16,20} (dictionary min(2^w, 65536) entries, UNCOMPRESSED v1 data pages of
1 Mi values, run mix ~90 % bit-packed / ~10 % RLE).  One step = one decode of the chunk at every width, inputs
already resident in HBM, outputs left in HBM.  value = decoded output bytes / wall time, summed over ranks.

    python bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

# Each handle enqueues on its own HIP stream; the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4) and kernels sharing a queue run one after the other.  With 14 handles in flight 4 queues serialise
# most of the overlap away, so the bench asks for 16 (PQD_HW_QUEUES overrides the figure).
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("PQD_HW_QUEUES", "16")

import torch  # first: its bundled HIP runtime must be the one in the process (see tests/conftest.py)
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import parquet_go_amd as pq  # noqa: E402
from parquet_go_amd import sharding  # noqa: E402
from gen import pqgen  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--values", type=int, default=256 << 20, help="values per GPU in the column chunk")
    ap.add_argument("--page-values", type=int, default=1 << 20)
    ap.add_argument("--widths", type=str, default="1,2,4,8,12,16,20")
    ap.add_argument("--cpu-pages", type=int, default=24, help="pages per width decoded by the CPU oracle (baseline sample)")
    ap.add_argument("--handles", type=int, default=7, help="chunks in flight (one handle / HIP stream each)")
    ap.add_argument("--depth", type=int, default=2, help="steps in flight (sets of handles)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the DELTA_BINARY_PACKED / host-to-host extras")
    ap.add_argument("--delta-values", type=int, default=512 << 20)
    ap.add_argument("--list-slots", type=int, default=128 << 20)
    ap.add_argument("--string-values", type=int, default=100_000_000)
    ap.add_argument("--iso-reps", type=int, default=3, help="serial launches per width for the isolated kernel durations")
    return ap.parse_args()


def to_device(a):
    t = torch.empty(a.size + 64, dtype=torch.uint8, device="cuda")
    t[:a.size].copy_(torch.from_numpy(a))
    t[a.size:].zero_()
    return t


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # one rank per GPU; on a box with fewer GPUs than ranks (a rehearsal of the launch path) ranks share devices and
    # PQD_BENCH_BACKEND=gloo carries the barriers, since RCCL refuses two ranks on one device
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("PQD_BENCH_BACKEND", "nccl")
    red_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    widths = [int(x) for x in args.widths.split(",")]
    n_values, page_values = args.values, args.page_values

    # ---- synthetic inputs, generated on the host and parked in HBM ----
    t_gen = time.time()
    work = []
    for w in widths:
        dict_n = min(1 << w, 65536)
        chunk, pages = pqgen.dict_chunk(0x5EED0002 + 1000 * rank + w, n_values, w, dict_n, page_values)
        work.append({"w": w, "dict_n": dict_n, "pages": pages, "dev": to_device(chunk), "nbytes": chunk.size,
                     "in_bytes": int(pages["length"].sum()), "host": chunk if rank == 0 else None})
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    # One handle (= one HIP stream + workspaces) per chunk in flight, as the reference runs one goroutine per column
    # (reader/reader.go:306-327): the serial run-header walk of one chunk overlaps the expansion of the others.
    n_handles = max(1, min(args.handles, len(widths)))
    decs = [pq.Decoder(local_rank, profiling=True) for _ in range(n_handles)]
    dec = decs[0]
    ktimes = {}

    def submit(item, d):
        d.submit_device(item["dev"].data_ptr(), item["nbytes"], item["pages"], pq.INT32)

    def finish(item, d, collect=False):
        r = d.wait()
        if r.status != 0:
            raise RuntimeError("decode failed: status %d at page %d (w=%d)" % (r.status, r.bad_page, item["w"]))
        if collect:
            for name, (ms, n) in d.kernel_times().items():
                e = ktimes.setdefault((item["w"], name), [0.0, 0])
                e[0] += ms
                e[1] += n
        return r

    def decode(item, collect=False):
        submit(item, dec)
        return finish(item, dec, collect)

    # --depth sets of handles: a step submits its chunks on the next set and only then collects the set that was
    # submitted --depth steps earlier, so the serial run-header walk of one step hides behind the expansion of the
    # previous one (a reader streaming row groups works the same way).  Every step is complete before the timed
    # region ends: drain() waits for the sets still in flight.
    depth = max(1, args.depth)
    sets = [decs] + [[pq.Decoder(local_rank, profiling=True) for _ in range(n_handles)] for _ in range(depth - 1)]
    pending = [[] for _ in range(depth)]
    step_no = [0]

    def collect_set(k, collect):
        for item, d in pending[k]:
            finish(item, d, collect)
        pending[k] = []

    def step(collect=False):
        k = step_no[0] % depth
        step_no[0] += 1
        for g in range(0, len(work), n_handles):
            collect_set(k, collect)
            group = work[g:g + n_handles]
            for item, d in zip(group, sets[k]):
                submit(item, d)
                pending[k].append((item, d))
        if depth == 1:
            collect_set(k, collect)

    def drain(collect=False):
        for k in range(depth):
            collect_set(k, collect)

    # ---- verification against the CPU oracle (untimed) + CPU baseline on a bounded sample ----
    cpu = None
    if rank == 0 and not args.no_cpu:
        from oracle import oracle as orc
        cpu_vals, cpu_s = 0, 0.0
        for item in work:
            k = min(args.cpu_pages, len(item["pages"]) - 1)
            sample_pages = item["pages"][:1 + k]
            t0 = time.time()
            o = orc.read_chunk(item["host"], sample_pages, orc.INT32)
            cpu_s += time.time() - t0
            assert o.status == 0, "oracle failed on the sample"
            cpu_vals += o.n_slots
            decode(item)
            g = dec.fetch()
            gv = g.values.view("<i4").reshape(-1)
            if not np.array_equal(gv[:o.n_slots], o.values.view("<i4").reshape(-1)):
                raise RuntimeError("GPU output differs from the oracle at w=%d" % item["w"])
            del g, gv, o
        cpu = {"value": round(cpu_vals * 4 / cpu_s / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
               "sample": "first %d pages (%d Mi values) of each of the %d widths, decoded by oracle/pq_oracle.c "
                         "(C restatement of the Go decoders; typed vectors instead of boxed interface{}), "
                         "output checked equal to the GPU's" % (args.cpu_pages, args.cpu_pages * page_values >> 20, len(widths))}
    for item in work:
        item["host"] = None

    # warm-up, then exactly --steps timed steps between barrier + device sync on both sides, MAX over ranks
    collect_flag = [False]

    def one_step():
        step(collect=collect_flag[0])

    dist = torch.distributed if world > 1 else None
    sharding.timed_region(one_step, 0, args.warmup, sync=torch.cuda.synchronize, dist=dist, device=red_dev, drain=drain)
    collect_flag[0] = True
    elapsed = sharding.timed_region(one_step, args.steps, 0, sync=torch.cuda.synchronize, dist=dist, device=red_dev,
                                    drain=lambda: drain(True))

    out_bytes_step = len(widths) * n_values * 4
    value = world * out_bytes_step * args.steps / elapsed / 1e9

    # ---- the same chunks once more, one at a time on one stream: per-kernel durations without other chunks' kernels
    # sharing the CUs (the durations taken inside the timed region above include that sharing) ----
    iso = {}
    for item in work:
        for _ in range(args.iso_reps):
            submit(item, dec)
            dec.wait()
            for name, (ms, n) in dec.kernel_times().items():
                e = iso.setdefault((item["w"], name), [0.0, 0])
                e[0] += ms
                e[1] += n

    # ---- roofline of the dominant kernel (bit unpack + dictionary gather), from HIP events around each launch ----
    def expand_stats(kt):
        per_width, tot_ms, tot_bytes, tot_n = {}, 0.0, 0.0, 0
        for item in work:
            w = item["w"]
            ms, n = kt.get((w, "k_hybrid_expand<dict>"), (0.0, 0))
            pre_ms, _ = kt.get((w, "k_hybrid_prescan(dict)"), (0.0, 0))
            alg = item["in_bytes"] + n_values * 4 + item["dict_n"] * 4  # packed input + decoded output + dictionary
            if n:
                per_width[str(w)] = {"expand_ms": round(ms / n, 4), "prescan_ms": round(pre_ms / n, 4),
                                     "expand_GBps": round(alg / (ms / n) / 1e6, 1),
                                     "alg_bytes_per_value": round(alg / n_values, 4)}
                tot_ms += ms
                tot_bytes += alg * n
                tot_n += n
        gbps = tot_bytes / tot_ms / 1e6 if tot_ms else 0.0
        return per_width, gbps, (tot_ms / tot_n if tot_n else None)

    all_ms = {}
    for (w, name), (ms, n) in ktimes.items():
        all_ms[name] = all_ms.get(name, 0.0) + ms
    per_width, achieved, avg_ms = expand_stats(ktimes)
    exp_total_ms = sum(ms for (w, name), (ms, n) in ktimes.items() if name == "k_hybrid_expand<dict>")
    exp_total_bytes = sum((item["in_bytes"] + n_values * 4 + item["dict_n"] * 4) * ktimes.get((item["w"], "k_hybrid_expand<dict>"), (0, 0))[1]
                          for item in work)
    in_flight = exp_total_ms / (elapsed * 1e3) if elapsed else 0.0
    aggregate = exp_total_bytes / elapsed / 1e9 if elapsed else 0.0
    iso_pw, iso_gbps, iso_avg = expand_stats(iso)
    roofline = {"bound": "hbm", "kernel": "k_hybrid_expand<dict> (bit unpack + dictionary gather)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": measured_traffic(args, widths),
                "avg_launch_ms": round(avg_ms, 4) if avg_ms else None, "per_width": per_width,
                "note": "durations from HIP events on each handle's stream inside the timed region, where %d chunks are in "
                        "flight and on average 'mean_in_flight' launches of this kernel are resident at once, each holding a "
                        "share of the CUs for that much longer; 'aggregate' = algorithmic bytes of all its launches / "
                        "elapsed time of the region; 'isolated' is the same launch with the GPU to itself"
                        % (n_handles * depth),
                "mean_in_flight": round(in_flight, 2),
                "aggregate": {"achieved": round(aggregate, 1), "frac": round(aggregate / HBM_PEAK_GBPS, 4)},
                "isolated": {"achieved": round(iso_gbps, 1), "frac": round(iso_gbps / HBM_PEAK_GBPS, 4),
                             "avg_launch_ms": round(iso_avg, 4) if iso_avg else None, "per_width": iso_pw},
                "kernel_ms_share": {k: round(v / max(sum(all_ms.values()), 1e-9), 4) for k, v in all_ms.items()}}

    extra = {}
    if rank == 0 and not args.no_extra:
        extra = run_extras(args, dec, local_rank)

    if rank == 0:
        res = {
            "metric": "device-resident decoded GB/s, INT32 RLE_DICTIONARY (bit widths 1-20)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "configs[1]: INT32 RLE_DICTIONARY column chunk, %d Mi values per GPU, one decode per bit "
                                   "width in %s, dictionary min(2^w,65536) entries, UNCOMPRESSED v1 pages of %d Ki values, "
                                   "~90%% bit-packed / ~10%% RLE runs; input and output resident in HBM"
                                   % (n_values >> 20, widths, page_values >> 10),
                       "values_per_gpu": n_values, "page_values": page_values, "widths": widths,
                       "chunks_in_flight": n_handles, "steps_in_flight": depth,
                       "hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),
                       "gen_seconds": round(t_gen, 1)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


def measured_traffic(args, widths):
    """HBM bytes per launch of the expand kernel from the memory-side L2 counters.  The counters need rocprofv3, which
    cannot run inside this process, so the figure comes from the committed summary of profiles/pmc_expand.sh for this
    very workload shape; None when no summary matches."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if t.get("values_per_gpu") != args.values or t.get("page_values") != args.page_values or t.get("widths") != widths:
        return None
    return t.get("traffic")


def run_extras(args, dec, local_rank):
    """Secondary numbers kept next to the headline: DELTA_BINARY_PACKED INT64 (configs[2]) device-resident, and the
    host-to-host rate (pinned staging + H2D + decode + D2H) of one dictionary chunk."""
    extra = {}
    try:
        n = args.delta_values
        chunk, pages = pqgen.delta_chunk(0x5EED0003, n, 12, 128 << 10)
        dev = to_device(chunk)
        torch.cuda.synchronize()
        best = None
        kt = {}
        for i in range(4):
            t0 = time.perf_counter()
            dec.submit_device(dev.data_ptr(), chunk.size, pages, pq.INT64)
            r = dec.wait()
            dt = time.perf_counter() - t0
            assert r.status == 0, r.status
            if i:
                best = dt if best is None else min(best, dt)
                kt = dec.kernel_times()
        in_bytes = int(pages["length"].sum())
        kms = kt.get("k_delta_decode", (0.0, 1))[0]
        extra["delta_int64"] = {"workload": "configs[2]: INT64 DELTA_BINARY_PACKED, %d Mi values, 128-value blocks / 4 miniblocks, "
                                            "delta width 12, pages of 128 Ki values" % (n >> 20),
                                "decoded_GBps": round(n * 8 / best / 1e9, 1), "ms": round(best * 1e3, 3),
                                "kernel_ms": round(kms, 3),
                                "kernel_alg_GBps": round((in_bytes + n * 8) / kms / 1e6, 1) if kms else None,
                                "roofline_frac": round((in_bytes + n * 8) / kms / 1e6 / HBM_PEAK_GBPS, 4) if kms else None}
        del dev
    except Exception as e:  # the extras never take the headline down
        extra["delta_int64"] = {"error": repr(e)}
    def timed_device(chunk, pages, ptype, max_def=0, max_rep=0, reps=3):
        dev = to_device(chunk)
        torch.cuda.synchronize()
        best, kt = None, {}
        for i in range(reps + 1):
            t0 = time.perf_counter()
            dec.submit_device(dev.data_ptr(), chunk.size, pages, ptype, 0, max_def, max_rep)
            r = dec.wait()
            dt = time.perf_counter() - t0
            assert r.status == 0, r.status
            if i and (best is None or dt < best):
                best, kt = dt, dec.kernel_times()
        return best, kt, r

    # configs[0]-shaped: what the reference's own writer produces (layout/dictpage.go:155-158, encodingwrite.go:183-214):
    # dictionary indices at width 32 as RLE runs only, one run per change of value, pages of about 8 KiB
    try:
        n = 64 << 20
        chunk, pages = pqgen.dict_chunk(0x5EED0007, n, 32, 4096, 2048, 1, 2)
        best, kt, r = timed_device(chunk, pages, pq.INT32)
        extra["parquet_go_written"] = {"workload": "INT32 dictionary chunk as parquet-go writes it: width-32 RLE runs only (1-2 values per "
                                                   "run), %d Mi values in %d pages of 2048 values" % (n >> 20, len(pages) - 1),
                                       "decoded_GBps": round(n * 4 / best / 1e9, 1), "ms": round(best * 1e3, 3),
                                       "in_bytes": int(pages["length"].sum()),
                                       "kernel_ms": {k_: round(v[0], 3) for k_, v in kt.items()}}
    except Exception as e:
        extra["parquet_go_written"] = {"error": repr(e)}

    # configs[4]: nested levels (optional list of optional int64), PLAIN values
    try:
        n = args.list_slots
        chunk, pages = pqgen.list_chunk(0x5EED0005, n, 128 << 10)
        in_bytes = int(pages["length"].sum())
        best, kt, r = timed_device(chunk, pages, pq.INT64, 3, 1)
        out_bytes = n * (4 + 4 + 8)
        lv_ms = kt.get("k_hybrid_expand<levels>", (0.0, 1))[0]
        extra["list_int64"] = {"workload": "configs[4]: optional list<optional int64>, %d Mi level slots (max_def 3, max_rep 1), pages of "
                                           "128 Ki slots, PLAIN values; output = rep + def (int32 each) + slot-aligned int64 values"
                                           % (n >> 20),
                               "decoded_GBps": round(out_bytes / best / 1e9, 1), "ms": round(best * 1e3, 3),
                               "rows": int(r.num_rows), "nulls": int(r.num_nulls),
                               "kernel_ms": {k_: round(v[0], 3) for k_, v in kt.items()},
                               "levels_expand_alg_GBps": round((n * 3 / 8 + n * 8) / lv_ms / 1e6, 1) if lv_ms else None}
    except Exception as e:
        extra["list_int64"] = {"error": repr(e)}

    # configs[3]: key-like strings, DELTA_LENGTH_BYTE_ARRAY and DELTA_BYTE_ARRAY
    for mode, name in ((0, "delta_length_byte_array"), (1, "delta_byte_array")):
        try:
            n = args.string_values
            chunk, pages = pqgen.strings_chunk(0x5EED0004, n, 64 << 10, mode)
            in_bytes = int(pages["length"].sum())
            dec._is_bytes = True
            best, kt, r = timed_device(chunk, pages, pq.BYTE_ARRAY)
            out_bytes = int(r.bytes_len) + 8 * (n + 1)
            extra[name] = {"workload": "configs[3]: %d M strings of 8-64 bytes sharing prefixes with their predecessor, pages of 64 Ki "
                                       "strings; output = int64 offsets + bytes" % (n // 1000000),
                           "decoded_GBps": round(out_bytes / best / 1e9, 1), "ms": round(best * 1e3, 3),
                           "in_bytes": in_bytes, "out_bytes": out_bytes,
                           "alg_GBps": round((in_bytes + out_bytes) / best / 1e9, 1),
                           "roofline_frac": round((in_bytes + out_bytes) / best / 1e9 / HBM_PEAK_GBPS, 4),
                           "kernel_ms": {k_: round(v[0], 3) for k_, v in kt.items()}}
        except Exception as e:
            extra[name] = {"error": repr(e)}
    dec._is_bytes = False

    try:
        n = 64 << 20
        chunk, pages = pqgen.dict_chunk(0x5EED0009, n, 12, 4096, 1 << 20)
        best = None
        for i in range(3):
            t0 = time.perf_counter()
            dec.submit(chunk, pages, pq.INT32)
            t = dec.fetch()
            dt = time.perf_counter() - t0
            assert t.status == 0
            best = dt if best is None else min(best, dt)
        extra["host_to_host"] = {"workload": "INT32 RLE_DICTIONARY w=12, %d Mi values: host bytes -> pinned staging -> H2D -> decode "
                                             "-> D2H into freshly allocated pageable numpy arrays (values and both level arrays)" % (n >> 20),
                                 "decoded_GBps": round(n * 4 / best / 1e9, 2), "ms": round(best * 1e3, 2)}
        best = None
        for i in range(3):
            t0 = time.perf_counter()
            dec.submit(chunk, pages, pq.INT32)
            t = dec.fetch_pinned()
            dt = time.perf_counter() - t0
            assert t.status == 0
            best = dt if best is None else min(best, dt)
        best_in = None
        for i in range(3):
            t0 = time.perf_counter()
            buf = dec.input_buffer(chunk.size)
            buf[:] = chunk          # stands for the file read: the caller's bytes land in page-locked memory directly
            t1 = time.perf_counter()
            dec.submit(buf, pages, pq.INT32)
            t = dec.fetch_pinned()
            dt = time.perf_counter() - t1
            assert t.status == 0
            best_in = dt if best_in is None else min(best_in, dt)
        extra["pinned_to_pinned"] = {"workload": "same chunk already read into the handle's page-locked input buffer "
                                                 "(pqd_input_buffer), result read in place (pqd_result_host)",
                                     "decoded_GBps": round(n * 4 / best_in / 1e9, 2), "ms": round(best_in * 1e3, 2)}
        extra["host_to_pinned"] = {"workload": "same chunk, result left in the handle's page-locked memory (pqd_result_host) "
                                               "for the caller to read in place",
                                   "decoded_GBps": round(n * 4 / best / 1e9, 2), "ms": round(best * 1e3, 2)}
    except Exception as e:
        extra["host_to_host"] = {"error": repr(e)}
    return extra


if __name__ == "__main__":
    main()
