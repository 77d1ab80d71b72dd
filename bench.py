#!/usr/bin/env python3
"""Benchmark of the page-decode path on MI355X.

Workload (BASELINE.json configs[1]): an INT32 RLE_DICTIONARY column chunk of 256 Mi values per GPU, decoded once
for each bit width in {1,2,4,8,12,16,20} (dictionary min(2^w, 65536) entries, UNCOMPRESSED v1 data pages of
1 Mi values, run mix ~90 % bit-packed / ~10 % RLE).  One step = one decode of the chunk at every width, inputs
already resident in HBM, outputs left in HBM.  value = decoded output bytes / wall time, summed over ranks.

    python bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no rank environment the script starts N rank processes itself (torch.distributed.run, one per
GPU) before anything touches the GPU, and exits with their status.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--values", type=int, default=256 << 20, help="values per GPU in the column chunk")
    ap.add_argument("--page-values", type=int, default=1 << 20)
    ap.add_argument("--widths", type=str, default="1,2,4,8,12,16,20")
    ap.add_argument("--cpu-pages", type=int, default=64, help="pages per width decoded by the CPU oracle (baseline sample)")
    ap.add_argument("--handles", type=int, default=7, help="chunks in flight (one handle / HIP stream each)")
    ap.add_argument("--depth", type=int, default=2, help="steps in flight (sets of handles)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other encodings, the scaling legs and the host-to-host extras")
    ap.add_argument("--delta-values", type=int, default=512 << 20)
    ap.add_argument("--list-slots", type=int, default=128 << 20)
    ap.add_argument("--string-values", type=int, default=100_000_000)
    ap.add_argument("--row-groups", type=int, default=1024, help="configs[4]: row groups of 1 Mi level slots over all GPUs")
    ap.add_argument("--rg-group", type=int, default=64, help="configs[4] row-group leg: chunks per decode call")
    ap.add_argument("--only-extra", type=str, default="", help="comma separated names: run only these extras (development aid)")
    ap.add_argument("--iso-reps", type=int, default=3, help="serial launches per width for the isolated kernel durations")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N without a rank environment: N fresh rank processes, one per GPU, started before any GPU call of this
    process (a process that has touched the GPU must not be replaced or forked)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    import torch
    if "PQD_BENCH_BACKEND" not in env and torch.cuda.device_count() < args.gpus:
        env["PQD_BENCH_BACKEND"] = "gloo"   # ranks share devices (a rehearsal of the launch path): RCCL refuses that
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))

    # Each handle enqueues on its own HIP stream; the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues
    # (default 4) and kernels sharing a queue run one after the other.  With 14 handles in flight 4 queues serialise
    # most of the overlap away, so the bench asks for 16 (PQD_HW_QUEUES overrides the figure).
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("PQD_HW_QUEUES", "16")

    import torch  # first: its bundled HIP runtime must be the one in the process (see tests/conftest.py)
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parquet_go_amd as pq
    from parquet_go_amd import sharding
    from gen import pqgen

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # one rank per GPU; on a box with fewer GPUs than ranks (a rehearsal of the launch path) ranks share devices and
    # PQD_BENCH_BACKEND=gloo carries the barriers, since RCCL refuses two ranks on one device
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("PQD_BENCH_BACKEND", "nccl")
    red_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    widths = [int(x) for x in args.widths.split(",") if x.strip().isdigit()]     # ("none": the other encodings only, a development aid)
    n_values, page_values = args.values, args.page_values
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))

    def to_device(a):
        t = torch.empty(a.size + 64, dtype=torch.uint8, device="cuda")
        t[:a.size].copy_(torch.from_numpy(a))
        t[a.size:].zero_()
        return t

    # ---- CPU baseline helpers: the oracle on a bounded sample of pages, one core and `cores` threads ----
    orc = None
    if rank == 0 and not args.no_cpu:
        from oracle import oracle as orc

    def cpu_sample(chunk, pages, lead, first, count, ptype, md=0, mr=0, out_bytes=None):
        """Decodes pages [first, first + count) (behind the `lead` dictionary pages) with the oracle: once on one
        core (-> table, seconds), once split over `cores` threads (-> seconds).  The C library runs without the GIL."""
        from concurrent.futures import ThreadPoolExecutor
        sel = np.concatenate([pages[:lead], pages[first:first + count]])
        t0 = time.perf_counter()
        tab = orc.read_chunk(chunk, sel, ptype, 0, md, mr)
        t1 = time.perf_counter() - t0
        assert tab.status == 0, "oracle failed on the sample"
        parts = [(first + count * k // cores, first + count * (k + 1) // cores) for k in range(cores)]
        parts = [(a, b) for a, b in parts if b > a]

        def work(ab):
            rc, n = orc.read_chunk_timed(chunk, np.concatenate([pages[:lead], pages[ab[0]:ab[1]]]), ptype, 0, md, mr)
            assert rc == 0
            return n
        t0 = time.perf_counter()
        with ThreadPoolExecutor(len(parts)) as ex:
            list(ex.map(work, parts))
        tn = time.perf_counter() - t0
        return tab, t1, tn

    def cpu_entry(nbytes, t1, tn, what):
        return {"GBps_1_core": round(nbytes / t1 / 1e9, 4), "GBps_threads": round(nbytes / tn / 1e9, 4), "threads": cores,
                "kind": "port", "sample": what}

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

    # --depth sets of handles: a step takes the next set; each of its handles first collects the chunk it was given
    # --depth steps earlier, so the serial run-header walk of one step hides behind the expansion of the earlier
    # ones (a reader streaming row groups works the same way).  Every step is complete before the timed region
    # ends: drain() waits for the sets still in flight.
    depth = max(1, args.depth)
    sets = [decs] + [[pq.Decoder(local_rank, profiling=True) for _ in range(n_handles)] for _ in range(depth - 1)]
    pending = [[] for _ in range(depth)]
    step_no = [0]

    def collect_set(k, collect):
        for item, d in pending[k]:
            finish(item, d, collect)
        pending[k] = []

    def step(collect=False):
        # Every handle works like one column's goroutine in the reference (reader/reader.go:306-327): it waits for its own
        # previous chunk and takes the next one at once, so its run-header walk starts while the other handles' chunks of
        # the earlier steps are still being expanded (collecting the whole set first leaves the GPU with nothing but
        # serial walks to do until the first of them ends).
        k = step_no[0] % depth
        step_no[0] += 1
        for g in range(0, len(work), n_handles):
            group = work[g:g + n_handles]
            prev = {id(d): item for item, d in pending[k]}
            pending[k] = []
            for item, d in zip(group, sets[k]):
                if id(d) in prev:
                    finish(prev.pop(id(d)), d, collect)
                submit(item, d)
                pending[k].append((item, d))
            for d in sets[k]:   # handles the group did not use (a last group smaller than the set)
                if id(d) in prev:
                    finish(prev.pop(id(d)), d, collect)
        if depth == 1:
            collect_set(k, collect)

    def drain(collect=False):
        for j in range(depth):
            collect_set((step_no[0] + j) % depth, collect)

    # ---- verification against the CPU oracle (untimed) + CPU baseline on a bounded sample ----
    cpu = None
    if orc is not None and work:      # (no widths asked for: nothing to sample)
        cpu_vals, cpu_s, cpu_sn = 0, 0.0, 0.0
        for item in work:
            k = min(args.cpu_pages, len(item["pages"]) - 1)
            o, t1, tn = cpu_sample(item["host"], item["pages"], 1, 1, k, pq.INT32)
            cpu_s += t1
            cpu_sn += tn
            cpu_vals += o.n_slots
            decode(item)
            g = dec.fetch()
            gv = g.values.view("<i4").reshape(-1)
            if not np.array_equal(gv[:o.n_slots], o.values.view("<i4").reshape(-1)):
                raise RuntimeError("GPU output differs from the oracle at w=%d" % item["w"])
            del g, gv, o
        cpu = {"value": round(cpu_vals * 4 / cpu_s / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
               "sample": "first %d pages (%d Mi values) of each of the %d widths, decoded by oracle/pq_oracle.c "
                         "(C restatement of the Go decoders; typed vectors instead of boxed interface{}, so faster than "
                         "the Go original), output checked equal to the GPU's" % (args.cpu_pages, args.cpu_pages * page_values >> 20, len(widths)),
               "threads": {"value": round(cpu_vals * 4 / cpu_sn / 1e9, 4), "cores": cores,
                           "note": "same sample, one page range per thread (the reference runs one goroutine per column, "
                                   "reader/reader.go:306-327); host cores available to the process: %d" % cores}}
    for item in work:
        item["host"] = None

    # warm-up, then exactly --steps timed steps between barrier + device sync on both sides, MAX over ranks
    collect_flag = [False]

    def one_step():
        step(collect=collect_flag[0])

    sharding.timed_region(one_step, 0, args.warmup, sync=torch.cuda.synchronize, dist=dist, device=red_dev, drain=drain)
    collect_flag[0] = True
    elapsed = sharding.timed_region(one_step, args.steps, 0, sync=torch.cuda.synchronize, dist=dist, device=red_dev,
                                    drain=lambda: drain(True))

    out_bytes_step = len(widths) * n_values * 4
    value = world * out_bytes_step * args.steps / elapsed / 1e9
    alg_step = sum(item["in_bytes"] + n_values * 4 + item["dict_n"] * 4 for item in work)   # packed input + decoded output + dictionary
    step_gbps = alg_step * args.steps / elapsed / 1e9

    # ---- the same chunks once more, one at a time on one stream: per-kernel durations without other chunks' kernels
    # sharing the CUs (the durations taken inside the timed region above include that sharing) ----
    iso = {}
    iso_wall = {}
    for item in work:
        for _ in range(args.iso_reps):
            t0 = time.perf_counter()
            submit(item, dec)
            dec.wait()
            dt = time.perf_counter() - t0
            iso_wall[item["w"]] = min(dt, iso_wall.get(item["w"], 1e9))
            for name, (ms, n) in dec.kernel_times().items():
                e = iso.setdefault((item["w"], name), [0.0, 0])
                e[0] += ms
                e[1] += n

    # ---- the dominant kernel, from HIP events around each launch: the one-pass kernel (run-header walk, bit unpack,
    # dictionary gather and stores of a page by one workgroup); what it leaves to the run-table path is listed beside it ----
    def expand_stats(kt):
        per_width, tot_ms, tot_bytes, tot_n = {}, 0.0, 0.0, 0
        for item in work:
            w = item["w"]
            ms, n = kt.get((w, "k_page_fused"), (0.0, 0))
            exp_ms, _ = kt.get((w, "k_hybrid_expand<dict>"), (0.0, 0))
            pre_ms, _ = kt.get((w, "k_hybrid_prescan(dict)"), (0.0, 0))
            alg = item["in_bytes"] + n_values * 4 + item["dict_n"] * 4
            if n:
                per_width[str(w)] = {"one_pass_ms": round(ms / n, 4), "run_table_path_ms": round((pre_ms + exp_ms) / n, 4),
                                     "one_pass_GBps": round(alg / (ms / n) / 1e6, 1),
                                     "alg_bytes_per_value": round(alg / n_values, 4)}
                tot_ms += ms
                tot_bytes += alg * n
                tot_n += n
        gbps = tot_bytes / tot_ms / 1e6 if tot_ms else 0.0
        return per_width, gbps, (tot_ms / tot_n if tot_n else None)

    all_ms = {}
    for (w, name), (ms, n) in ktimes.items():
        all_ms[name] = all_ms.get(name, 0.0) + ms
    per_width, launch_gbps, avg_ms = expand_stats(ktimes)
    exp_total_ms = sum(ms for (w, name), (ms, n) in ktimes.items() if name == "k_page_fused")
    in_flight = exp_total_ms / (elapsed * 1e3) if elapsed else 0.0
    iso_pw, iso_gbps, iso_avg = expand_stats(iso)
    for item in work:
        if str(item["w"]) in iso_pw:
            iso_pw[str(item["w"])]["chunk_wall_ms"] = round(iso_wall[item["w"]] * 1e3, 4)
    traffic, traffic_src = measured_traffic(args, widths)
    roofline = {"bound": "hbm", "kernel": "the whole step: one-pass kernel (run-header walk, bit unpack, dictionary gather, stores) and "
                                          "page bookkeeping of the %d chunks" % len(widths),
                "achieved": round(step_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(step_gbps / HBM_PEAK_GBPS, 4),
                "definition": "algorithmic bytes of one step (page bytes + 4 B per value + dictionaries = %d) / ms_per_step" % alg_step,
                "traffic": traffic, "traffic_source": traffic_src,
                "one_pass_kernel": {
                    "in_timed_region": {"achieved": round(launch_gbps, 1), "frac": round(launch_gbps / HBM_PEAK_GBPS, 4),
                                        "avg_launch_ms": round(avg_ms, 4) if avg_ms else None, "mean_in_flight": round(in_flight, 2),
                                        "per_width": per_width,
                                        "note": "HIP events on each handle's stream; %d chunks are in flight and on average "
                                                "'mean_in_flight' launches of the kernel share the CUs, so a launch lasts that much "
                                                "longer than alone" % (n_handles * depth)},
                    "isolated": {"achieved": round(iso_gbps, 1), "frac": round(iso_gbps / HBM_PEAK_GBPS, 4),
                                 "avg_launch_ms": round(iso_avg, 4) if iso_avg else None, "per_width": iso_pw}},
                "kernel_ms_share": {k: round(v / max(sum(all_ms.values()), 1e-9), 4) for k, v in all_ms.items()}}

    for item in work:       # the headline's inputs make room for the extras
        item["dev"] = None
    torch.cuda.empty_cache()

    extra = {}
    if not args.no_extra:
        extra = run_extras(args, dec, local_rank, rank, world, dist, red_dev, cpu_sample if orc is not None else None, cpu_entry, to_device)

    # HBM bytes per decode call of the other encodings, from the committed counter summary of these very workloads
    # (profiles/collect.sh: rocprofv3 cannot run inside this process), next to their algorithmic bytes
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tex = json.load(f).get("extras", {})
    except (OSError, ValueError):
        tex = {}
    for name, e in extra.items():
        t = tex.get(name)
        if t and isinstance(e, dict) and e.get("in_bytes") and t.get("in_bytes") == e.get("in_bytes") and t.get("out_bytes") == e.get("out_bytes"):
            e["traffic"] = t["traffic"]
            e["traffic_over_algorithmic"] = round(t["traffic"] / (e["in_bytes"] + e["out_bytes"]), 3)
            e["traffic_source"] = "profiles/traffic.json (builder's rocprofv3 --pmc passes over this workload, not this run): 2*FETCH_SIZE + WRITE_SIZE over all kernels of one decode call"

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
        dist.destroy_process_group()


def measured_traffic(args, widths):
    """HBM bytes per launch of the expand kernel from the memory-side L2 counters.  The counters need rocprofv3, which
    cannot run inside this process, so the figure comes from the committed summary of profiles/collect.sh for this
    very workload shape; (None, None) when no summary matches."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, None
    if t.get("values_per_gpu") != args.values or t.get("page_values") != args.page_values or t.get("widths") != widths:
        return None, None
    return t.get("traffic"), "profiles/traffic.json (builder's rocprofv3 --pmc run of this workload, not this run); per launch of k_page_fused"


def run_extras(args, dec, local_rank, rank, world, dist, red_dev, cpu_sample, cpu_entry, to_device):
    """Secondary numbers kept next to the headline: the other encodings device-resident with the oracle on a sample of
    each beside them, the two strong-scaling legs (row groups dealt over the ranks), and host-to-host rates."""
    import numpy as np
    import torch
    import parquet_go_amd as pq
    from parquet_go_amd import sharding
    from gen import pqgen
    extra = {}
    solo = rank == 0    # single-GPU extras run on rank 0 only; the scaling legs run on every rank
    only = set(x for x in args.only_extra.split(",") if x)

    def wanted(name):
        return not only or name in only

    def timed_device(chunk, pages, ptype, max_def=0, max_rep=0, reps=3, dev=None):
        dev = to_device(chunk) if dev is None else dev
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

    def fixed_equal(g, o, esz):
        return np.array_equal(g.values[:o.n_slots], o.values) and np.array_equal(g.def_levels[:o.n_slots], o.def_levels) and \
            np.array_equal(g.rep_levels[:o.n_slots], o.rep_levels)

    # ---- configs[2]: DELTA_BINARY_PACKED INT64 (and the INT32 variant) ----
    for esz, name, ptype in ((8, "delta_int64", pq.INT64), (4, "delta_int32", pq.INT32)):
        if not solo:
            break
        if not wanted(name):
            continue
        try:
            n = args.delta_values
            chunk, pages = pqgen.delta_chunk(0x5EED0003, n, 12, 128 << 10, esz) if esz == 8 else \
                pqgen.delta_chunk(0x5EED0003, n, 12, 128 << 10, esz, 0)        # (INT32 values wrap around, as the reference's do)
            best, kt, r = timed_device(chunk, pages, ptype)
            in_bytes = int(pages["length"].sum())
            kms = kt.get("k_delta_decode", (0.0, 1))[0]
            e = {"workload": "configs[2]: INT%d DELTA_BINARY_PACKED, %d Mi values, 128-value blocks / 4 miniblocks, "
                             "delta width 12, pages of 128 Ki values" % (8 * esz, n >> 20),
                 "decoded_GBps": round(n * esz / best / 1e9, 1), "ms": round(best * 1e3, 3),
                 "alg_GBps": round((in_bytes + n * esz) / best / 1e9, 1),
                 "roofline_frac": round((in_bytes + n * esz) / best / 1e9 / HBM_PEAK_GBPS, 4),
                 "in_bytes": in_bytes, "out_bytes": n * esz,
                 "kernel_ms": round(kms, 3),
                 "kernel_roofline_frac": round((in_bytes + n * esz) / kms / 1e6 / HBM_PEAK_GBPS, 4) if kms else None}
            if cpu_sample:
                k = min(256, len(pages))
                o, t1, tn = cpu_sample(chunk, pages, 0, 0, k, ptype)
                dec.submit(chunk[:int(pages["offset"][k - 1] + pages["length"][k - 1])], pages[:k], ptype)
                g = dec.fetch()
                assert g.status == 0 and np.array_equal(g.values, o.values), "delta sample differs from the oracle"
                e["cpu_baseline"] = cpu_entry(o.n_slots * esz, t1, tn, "first %d pages, output checked equal to the GPU's" % k)
            extra[name] = e
        except Exception as ex:  # the extras never take the headline down
            extra[name] = {"error": repr(ex)}

    # ---- PLAIN INT64: the copy-rate reference point ----
    if solo and wanted("plain_int64"):
        try:
            n = 256 << 20
            vals = np.arange(n, dtype=np.int64)
            pages = np.zeros(256, pq.PAGE_DESC_DTYPE)
            per = n // 256
            pages["page_type"], pages["encoding"], pages["num_values"] = 0, 0, per
            pages["offset"], pages["length"] = np.arange(256, dtype=np.int64) * per * 8, per * 8
            chunk = vals.view(np.uint8)
            best, kt, r = timed_device(chunk, pages, pq.INT64)
            extra["plain_int64"] = {"workload": "PLAIN INT64, %d Mi values in 256 pages" % (n >> 20),
                                    "decoded_GBps": round(n * 8 / best / 1e9, 1), "ms": round(best * 1e3, 3),
                                    "roofline_frac": round(2 * n * 8 / best / 1e9 / HBM_PEAK_GBPS, 4)}
            del vals, chunk
        except Exception as ex:
            extra["plain_int64"] = {"error": repr(ex)}

    # ---- configs[0]-shaped: what the reference's own writer produces (layout/dictpage.go:155-158,
    # encodingwrite.go:183-214): dictionary indices at width 32 as RLE runs only, pages of about 8 KiB ----
    if solo and wanted("parquet_go_written"):
        try:
            n = 64 << 20
            chunk, pages = pqgen.dict_chunk(0x5EED0007, n, 32, 4096, 2048, 1, 2)
            best, kt, r = timed_device(chunk, pages, pq.INT32)
            e = {"workload": "INT32 dictionary chunk as parquet-go writes it: width-32 RLE runs only (1-2 values per run), "
                             "%d Mi values in %d pages of 2048 values" % (n >> 20, len(pages) - 1),
                 "decoded_GBps": round(n * 4 / best / 1e9, 1), "ms": round(best * 1e3, 3),
                 "in_bytes": int(pages["length"].sum()),
                 "kernel_ms": {k_: round(v[0], 3) for k_, v in kt.items()}}
            # two handles taking turns: the host side of one call (page table, descriptor upload) overlaps the kernels
            # of the other, as it does for a reader that streams row groups
            d2 = pq.Decoder(local_rank)
            dev = to_device(chunk)
            torch.cuda.synchronize()
            pair = [dec, d2]
            for rep in range(2):
                t0 = time.perf_counter()
                for i in range(8):
                    d = pair[i & 1]
                    if i >= 2:
                        assert d.wait().status == 0
                    d.submit_device(dev.data_ptr(), chunk.size, pages, pq.INT32)
                for d in pair:
                    assert d.wait().status == 0
                dt = (time.perf_counter() - t0) / 8
            e["pipelined_GBps"] = round(n * 4 / dt / 1e9, 1)
            e["pipelined_ms"] = round(dt * 1e3, 3)
            d2.close()
            if cpu_sample:
                k = 2048
                o, t1, tn = cpu_sample(chunk, pages, 1, 1, k, pq.INT32)
                e["cpu_baseline"] = cpu_entry(o.n_slots * 4, t1, tn, "first %d pages" % k)
            extra["parquet_go_written"] = e
            del dev
        except Exception as ex:
            extra["parquet_go_written"] = {"error": repr(ex)}

    # ---- configs[4]: nested levels (optional list of optional int64), PLAIN values, one merged chunk ----
    if solo and wanted("list_int64"):
        try:
            n = args.list_slots
            chunk, pages = pqgen.list_chunk(0x5EED0005, n, 128 << 10)
            best, kt, r = timed_device(chunk, pages, pq.INT64, 3, 1)
            out_bytes = n * (4 + 4 + 8)
            in_bytes = int(pages["length"].sum())
            e = {"workload": "configs[4] merged: optional list<optional int64>, %d Mi level slots (max_def 3, max_rep 1) in one "
                             "chunk, pages of 128 Ki slots, PLAIN values; output = rep + def (int32 each) + slot-aligned int64 values"
                             % (n >> 20),
                 "decoded_GBps": round(out_bytes / best / 1e9, 1), "ms": round(best * 1e3, 3),
                 "roofline_frac": round((in_bytes + out_bytes) / best / 1e9 / HBM_PEAK_GBPS, 4),
                 "rows": int(r.num_rows), "nulls": int(r.num_nulls),
                 "in_bytes": in_bytes, "out_bytes": out_bytes,
                 "kernel_ms": {k_: round(v[0], 3) for k_, v in kt.items()}}
            if cpu_sample:
                k = min(32, len(pages))
                o, t1, tn = cpu_sample(chunk, pages, 0, 0, k, pq.INT64, 3, 1)
                e["cpu_baseline"] = cpu_entry(o.n_slots * 16, t1, tn, "first %d pages" % k)
            extra["list_int64"] = e
        except Exception as ex:
            extra["list_int64"] = {"error": repr(ex)}

    # ---- configs[3]: key-like strings, DELTA_LENGTH_BYTE_ARRAY and DELTA_BYTE_ARRAY ----
    for mode, name in ((0, "delta_length_byte_array"), (1, "delta_byte_array"), (3, "delta_byte_array_sorted_keys")):
        if not solo:
            break
        if not wanted(name):
            continue
        try:
            n = args.string_values
            chunk, pages = pqgen.strings_chunk(0x5EED0004, n, 64 << 10, mode)
            in_bytes = int(pages["length"].sum())
            best, kt, r = timed_device(chunk, pages, pq.BYTE_ARRAY)
            out_bytes = int(r.bytes_len) + 8 * (n + 1)
            what = ("sharing prefixes with their predecessor (prefix length U[0, len-1])" if mode < 3 else
                    "sorted keys: a zero-padded counter and a suffix, so that no string inside a page has prefix length 0")
            e = {"workload": "configs[3]: %d M strings of 8-64 bytes %s, pages of 64 Ki strings; output = int64 offsets + bytes"
                             % (n // 1000000, what),
                 "decoded_GBps": round(out_bytes / best / 1e9, 1), "ms": round(best * 1e3, 3),
                 "in_bytes": in_bytes, "out_bytes": out_bytes,
                 "alg_GBps": round((in_bytes + out_bytes) / best / 1e9, 1),
                 "roofline_frac": round((in_bytes + out_bytes) / best / 1e9 / HBM_PEAK_GBPS, 4),
                 "kernel_ms": {k_: round(v[0], 3) for k_, v in kt.items()}}
            if cpu_sample:
                k = min(64, len(pages))
                o, t1, tn = cpu_sample(chunk, pages, 0, 0, k, pq.BYTE_ARRAY)
                dec.submit(chunk[:int(pages["offset"][k - 1] + pages["length"][k - 1])], pages[:k], pq.BYTE_ARRAY)
                g = dec.fetch()
                assert g.status == 0 and np.array_equal(g.offsets, o.offsets) and np.array_equal(g.bytes, o.bytes), "string sample differs"
                e["cpu_baseline"] = cpu_entry(int(o.offsets[-1]) + 8 * o.n_slots, t1, tn, "first %d pages, output checked equal to the GPU's" % k)
            extra[name] = e
        except Exception as ex:
            extra[name] = {"error": repr(ex)}

    # ---- strong-scaling legs: a fixed set of row-group chunks dealt over the ranks (sharding.assign_chunks), every
    # rank decoding its share; time = MAX over ranks.  Each rank generates only its own chunks. ----
    def strong_leg(name, n_chunks, make, decode_args, out_bytes_of, group, describe):
        if not wanted(name):
            return
        try:
            sizes = [1] * n_chunks
            mine = sharding.my_chunks(sizes, rank, world)
            uniq = {}
            items = []
            for idx in mine:
                key = idx % 128                      # at most 128 distinct chunks per rank are generated; beyond that they
                if key not in uniq:                  # are reused in turn (about 1 GB apart in time, far beyond the caches)
                    chunk, pages = make(key)
                    uniq[key] = (to_device(chunk), chunk.size, pages)
                items.append(uniq[key])
            torch.cuda.synchronize()
            handles = [dec, pq.Decoder(local_rank)]

            def run():
                pend = [None, None]
                for b, g0 in enumerate(range(0, len(items), group)):
                    h = handles[b & 1]
                    if pend[b & 1]:
                        assert h.wait().status == 0
                    part = items[g0:g0 + group]
                    h.submit_chunks_device([t.data_ptr() for t, _, _ in part], [nb for _, nb, _ in part], [p for _, _, p in part], *decode_args)
                    pend[b & 1] = True
                for k in range(2):
                    if pend[k]:
                        assert handles[k].wait().status == 0
            t = sharding.timed_region(run, 3, 1, sync=torch.cuda.synchronize, dist=dist, device=red_dev) / 3
            handles[1].close()
            if rank == 0:
                total_out = out_bytes_of(n_chunks)
                extra[name] = {"workload": describe, "chunks": n_chunks, "chunks_per_call": group, "n_gpus": world, "scaling": "strong",
                               "decoded_GBps": round(total_out / t / 1e9, 1), "ms": round(t * 1e3, 3)}
        except Exception as ex:
            if rank == 0:
                extra[name] = {"error": repr(ex)}

    rg_delta = 64
    per_rg = args.delta_values // rg_delta
    strong_leg("delta_int64_row_groups", rg_delta, lambda k: pqgen.delta_chunk(0x5EED1000 + k, per_rg, 12, 128 << 10),
               (pq.INT64,), lambda n: n * per_rg * 8, 8,
               "configs[2] as %d row-group chunks of %d Mi values (pages of 128 Ki values)" % (rg_delta, per_rg >> 20))
    strong_leg("list_int64_row_groups", args.row_groups, lambda k: pqgen.list_chunk(0x5EED2000 + k, 1 << 20, 128 << 10),
               (pq.INT64, 0, 3, 1), lambda n: n * (1 << 20) * 16, args.rg_group,
               "configs[4] as stated: %d row groups of 1 Mi level slots (8 pages of 128 Ki slots each), optional list<optional int64>, "
               "PLAIN values; output = rep + def + slot-aligned values" % args.row_groups)

    # ---- host to host ----
    if solo and wanted("host_to_host"):
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
        except Exception as ex:
            extra["host_to_host"] = {"error": repr(ex)}

    # ---- host to host for the other encodings: chunk read into the handle's page-locked input buffer, result read in
    # place from its page-locked memory (the fastest host route; PCIe both ways) ----
    if solo and wanted("host_to_host_others"):
        def e2e(chunk, pages, ptype, md, mr, out_bytes_of):
            best = None
            for i in range(3):
                buf = dec.input_buffer(chunk.size)
                buf[:] = chunk
                t1 = time.perf_counter()
                dec.submit(buf, pages, ptype, 0, md, mr)
                t = dec.fetch_pinned()
                dt = time.perf_counter() - t1
                assert t.status == 0
                best = dt if best is None else min(best, dt)
            ob = out_bytes_of(t)
            return {"decoded_GBps": round(ob / best / 1e9, 2), "ms": round(best * 1e3, 2), "in_bytes": int(chunk.size), "out_bytes": int(ob)}
        try:
            res = {}
            chunk, pages = pqgen.delta_chunk(0x5EED0013, 64 << 20, 12, 128 << 10, 8)
            res["delta_int64_64Mi"] = e2e(chunk, pages, pq.INT64, 0, 0, lambda t: t.n_slots * 8)
            chunk, pages = pqgen.strings_chunk(0x5EED0014, 16_000_000, 64 << 10, 0)
            res["delta_length_byte_array_16M_strings"] = e2e(chunk, pages, pq.BYTE_ARRAY, 0, 0, lambda t: t.offsets.nbytes + t.bytes.nbytes)
            chunk, pages = pqgen.strings_chunk(0x5EED0014, 16_000_000, 64 << 10, 1)
            res["delta_byte_array_16M_strings"] = e2e(chunk, pages, pq.BYTE_ARRAY, 0, 0, lambda t: t.offsets.nbytes + t.bytes.nbytes)
            chunk, pages = pqgen.list_chunk(0x5EED0015, 32 << 20, 128 << 10)
            res["list_int64_32Mi_slots"] = e2e(chunk, pages, pq.INT64, 3, 1, lambda t: t.n_slots * 16)
            res["workload"] = "page-locked input buffer of the handle (pqd_input_buffer) -> H2D -> decode -> D2H into the handle's page-locked memory (pqd_result_host)"
            extra["pinned_to_pinned_others"] = res
        except Exception as ex:
            extra["pinned_to_pinned_others"] = {"error": repr(ex)}
    return extra


if __name__ == "__main__":
    main()
