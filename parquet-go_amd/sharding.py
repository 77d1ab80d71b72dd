"""Column chunks over several GPUs: one process per GPU, chunks dealt out by size, nothing exchanged.

Every (row group, column) chunk is self-contained - its own dictionary page (reader/columnbuffer.go:101 resets
DictPage per row group), its own page sequence, delta and prefix chains that restart at each page - so ranks never
need each other's data.  torch.distributed is used for the rendezvous, the barriers around a timed region and the
reduction of counters; the decode path has no collective.
"""
import time


def assign_chunks(byte_sizes, world):
    """Greedy longest-first bin packing: chunk i goes to the rank with the fewest bytes so far.
    Returns one list of chunk indices per rank (ascending within a rank); every rank computes the same answer."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(byte_sizes)), key=lambda i: (-int(byte_sizes[i]), i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(byte_sizes[i])
    for lst in out:
        lst.sort()
    return out


def my_chunks(byte_sizes, rank, world):
    return assign_chunks(byte_sizes, world)[rank]


def timed_region(step, steps, warmup, sync=None, dist=None, device=None, drain=None):
    """Runs `step()` warmup + steps times; the timed part is bracketed by barrier + device sync on both sides and
    the result is the MAX over ranks (seconds).  `sync` waits for the local device; `dist` is torch.distributed
    when a process group is up, else None.  `drain` completes work the steps left in flight (it runs inside the
    timed part, before the closing device sync)."""
    import torch

    def fence():
        if dist is not None:
            dist.barrier()
        if sync is not None:
            sync()

    for _ in range(warmup):
        step()
    if drain is not None and warmup:
        drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if drain is not None:
        drain()
    if sync is not None:
        sync()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def sum_over_ranks(values, dist=None, device=None):
    """Element-wise sum of a small list of integer counters (slots, rows, nulls, bytes) over the ranks."""
    if dist is None:
        return [int(v) for v in values]
    import torch
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]
