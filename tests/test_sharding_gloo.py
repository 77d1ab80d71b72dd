"""Two ranks on the CPU (gloo): chunk assignment is a partition, the timed region takes the max over ranks, and the
counters of a sharded decode add up to the single-process ones.  The decode itself is played by the CPU oracle here
(no GPU in this test); on a GPU box the same host code drives parquet_go_amd.Decoder."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_chunks():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pagegen as pg
    rng = np.random.default_rng(77)
    chunks = []
    for k in range(7):
        cb = pg.ChunkBuilder()
        nd = 64
        cb.dict_page(rng.integers(-1000, 1000, nd, dtype=np.int32).tobytes(), nd)
        for _ in range(1 + k % 3):
            n = int(rng.integers(100, 4000)) * (k + 1)
            dl = (rng.random(n) < 0.9).astype(np.int64)
            body = pg.dict_indices_body(rng.integers(0, nd, int(dl.sum())), 6, rng)
            cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, body, b"", pg.level_block_v1(dl, 1, rng))
        chunks.append(cb.finish())
    return chunks


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parquet_go_amd import sharding
    from oracle import oracle as orc
    chunks = make_chunks()
    sizes = [len(c) for c, _ in chunks]
    mine = sharding.my_chunks(sizes, rank, world)
    tot = [0, 0, 0]
    calls = [0]

    def step():
        calls[0] += 1
        tot[:] = [0, 0, 0]
        for i in mine:
            c, pages = chunks[i]
            t = orc.read_chunk(c, pages, orc.INT32, 0, 1, 0)
            assert t.status == 0
            tot[0] += t.n_slots
            tot[1] += t.n_rows
            tot[2] += t.n_nulls

    elapsed = sharding.timed_region(step, steps=2, warmup=1, dist=dist)
    summed = sharding.sum_over_ranks(tot, dist)
    q.put((rank, mine, elapsed, summed, calls[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_assignment_is_a_balanced_partition():
    sys.path.insert(0, ROOT)
    from parquet_go_amd import sharding
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 8):
        sizes = rng.integers(1, 1 << 30, 64).tolist()
        parts = sharding.assign_chunks(sizes, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(64))
        loads = [sum(sizes[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(sizes)
    assert sharding.assign_chunks([], 4) == [[], [], [], []]
    with pytest.raises(ValueError):
        sharding.assign_chunks([1], 0)


def test_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, e0, sum0, calls0), (r1, mine1, e1, sum1, calls1) = res
    assert sorted(mine0 + mine1) == list(range(7)) and not set(mine0) & set(mine1)
    assert e0 == e1 and e0 > 0            # MAX over ranks: both see the same number
    assert sum0 == sum1 and calls0 == calls1 == 3
    # the same counters from one process
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    tot = [0, 0, 0]
    for c, pages in make_chunks():
        t = orc.read_chunk(c, pages, orc.INT32, 0, 1, 0)
        tot[0] += t.n_slots
        tot[1] += t.n_rows
        tot[2] += t.n_nulls
    assert sum0 == tot
