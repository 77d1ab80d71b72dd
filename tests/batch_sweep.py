"""Sweep of calls that hold several chunks of a column: 2 to 7 valid chunks of one kind and level shape (the base cases
of tests/fuzzcases.py), now and then one without pages, handed over as host buffers or as separate device buffers;
compared with the oracle's results for the chunks read one after the other.
python tests/batch_sweep.py <first_seed> <n_batches>   (needs a GPU; not collected by pytest)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import torch
import parquet_go_amd as pq
from oracle import oracle
import pagegen as pg
import fuzzcases
from test_gpu_batch import concat_oracle, same_table

first, n_batches = int(sys.argv[1]), int(sys.argv[2])
dec = pq.Decoder(0)
t0, fails = time.time(), []
for bi in range(n_batches):
    rng = np.random.default_rng(first + bi)
    kind = fuzzcases.KINDS[bi % len(fuzzcases.KINDS)]
    parts, shape, seed = [], None, first * 1000 + bi * 100
    want = int(rng.integers(2, 8))
    while len(parts) < want:
        seed += 1
        cb, ptype, tl, md, mr = fuzzcases.base_case(kind, np.random.default_rng(seed))
        if shape is None: shape = (ptype, tl, md, mr)
        if (ptype, tl, md, mr) != shape: continue
        parts.append(cb.finish())
    if rng.integers(0, 4) == 0:
        parts.insert(int(rng.integers(0, len(parts) + 1)), (b"", np.zeros(0, pg.PAGE_DESC_DTYPE)))
    ptype, tl, md, mr = shape
    try:
        o = concat_oracle(oracle, [p for p in parts if len(p[1])], ptype, tl, md, mr)
        if bi % 2:
            keep = []
            for c, _ in parts:
                t = torch.zeros(len(c) + 64, dtype=torch.uint8, device="cuda")
                if len(c): t[:len(c)].copy_(torch.from_numpy(np.frombuffer(c, np.uint8).copy()))
                keep.append(t)
            torch.cuda.synchronize()
            dec.submit_chunks_device([t.data_ptr() for t in keep], [len(c) for c, _ in parts], [p for _, p in parts], ptype, tl, md, mr)
        else:
            dec.submit_chunks([c for c, _ in parts], [p for _, p in parts], ptype, tl, md, mr)
        g = dec.fetch()
        same_table(g, o)
    except AssertionError as e:
        fails.append("batch %d %s shape %s chunks %d: %s" % (first + bi, kind, shape, len(parts), str(e)[:200]))
    if bi % 100 == 99:
        print("%d batches, %d failures, %.0f s" % (bi + 1, len(fails), time.time() - t0), flush=True)
for f in fails[:30]: print(f)
print("done: %d batches, %d failures" % (n_batches, len(fails)))
sys.exit(1 if fails else 0)
