"""The entry point that allocates nothing against the allocating one over the fixed-width base cases of
tests/fuzzcases.py, valid and edited, one call after the other on the same work space.
python tests/noalloc_sweep.py <first_seed> <n_cases>   (needs a GPU; not collected by pytest)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import torch
import parquet_go_amd as pq
import pagegen as pg
import fuzzcases
from test_zz_no_alloc import run_both

first, n_cases = int(sys.argv[1]), int(sys.argv[2])
dec = pq.Decoder(0)
t0, fails, ran = time.time(), [], 0
for k in range(n_cases):
    rng = np.random.default_rng(first + k)
    kind = fuzzcases.KINDS[k % len(fuzzcases.KINDS)]
    cb, ptype, tl, md, mr = fuzzcases.base_case(kind, rng)
    if ptype == pg.BYTE_ARRAY: continue
    chunk, pages = cb.finish()
    if k % 3 == 0: chunk, pages = fuzzcases.mutate(chunk, pages, rng)
    ran += 1
    try:
        run_both(torch, pq, dec, chunk, pages, ptype, tl, md, mr)
    except AssertionError as e:
        fails.append("case %d %s: %s" % (first + k, kind, str(e)[:300]))
    if k % 200 == 199:
        print("%d cases (%d run), %d failures, %.0f s" % (k + 1, ran, len(fails), time.time() - t0), flush=True)
for f in fails[:30]: print(f)
print("done: %d cases run, %d failures" % (ran, len(fails)))
sys.exit(1 if fails else 0)
