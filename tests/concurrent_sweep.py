"""Several handles with a decode in flight at once (the library picks other kernel variants in a crowd): eight handles,
each call on a chunk of its own (valid or edited base cases of tests/fuzzcases.py), all submitted before any is waited
for; every result compared with the oracle.
python tests/concurrent_sweep.py <first_seed> <n_rounds>   (needs a GPU; not collected by pytest)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
import parquet_go_amd as pq
from oracle import oracle
import fuzzcases
from test_fuzz import disagreement

first, n_rounds = int(sys.argv[1]), int(sys.argv[2])
decs = [pq.Decoder(0) for _ in range(8)]
t0, bad, kinds, total = time.time(), [], 0, 0
for r in range(n_rounds):
    jobs = []
    for k, dec in enumerate(decs):
        seed = first + r * 8 + k
        rng = np.random.default_rng(seed)
        kind = fuzzcases.KINDS[seed % len(fuzzcases.KINDS)]
        cb, ptype, tl, md, mr = fuzzcases.base_case(kind, rng)
        chunk, pages = cb.finish()
        if seed % 3 == 0: chunk, pages = fuzzcases.mutate(chunk, pages, rng)
        jobs.append((seed, kind, chunk, pages, ptype, tl, md, mr))
    for dec, (seed, kind, chunk, pages, ptype, tl, md, mr) in zip(decs, jobs):
        dec.submit(chunk, pages, ptype, tl, md, mr)
    for dec, (seed, kind, chunk, pages, ptype, tl, md, mr) in zip(decs, jobs):
        g = dec.fetch()
        o = oracle.read_chunk(chunk, pages, ptype, tl, md, mr)
        total += 1
        msg = disagreement(g, o)
        if msg == "kind": kinds += 1
        elif msg: bad.append("seed %d %s: %s" % (seed, kind, msg))
    if r % 50 == 49:
        print("%d rounds, %d calls, %d bad, %d other kind, %.0f s" % (r + 1, total, len(bad), kinds, time.time() - t0), flush=True)
for b in bad[:30]: print(b)
print("done: %d calls, %d bad" % (total, len(bad)))
sys.exit(1 if bad else 0)
