"""Extended malformed-input sweep: more seeds than the suite runs, both run-header walks, the one-pass kernel forced on
for pages of any size.  python tests/fuzz_sweep.py <first_seed> <n_seeds> <cases_per_seed>  (needs a GPU; not collected by pytest)"""
import sys, os, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import torch, numpy as np
import parquet_go_amd as pq
from oracle import oracle
import fuzzcases
from test_fuzz import disagreement, REFUSALS
first, n_seeds, per = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bad, kinds, refused, total = [], 0, 0, 0
t0 = time.time()
for si in range(n_seeds):
    seed = first + 7919 * si
    dec = pq.Decoder(0)
    variant = si % 4
    if variant == 1: dec.set_option(pq.binding.OPT_PRESCAN, 0)
    if variant == 2: dec.set_option(pq.binding.OPT_PRESCAN, 1)
    if variant == 3: dec.set_option(pq.binding.OPT_FUSED, 1)
    for name, chunk, pages, ptype, tl, md, mr in fuzzcases.cases(per, seed):
        o = oracle.read_chunk(chunk, pages, ptype, tl, md, mr)
        g = dec.decode(chunk, pages, ptype, tl, md, mr)
        total += 1
        refused += g.status in REFUSALS
        msg = disagreement(g, o)
        if msg == "kind": kinds += 1
        elif msg: bad.append("seed %d variant %d %s: %s" % (seed, variant, name, msg))
    dec.close()
    print("seed %d variant %d done: %d cases, %d bad, %d other kind, %d refused, %.0f s" % (seed, variant, total, len(bad), kinds, refused, time.time() - t0), flush=True)
for b in bad[:40]: print(b)
sys.exit(1 if bad else 0)
