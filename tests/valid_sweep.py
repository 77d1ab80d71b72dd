"""Valid chunks of every encoding, physical type and level shape (tests/fuzzcases.py's base cases, unedited), many
seeds, the four kernel-variant settings; compared with the oracle.
python tests/valid_sweep.py <first_seed> <n_cases>   (needs a GPU; not collected by pytest)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
import parquet_go_amd as pq
from oracle import oracle
import fuzzcases
from test_gpu_parity import check_same

first, n_cases = int(sys.argv[1]), int(sys.argv[2])
B = pq.binding
decs = []
for variant in range(4):
    d = pq.Decoder(0)
    if variant == 1: d.set_option(B.OPT_PRESCAN, 0)
    if variant == 2: d.set_option(B.OPT_PRESCAN, 1); d.set_option(B.OPT_DIRECT_RLE, 0)
    if variant == 3: d.set_option(B.OPT_FUSED, 1)
    decs.append(d)
t0, fails = time.time(), []
for k in range(n_cases):
    rng = np.random.default_rng(first + k)
    kind = fuzzcases.KINDS[k % len(fuzzcases.KINDS)]
    cb, ptype, tl, max_def, max_rep = fuzzcases.base_case(kind, rng)
    chunk, pages = cb.finish()
    try:
        g, o = check_same(decs[(k // len(fuzzcases.KINDS)) % 4], oracle, chunk, pages, ptype, tl, max_def, max_rep)
        assert o.status == 0, "oracle status %d" % o.status
    except AssertionError as e:
        fails.append("case %d %s variant %d levels (%d, %d): %s" % (first + k, kind, (k // len(fuzzcases.KINDS)) % 4, max_def, max_rep, str(e)[:200]))
    if k % 200 == 199:
        print("%d cases, %d failures, %.0f s" % (k + 1, len(fails), time.time() - t0), flush=True)
for f in fails[:30]: print(f)
print("done: %d cases, %d failures" % (n_cases, len(fails)))
sys.exit(1 if fails else 0)
