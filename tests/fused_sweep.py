"""Sweep of the one-pass kernel (k_page_fused) over random index streams: every bit width, dictionaries of any size,
run mixes from all-RLE to all-bit-packed, long runs that cross many windows of the walk, stretches of tiny runs, the
odd streams the kernel must hand to the general path, counts that end inside the last run, nulls.  Each chunk is
decoded with the kernel forced on and compared with the oracle.
python tests/fused_sweep.py <first_seed> <n_seeds>   (needs a GPU; not collected by pytest)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
import parquet_go_amd as pq
from oracle import oracle
import pagegen as pg
from test_gpu_parity import _runs_body, rand_dict, check_same

first, n_seeds = int(sys.argv[1]), int(sys.argv[2])
dec = pq.Decoder(0)
dec.set_option(pq.binding.OPT_FUSED, 1)
t0, fails, direct_total, stream_total = time.time(), [], 0, 0
for si in range(n_seeds):
    seed = first + 104729 * si
    rng = np.random.default_rng(seed)
    w = int(rng.integers(1, 33))
    ptype = [pg.INT32, pg.INT64, pg.FLOAT, pg.DOUBLE][int(rng.integers(0, 4))]
    nd = int(rng.integers(1, min(1 << min(w, 17), 70000) + 1))
    mix = {"bp": int(rng.integers(0, 60)), "rle": int(rng.integers(0, 60)), "rle2": int(rng.integers(0, 8)), "bp2": int(rng.integers(0, 6)),
           "short": int(rng.integers(0, 80)) * int(rng.integers(0, 2))}
    if sum(mix.values()) == 0: mix["bp"] = 1
    odd = int(rng.integers(0, 6))       # one chunk in six carries a page of an odd kind
    max_def = int(rng.integers(0, 3)) * int(rng.integers(0, 2))
    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 40)), lead=int(rng.integers(0, 17)))
    cb.dict_page(rand_dict(rng, ptype, nd), nd)
    n_pages = int(rng.integers(1, 4))
    for k in range(n_pages):
        n_target = int(rng.integers(1, [300, 20000, 150000][int(rng.integers(0, 3))]))
        m = dict(mix)
        if odd == 0 and k == n_pages - 1:
            m[["rle3", "empty", "emptybp"][int(rng.integers(0, 3))]] = max(1, sum(mix.values()) // 50)
        body, vals = _runs_body(rng, w, nd, n_target, m)
        nn = len(vals) - int(rng.integers(0, min(8, len(vals))))
        tail = bytes(int(rng.integers(1, 4))) if odd == 1 and k == 0 else b""
        if max_def:
            slots = nn + int(rng.integers(0, nn + 1))
            dl = np.full(slots, max_def, np.int64)
            dl[rng.choice(slots, slots - nn, replace=False)] = rng.integers(0, max_def, slots - nn)
            cb.data_page_v1(pg.ENC_RLE_DICTIONARY, slots, bytes([w]) + body + tail, b"", pg.level_block_v1(dl, max_def, rng, "mixed"))
        else:
            cb.data_page_v1(pg.ENC_RLE_DICTIONARY, nn, bytes([w]) + body + tail)
    chunk, pages = cb.finish()
    try:
        check_same(dec, oracle, chunk, pages, ptype, 0, max_def, 0)
        direct_total += dec.direct_streams(); stream_total += n_pages
    except AssertionError as e:
        fails.append("seed %d w %d nd %d ptype %d mix %s odd %d max_def %d: %s" % (seed, w, nd, ptype, mix, odd, max_def, str(e)[:200]))
    if si % 20 == 19:
        print("%d chunks, %d failures, %d of %d streams finished by the kernel, %.0f s" % (si + 1, len(fails), direct_total, stream_total, time.time() - t0), flush=True)
for f in fails[:30]: print(f)
print("done: %d chunks, %d failures" % (n_seeds, len(fails)))
sys.exit(1 if fails else 0)
