"""Sweep of nullable and repeated columns over random valid chunks: four level shapes, level streams in every style
(mixed, RLE only, bit-packed only, Arrow's), v1 and v2 pages of up to 150 000 slots, pages of nothing but nulls or of
none, PLAIN / dictionary / DELTA_BINARY_PACKED values of four physical types.  Compared with the oracle.
python tests/levels_sweep.py <first_seed> <n_seeds>   (needs a GPU; not collected by pytest)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
import parquet_go_amd as pq
from oracle import oracle
import pagegen as pg
from test_gpu_parity import check_same, PHYS_DT

first, n_seeds = int(sys.argv[1]), int(sys.argv[2])
B = pq.binding
decs = []
for variant in range(5):
    d = pq.Decoder(0)
    if variant == 4: d.set_option(B.OPT_PLAIN_SCATTER, 2); d.set_option(B.OPT_DIRECT_RLE, 0)
    if variant == 1: d.set_option(B.OPT_PRESCAN, 0); d.set_option(B.OPT_JUMP, 0)
    if variant == 2: d.set_option(B.OPT_PRESCAN, 1); d.set_option(B.OPT_DIRECT_RLE, 0)
    if variant == 3: d.set_option(B.OPT_PLAIN_SCATTER, 0); d.set_option(B.OPT_JUMP, 1)
    decs.append(d)
t0, fails = time.time(), []
for si in range(n_seeds):
    seed = first + 49979687 * si
    rng = np.random.default_rng(seed)
    ptype = [pg.INT32, pg.INT64, pg.FLOAT, pg.DOUBLE][int(rng.integers(0, 4))]
    dt = np.dtype(PHYS_DT[ptype])
    enc = ["plain", "dict", "delta"][int(rng.integers(0, 3))]
    if enc == "delta" and ptype not in (pg.INT32, pg.INT64): enc = "plain"
    max_def, max_rep = [(1, 0), (2, 1), (3, 2), (7, 0), (1, 1)][int(rng.integers(0, 5))]
    style = ["mixed", "rle", "bp", "arrow"][int(rng.integers(0, 4))]
    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 30)), lead=int(rng.integers(0, 9)))
    nd = int(rng.integers(1, 5000))
    if enc == "dict":
        cb.dict_page(rng.integers(0, 256, nd * dt.itemsize, dtype=np.uint8).tobytes(), nd)
    desc = []
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.integers(1, [20, 3000, 40000, 150000][int(rng.integers(0, 4))]))
        shape = int(rng.integers(0, 6))
        dl = rng.integers(0, max_def + 1, n)
        if shape == 0: dl[:] = max_def
        elif shape == 1: dl[:] = rng.integers(0, max_def)
        elif shape >= 3:      # long stretches of equal levels, as real columns have
            for r in rng.integers(0, n, max(1, n // 200)):
                dl[r:r + int(rng.integers(5, 600))] = rng.integers(0, max_def + 1)
        rl = np.zeros(n, np.int64)
        if max_rep:
            rl = rng.integers(0, max_rep + 1, n)
            if shape >= 4:
                for r in rng.integers(0, n, max(1, n // 300)):
                    rl[r:r + int(rng.integers(5, 400))] = rng.integers(0, max_rep + 1)
            rl[0] = 0
        nn = int((dl == max_def).sum())
        if enc == "plain":
            body, e = rng.integers(0, 256, nn * dt.itemsize, dtype=np.uint8).tobytes(), pg.ENC_PLAIN
        elif enc == "dict":
            w = max(1, (nd - 1).bit_length())
            body, e = pg.dict_indices_body(rng.integers(0, nd, nn), w, rng, ["mixed", "rle", "bp"][int(rng.integers(0, 3))]), pg.ENC_RLE_DICTIONARY
        else:
            it = np.int64 if ptype == pg.INT64 else np.int32
            body = pg.delta_encode_fast(np.cumsum(rng.integers(0, 1000, nn)).astype(it), it) if nn else b""
            e = pg.ENC_DELTA_BINARY_PACKED
        v2 = bool(rng.integers(0, 2))
        if v2:
            cb.data_page_v2(e, n, body, pg.hybrid_encode(rl, int(max_rep).bit_length(), rng, style) if max_rep else b"",
                            pg.hybrid_encode(dl, int(max_def).bit_length(), rng, style))
        else:
            cb.data_page_v1(e, n, body, pg.level_block_v1(rl, max_rep, rng, style) if max_rep else b"",
                            pg.level_block_v1(dl, max_def, rng, style))
        desc.append((n, shape, "v2" if v2 else "v1"))
    chunk, pages = cb.finish()
    if os.environ.get("SWEEP_VERBOSE"):
        print("seed %d ptype %d %s levels (%d, %d) %s pages %s" % (seed, ptype, enc, max_def, max_rep, style, desc), flush=True)
    try:
        g, o = check_same(decs[si % 5], oracle, chunk, pages, ptype, 0, max_def, max_rep)
        assert o.status == 0, "oracle status %d" % o.status
    except AssertionError as e:
        fails.append("seed %d variant %d ptype %d %s levels (%d, %d) %s pages %s: %s" % (seed, si % 5, ptype, enc, max_def, max_rep, style, desc, str(e)[:200]))
    if si % 20 == 19:
        print("%d chunks, %d failures, %.0f s" % (si + 1, len(fails), time.time() - t0), flush=True)
for f in fails[:30]: print(f)
print("done: %d chunks, %d failures" % (n_seeds, len(fails)))
sys.exit(1 if fails else 0)
