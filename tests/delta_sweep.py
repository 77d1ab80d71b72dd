"""Sweep of the DELTA_BINARY_PACKED kernels over random valid streams: INT32 / INT64, pages from one value to a few
hundred thousand, stretches of blocks of one length (every miniblock forced to the same width, so that the walk of the
block headers takes its widest rounds), stretches whose lengths change block by block, width 0, deltas that wrap.
Compared with the oracle.
python tests/delta_sweep.py <first_seed> <n_seeds> [delta_blocks 0|1|2]   (needs a GPU; not collected by pytest)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
import parquet_go_amd as pq
from oracle import oracle
import pagegen as pg
from test_gpu_parity import check_same

first, n_seeds = int(sys.argv[1]), int(sys.argv[2])
dec = pq.Decoder(0)
if len(sys.argv) > 3:       # PQD_OPT_DELTA_BLOCKS: 0 a wavefront per page, 1 block-parallel, 2 block-parallel with every stream walked
    dec.set_option(pq.binding.OPT_DELTA_BLOCKS, int(sys.argv[3]))
t0, fails = time.time(), []
for si in range(n_seeds):
    seed = first + 15485863 * si
    rng = np.random.default_rng(seed)
    dtype, ptype = (np.int64, pg.INT64) if rng.integers(0, 2) else (np.int32, pg.INT32)
    bits = 64 if dtype == np.int64 else 32
    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 30)), lead=int(rng.integers(0, 9)))
    desc = []
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.integers(1, [40, 3000, 60000, 300000][int(rng.integers(0, 4))]))
        d = np.zeros(n, np.int64)
        pos = 0
        while pos < n:
            seg = int(rng.integers(1, max(2, n // int(rng.integers(1, 6)))))
            kind = int(rng.integers(0, 5))
            m = min(seg, n - pos)
            if kind == 0:        # one width for every miniblock of the stretch: extremes forced into every 32 deltas
                k = int(rng.integers(0, 25))
                x = rng.integers(0, 1 << k, m) if k else np.zeros(m, np.int64)
                if k:
                    x[0::32] = 0; x[1::32] = (1 << k) - 1
                base = int(rng.integers(-1000, 1000))
                d[pos:pos + m] = x + base
            elif kind == 1:      # widths that change all the time
                k = rng.integers(0, 20, (m + 31) // 32).repeat(32)[:m]
                d[pos:pos + m] = (rng.random(m) * (1 << k)).astype(np.int64) - int(rng.integers(0, 100))
            elif kind == 2:      # constant
                d[pos:pos + m] = int(rng.integers(-5, 5))
            elif kind == 3:      # anything the type holds
                info = np.iinfo(dtype)
                d[pos:pos + m] = rng.integers(info.min, info.max, m, dtype=np.int64 if bits == 64 else np.int64)
            else:                # small noise
                s = int(rng.integers(1, 5000))
                d[pos:pos + m] = rng.integers(-s, s + 1, m)
            pos += m
        with np.errstate(over="ignore"):
            vals = np.cumsum(d.astype(np.uint64)).astype(np.uint64)
        vals = vals.astype(np.uint32).view(np.int32) if bits == 32 else vals.view(np.int64)
        enc = pg.delta_encode(vals, dtype, trim=bool(rng.integers(0, 2))) if n < 3000 else pg.delta_encode_fast(vals, dtype)
        cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, n, enc)
        desc.append(n)
    chunk, pages = cb.finish()
    if os.environ.get("SWEEP_VERBOSE"):
        print("seed %d bits %d pages %s" % (seed, bits, desc), flush=True)
    try:
        g, o = check_same(dec, oracle, chunk, pages, ptype)
        assert o.status == 0, "oracle status %d" % o.status
    except AssertionError as e:
        fails.append("seed %d bits %d pages %s: %s" % (seed, bits, desc, str(e)[:200]))
    if si % 20 == 19:
        print("%d chunks, %d failures, %.0f s" % (si + 1, len(fails), time.time() - t0), flush=True)
for f in fails[:30]: print(f)
print("done: %d chunks, %d failures" % (n_seeds, len(fails)))
sys.exit(1 if fails else 0)
