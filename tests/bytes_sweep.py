"""Sweep of the BYTE_ARRAY paths over random valid chunks: PLAIN, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY and
dictionary pages of 1 to 80 000 strings, sorted keys that never restart inside a page, random strings that often do,
empty strings, a few long ones, nulls.  Compared with the oracle.
python tests/bytes_sweep.py <first_seed> <n_seeds>   (needs a GPU; not collected by pytest)"""
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
t0, fails = time.time(), []


def strings(rng, n, style):
    if style == 0:      # sorted keys: a counter and a suffix
        start = int(rng.integers(0, 10**9))
        width = int(rng.integers(4, 17))
        suf = [b"", b"-x", b"/part"][int(rng.integers(0, 3))]
        return [b"key-%0*d%s" % (width, start + i * int(rng.integers(1, 3)), suf) for i in range(n)]
    if style == 1:      # random bytes of random length
        hi = int(rng.integers(1, 70))
        lens = rng.integers(0, hi + 1, n)
        blob = rng.integers(97, 123, int(lens.sum()) + 1, dtype=np.uint8).tobytes()
        out, p = [], 0
        for ln in lens.tolist():
            out.append(blob[p:p + ln]); p += ln
        return out
    if style == 2:      # few distinct words, many repeats (long shared prefixes, many empty suffixes)
        words = [bytes(rng.integers(97, 100, int(rng.integers(0, 30)), dtype=np.uint8)) for _ in range(int(rng.integers(1, 20)))]
        return [words[i] for i in rng.integers(0, len(words), n).tolist()]
    # a few long strings among short ones
    out = strings(rng, n, 1)
    for i in rng.integers(0, n, max(1, n // 500)).tolist():
        out[i] = bytes(rng.integers(65, 91, int(rng.integers(500, 3000)), dtype=np.uint8))
    return out


for si in range(n_seeds):
    seed = first + 32452843 * si
    rng = np.random.default_rng(seed)
    kind = ["plain", "dlba", "dba", "dict"][int(rng.integers(0, 4))]
    max_def = int(rng.integers(0, 3)) * int(rng.integers(0, 2))
    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 30)), lead=int(rng.integers(0, 9)))
    words = None
    if kind == "dict":
        words = strings(rng, int(rng.integers(1, 3000)), int(rng.integers(0, 3)))
        cb.dict_page(pg.plain_byte_array(words), len(words))
    desc = []
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.integers(1, [50, 5000, 80000][int(rng.integers(0, 3))]))
        style = int(rng.integers(0, 4))
        if style == 3 and n > 20000: style = 1
        if kind == "dict":
            w = max(1, (len(words) - 1).bit_length())
            body = pg.dict_indices_body(rng.integers(0, len(words), n), w, rng, "mixed")
            enc = pg.ENC_RLE_DICTIONARY
        else:
            ss = strings(rng, n, style)
            body = {"plain": pg.plain_byte_array, "dlba": pg.dlba_encode, "dba": pg.dba_encode}[kind](ss)
            enc = {"plain": pg.ENC_PLAIN, "dlba": pg.ENC_DELTA_LENGTH_BYTE_ARRAY, "dba": pg.ENC_DELTA_BYTE_ARRAY}[kind]
        if max_def:
            slots = n + int(rng.integers(0, n + 1))
            dl = np.full(slots, max_def, np.int64)
            dl[rng.choice(slots, slots - n, replace=False)] = rng.integers(0, max_def, slots - n)
            cb.data_page_v1(enc, slots, body, b"", pg.level_block_v1(dl, max_def, rng, "mixed"))
        else:
            cb.data_page_v1(enc, n, body)
        desc.append((n, style))
    chunk, pages = cb.finish()
    if os.environ.get("SWEEP_VERBOSE"):
        print("seed %d %s max_def %d pages %s" % (seed, kind, max_def, desc), flush=True)
    try:
        g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY, 0, max_def, 0)
        assert o.status == 0, "oracle status %d" % o.status
    except AssertionError as e:
        fails.append("seed %d %s max_def %d pages %s: %s" % (seed, kind, max_def, desc, str(e)[:200]))
    if si % 20 == 19:
        print("%d chunks, %d failures, %.0f s" % (si + 1, len(fails), time.time() - t0), flush=True)
for f in fails[:30]: print(f)
print("done: %d chunks, %d failures" % (n_seeds, len(fails)))
sys.exit(1 if fails else 0)
