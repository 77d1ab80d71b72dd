"""Seeded malformed chunks: valid pages of every encoding on the path, then a few random edits (byte flips inside
page bodies, page lengths cut short or stretched, value counts and v2 level-section lengths changed).  The oracle
and the GPU library must agree on every one of them: same status and failing page, and the same table when the
reference would have decoded it.  The reference's own error behaviour is what the oracle restates
(encoding/encodingread.go, layout/page.go:633-937)."""
import numpy as np

import pagegen as pg


def _levels(rng, n, max_def, max_rep):
    dl = rng.integers(0, max_def + 1, n) if max_def else np.zeros(n, np.int64)
    for r in rng.integers(0, n, max(1, n // 100)):
        dl[r:r + int(rng.integers(5, 100))] = max_def
    rl = np.zeros(n, np.int64)
    if max_rep:
        rl = rng.integers(0, max_rep + 1, n)
        rl[0] = 0
    return dl, rl


def _strings(rng, n, lo=0, hi=24):
    lens = rng.integers(lo, hi + 1, n)
    return [bytes(rng.integers(97, 123, int(k), dtype=np.uint8)) for k in lens]


def _keys(rng, n, hi=40):
    out, prev = [], b""
    for _ in range(n):
        ln = int(rng.integers(0, hi + 1))
        pl = int(rng.integers(0, min(len(prev), ln) + 1))
        cur = prev[:pl] + bytes(rng.integers(97, 123, ln - pl, dtype=np.uint8))
        out.append(cur)
        prev = cur
    return out


KINDS = ["dict32", "dict_ba", "plain64", "plain_ba", "delta32", "delta64", "dlba", "dba", "bool", "bss",
         "dict_flba", "plain96", "dict64", "dict_short"]


def base_case(kind, rng):
    """-> (ChunkBuilder with 2-4 valid pages, ptype, type_length, max_def, max_rep)"""
    max_def, max_rep = [(0, 0), (1, 0), (2, 1), (3, 2)][int(rng.integers(0, 4))]
    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 5)), lead=int(rng.integers(0, 4)))
    ptype, tl = pg.INT32, 0
    words = None
    if kind == "dict32":
        w = int(rng.choice([1, 2, 3, 5, 8, 11, 16]))
        nd = min(1 << w, 3000)
        cb.dict_page(rng.integers(0, 256, nd * 4, dtype=np.uint8).tobytes(), nd)
    elif kind == "dict_short":     # runs of a few values each, in the indices and in the levels
        w = int(rng.choice([1, 2, 3, 5, 8]))
        nd = 1 << w
        cb.dict_page(rng.integers(0, 256, nd * 4, dtype=np.uint8).tobytes(), nd)
    elif kind == "dict_ba":
        ptype = pg.BYTE_ARRAY
        words = _strings(rng, 40, 0, 12)
        cb.dict_page(pg.plain_byte_array(words), len(words))
    elif kind == "dict_flba":
        ptype, tl = pg.FIXED_LEN_BYTE_ARRAY, 5
        nd = 300
        cb.dict_page(rng.integers(0, 256, nd * tl, dtype=np.uint8).tobytes(), nd)
    elif kind == "dict64":
        ptype = pg.INT64
        nd = 20000
        cb.dict_page(rng.integers(0, 256, nd * 8, dtype=np.uint8).tobytes(), nd)
    elif kind == "plain96":
        ptype = pg.INT96
    elif kind in ("plain64", "delta64"):
        ptype = pg.INT64
    elif kind in ("plain_ba", "dlba", "dba"):
        ptype = pg.BYTE_ARRAY
    elif kind == "bool":
        ptype = pg.BOOLEAN
    elif kind == "bss":
        ptype = pg.DOUBLE
    mg = 2 if kind == "dict_short" else 63     # bit-packed groups per run at most
    for _ in range(int(rng.integers(2, 5))):
        n = int(rng.choice([1, 7, 64, 300, 1500, 5000]))
        if kind == "dict_short":
            n = int(rng.choice([700, 3000, 20000]))
        dl, rl = _levels(rng, n, max_def, max_rep)
        nn = int((dl == max_def).sum())
        if kind == "dict_short":
            idx = np.repeat(rng.integers(0, nd, nn // 3 + 1), rng.integers(1, 6, nn // 3 + 1))[:nn]
            idx = np.concatenate([idx, rng.integers(0, nd, nn - len(idx))])
            # (every other page the way the reference's writer lays it out: RLE runs only)
            body, enc = bytes([w]) + pg.hybrid_encode(idx, w, rng, "rle" if rng.integers(0, 2) else "mixed", mg), pg.ENC_RLE_DICTIONARY
        elif kind == "dict32":
            body, enc = pg.dict_indices_body(rng.integers(0, nd, nn), w, rng), pg.ENC_RLE_DICTIONARY
        elif kind == "dict_ba":
            body, enc = pg.dict_indices_body(rng.integers(0, len(words), nn), 6, rng), pg.ENC_PLAIN_DICTIONARY
        elif kind == "dict_flba":
            body, enc = pg.dict_indices_body(rng.integers(0, nd, nn), 9, rng), pg.ENC_RLE_DICTIONARY
        elif kind == "dict64":
            body, enc = pg.dict_indices_body(rng.integers(0, nd, nn), 15, rng), pg.ENC_PLAIN_DICTIONARY
        elif kind == "plain96":
            body, enc = rng.integers(0, 256, nn * 12, dtype=np.uint8).tobytes(), pg.ENC_PLAIN
        elif kind == "plain64":
            body, enc = rng.integers(-2**62, 2**62, nn, dtype=np.int64).tobytes(), pg.ENC_PLAIN
        elif kind == "plain_ba":
            body, enc = pg.plain_byte_array(_strings(rng, nn)), pg.ENC_PLAIN
        elif kind == "delta32":
            body, enc = pg.delta_encode(rng.integers(-2**20, 2**20, nn).astype(np.int32), np.int32), pg.ENC_DELTA_BINARY_PACKED
        elif kind == "delta64":
            body, enc = pg.delta_encode(np.cumsum(rng.integers(-1000, 5000, nn)).astype(np.int64), np.int64), pg.ENC_DELTA_BINARY_PACKED
        elif kind == "dlba":
            body, enc = pg.dlba_encode(_strings(rng, nn)), pg.ENC_DELTA_LENGTH_BYTE_ARRAY
        elif kind == "dba":
            body, enc = pg.dba_encode(_keys(rng, nn)), pg.ENC_DELTA_BYTE_ARRAY
        elif kind == "bool":
            body, enc = np.packbits(rng.integers(0, 2, nn, dtype=np.uint8), bitorder="little").tobytes(), pg.ENC_PLAIN
        else:
            v = rng.standard_normal(nn)
            body, enc = (np.frombuffer(v.tobytes(), np.uint8).reshape(nn, 8).T.copy().tobytes() if nn else b""), 9
        if nn == 0 and kind in ("dlba", "dba", "delta32", "delta64"):
            body = b""
        if rng.integers(0, 3) == 0:
            cb.data_page_v2(enc, n, body, pg.hybrid_encode(rl, max(1, int(max_rep).bit_length()), rng, "mixed", mg) if max_rep else b"",
                            pg.hybrid_encode(dl, max(1, int(max_def).bit_length()), rng, "mixed", mg) if max_def else b"")
        else:
            block = lambda lv, mx: (lambda b: len(b).to_bytes(4, "little") + b)(pg.hybrid_encode(lv, int(mx).bit_length(), rng, "mixed", mg))
            cb.data_page_v1(enc, n, body, block(rl, max_rep) if max_rep else b"", block(dl, max_def) if max_def else b"")
    return cb, ptype, tl, max_def, max_rep


def mutate(chunk, pages, rng):
    """A few edits in place of a copy; -> (chunk bytes, pages)."""
    buf = bytearray(chunk)
    pages = pages.copy()
    data = [i for i in range(pages.size) if pages["page_type"][i] != pg.DICTIONARY_PAGE] or list(range(pages.size))
    for _ in range(int(rng.integers(1, 4))):
        i = int(rng.choice(data)) if rng.integers(0, 8) else int(rng.integers(0, pages.size))   # one edit in eight may hit the dictionary page
        off, ln = int(pages["offset"][i]), int(pages["length"][i])
        op = int(rng.integers(0, 10))
        if op <= 4 and ln > 0:      # flip bytes, mostly near the start where the level sections and headers are
            for _ in range(int(rng.integers(1, 4))):
                span = ln if rng.integers(0, 2) else min(ln, 24)
                buf[off + int(rng.integers(0, span))] = int(rng.integers(0, 256))
        elif op == 5 and ln > 0:    # page cut short
            pages["length"][i] = int(rng.integers(0, ln))
        elif op == 6:               # page stretched into what follows (never past the chunk)
            pages["length"][i] = min(ln + int(rng.integers(1, 40)), len(buf) - off)
        elif op == 7:               # value count changed
            pages["num_values"][i] = max(0, int(pages["num_values"][i]) + int(rng.integers(-3, 20)))
        elif op == 8 and pages["page_type"][i] == pg.DATA_PAGE_V2:
            pages["rep_len"][i] = max(0, int(pages["rep_len"][i]) + int(rng.integers(-2, 3)))
            pages["def_len"][i] = max(0, int(pages["def_len"][i]) + int(rng.integers(-2, 3)))
        elif op == 9 and ln > 0:    # a run of zero or 0xFF bytes
            k = int(rng.integers(0, ln))
            fill = 0xFF if rng.integers(0, 2) else 0
            for j in range(k, min(ln, k + int(rng.integers(1, 12)))):
                buf[off + j] = fill
    return bytes(buf), pages


def cases(n_cases, seed=20240):
    """Yields (name, chunk, pages, ptype, type_length, max_def, max_rep)."""
    for k in range(n_cases):
        rng = np.random.default_rng(seed + k)
        kind = KINDS[k % len(KINDS)]
        cb, ptype, tl, max_def, max_rep = base_case(kind, rng)
        chunk, pages = cb.finish()
        chunk, pages = mutate(chunk, pages, rng)
        yield "%s#%d" % (kind, k), chunk, pages, ptype, tl, max_def, max_rep
