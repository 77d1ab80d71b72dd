"""GPU parity: libpqdecode.so (HIP kernels, through the C ABI) against the CPU oracle on the same
seeded chunks.  Bit-exact: values, levels, counts and error classes."""
import numpy as np
import pytest

import pagegen as pg

pytestmark = pytest.mark.gpu

PHYS_DT = {pg.INT32: "<i4", pg.INT64: "<i8", pg.FLOAT: "<f4", pg.DOUBLE: "<f8"}


@pytest.fixture(scope="module")
def shared_dec():
    import parquet_go_amd as pq
    d = pq.Decoder(0)
    yield d
    d.close()


@pytest.fixture
def dec(shared_dec):
    # one handle for the module (its work buffers are reused from test to test, as a reader reuses them from chunk
    # to chunk); kernel-variant choices a test pins are handed back to the library afterwards
    yield shared_dec
    for opt in range(10):
        shared_dec.set_option(opt, -1)


def check_same(dec, oracle, chunk, pages, ptype, type_length=0, max_def=0, max_rep=0):
    g = dec.decode(chunk, pages, ptype, type_length, max_def, max_rep)
    o = oracle.read_chunk(chunk, pages, ptype, type_length, max_def, max_rep)
    assert g.status == o.status, "status gpu=%d oracle=%d (gpu page %d, oracle page %d)" % (g.status, o.status, g.bad_page, o.bad_page)
    if o.status != 0:
        assert g.bad_page == o.bad_page
        return g, o
    assert (g.n_slots, g.n_rows, g.n_nulls) == (o.n_slots, o.n_rows, o.n_nulls)
    assert np.array_equal(g.def_levels, o.def_levels)
    assert np.array_equal(g.rep_levels, o.rep_levels)
    assert g.kind == o.kind
    if o.kind == "fixed":
        assert g.values.shape == o.values.shape
        if not np.array_equal(g.values, o.values):
            bad = np.nonzero((g.values != o.values).any(axis=1))[0]
            raise AssertionError("values differ at %d slots, first %d: gpu %s oracle %s" % (
                len(bad), bad[0], g.values[bad[0]].tolist(), o.values[bad[0]].tolist()))
    else:
        assert np.array_equal(g.offsets, o.offsets)
        assert np.array_equal(g.bytes, o.bytes)
    # the same result read in place from the handle's page-locked memory
    h = dec.fetch_pinned()
    assert h.status == 0 and h.kind == g.kind and (h.n_slots, h.n_rows, h.n_nulls) == (g.n_slots, g.n_rows, g.n_nulls)
    assert np.array_equal(h.def_levels, g.def_levels) and np.array_equal(h.rep_levels, g.rep_levels)
    if g.kind == "fixed":
        assert np.array_equal(h.values, g.values)
    else:
        assert np.array_equal(h.offsets, g.offsets) and np.array_equal(h.bytes, g.bytes)
    return g, o


def rand_dict(rng, ptype, n, type_length=0):
    if ptype in PHYS_DT:
        nbytes = np.dtype(PHYS_DT[ptype]).itemsize
    elif ptype == pg.INT96:
        nbytes = 12
    else:
        nbytes = type_length
    return rng.integers(0, 256, n * nbytes, dtype=np.uint8).tobytes()


@pytest.mark.parametrize("w", [0, 1, 2, 3, 4, 5, 7, 8, 11, 12, 15, 16, 17, 20, 24, 30, 31, 32])
def test_dict_int32_widths(dec, oracle, w):
    rng = np.random.default_rng(100 + w)
    nd = min(1 << w, 70000) if w else 1
    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 9)), lead=int(rng.integers(0, 5)))
    cb.dict_page(rand_dict(rng, pg.INT32, nd), nd)
    for n, style in ((1, "mixed"), (8, "bp"), (4095, "mixed"), (4096, "arrow"), (4097, "rle"), (20000, "mixed"), (70001, "bp")):
        idx = rng.integers(0, nd, n)
        for r in rng.integers(0, n, 10):
            idx[r:r + int(rng.integers(2, 400))] = idx[r]
        if style == "rle" and n > 3000:
            idx = np.repeat(idx[:n // 7 + 1], 7)[:n]
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, pg.dict_indices_body(idx, w, rng, style))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.INT32)


@pytest.mark.parametrize("ptype,tl", [(pg.INT64, 0), (pg.FLOAT, 0), (pg.DOUBLE, 0), (pg.INT96, 0),
                                      (pg.FIXED_LEN_BYTE_ARRAY, 5), (pg.FIXED_LEN_BYTE_ARRAY, 16)])
def test_dict_other_types(dec, oracle, ptype, tl):
    rng = np.random.default_rng(7 + ptype + tl)
    nd = 1000
    cb = pg.ChunkBuilder(gap=1)
    cb.dict_page(rand_dict(rng, ptype, nd, tl), nd)
    for n in (3, 5000, 12345):
        idx = rng.integers(0, nd, n)
        cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, n, pg.dict_indices_body(idx, 10, rng, "mixed"))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, ptype, tl)


def test_dict_large_dictionary_global_gather(dec, oracle):
    rng = np.random.default_rng(9)
    nd = 65536
    cb = pg.ChunkBuilder()
    cb.dict_page(rand_dict(rng, pg.INT32, nd), nd)
    idx = rng.integers(0, nd, 300000)
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, len(idx), pg.dict_indices_body(idx, 16, rng, "bp"))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.INT32)


@pytest.mark.parametrize("ptype,nd,w", [(pg.INT32, 20000, 15), (pg.INT32, 40000, 16), (pg.INT64, 9000, 14), (pg.INT64, 30000, 15),
                                         (pg.DOUBLE, 70000, 17), (pg.INT32, 300000, 19), (pg.INT96, 9000, 14)])
def test_dict_mid_and_large_dictionaries(dec, oracle, ptype, nd, w):
    # dictionaries between 48 KiB and 1 MiB take the kernel variant that keeps the head of the dictionary in LDS
    # and gathers the rest from memory; beyond that (and for 12-byte entries) everything is gathered from memory
    rng = np.random.default_rng(500 + nd)
    cb = pg.ChunkBuilder(gap=3)
    cb.dict_page(rand_dict(rng, ptype, nd), nd)
    for n, style in ((70000, "mixed"), (9001, "bp"), (300, "rle"), (50000, "arrow")):
        idx = rng.integers(0, nd, n)
        idx[rng.integers(0, n, 5)] = nd - 1
        idx[rng.integers(0, n, 5)] = 0
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, pg.dict_indices_body(idx, w, rng, style))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, ptype)
    assert o.status == 0
    # an index just past the end, in the part served from memory
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 16, pg.dict_indices_body(np.array([1, 2, nd, 3] * 4), w + 1, rng, "bp"))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, ptype)
    assert o.status == oracle.PANIC


def test_dict_parquet_go_style_width32_rle_only(dec, oracle):
    # what the reference's own writer emits: width 32, RLE runs only (layout/dictpage.go:155-158), 8 KiB pages
    rng = np.random.default_rng(10)
    nd = 500
    cb = pg.ChunkBuilder(gap=20)
    cb.dict_page(rand_dict(rng, pg.INT32, nd), nd)
    for _ in range(40):
        n = int(rng.integers(1500, 2100))
        idx = rng.integers(0, nd, n)
        cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, n, bytes([32]) + oracle.write_rle_int32(idx.astype(np.int32), 32))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.INT32)
    # every one of these pages (7 to 10 KiB of runs, more than one round of the kernel each) is expanded straight from
    # the page; falling back to the general path would give the same values and hide a fault in the direct one
    assert dec.direct_streams() == 40


@pytest.mark.parametrize("direct", ["1", "0"])
@pytest.mark.parametrize("fused", ["1", "0"])
def test_index_out_of_range_in_a_run_the_page_never_reaches(dec, oracle, direct, fused):
    # ReadRLEBitPackedHybrid stops reading once it has the page's values (encodingread.go:240-268): an index beyond
    # the dictionary in a run behind them is never looked up; in the run that the count ends in, it is
    dec.set_option(1, int(direct))           # PQD_OPT_DIRECT_RLE
    dec.set_option(7, int(fused))            # PQD_OPT_FUSED
    rng = np.random.default_rng(4711)
    nd = 50
    for w in (8, 32):
        nb = (w + 7) // 8
        run = lambda c, v: bytes([c << 1]) + int(v).to_bytes(4, "little")[:nb]
        for last_ok in (True, False):
            cb = pg.ChunkBuilder(gap=3)
            cb.dict_page(rand_dict(rng, pg.INT32, nd), nd)
            for n in (7, 300, 2048):
                body, have = bytearray(), 0
                while have + 5 < n:
                    c = int(rng.integers(1, 4))
                    body += run(c, rng.integers(0, nd)); have += c
                # the run the count ends in, then runs nobody reads
                body += run(n - have + 2, rng.integers(0, nd) if last_ok else nd + 3)
                body += run(3, nd + 7) + run(1, 0xFFFFFFFF >> (32 - 8 * nb))
                cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, n, bytes([w]) + bytes(body))
            chunk, pages = cb.finish()
            g, o = check_same(dec, oracle, chunk, pages, pg.INT32)
            assert (o.status == 0) == last_ok


@pytest.mark.parametrize("direct", ["1", "0"])
@pytest.mark.parametrize("ptype,w", [(pg.INT32, 32), (pg.INT64, 32), (pg.INT32, 9), (pg.DOUBLE, 1), (pg.BYTE_ARRAY, 32), (pg.INT96, 32),
                                     (pg.BYTE_ARRAY, 7)])
def test_rle_only_pages_expanded_straight_from_the_page(dec, oracle, ptype, w, direct):
    # index streams of one-byte-header RLE runs only (WriteRLE, encodingwrite.go:149-214) skip the run table when the
    # entries are 4 or 8 bytes wide; with nulls; a count that ends inside a run; and the pages that must stay on the
    # general path: a trailing byte, an empty run, a two-byte header, a bit-packed run, too few values; an index
    # beyond the dictionary
    dec.set_option(1, int(direct))           # PQD_OPT_DIRECT_RLE
    rng = np.random.default_rng(8100 + w + ptype)
    nd = min(1 << w, 300)
    nb = (w + 7) // 8

    def runs(n_values, max_run=3, hi=None):
        out, vals = bytearray(), []
        while len(vals) < n_values:
            c, v = int(rng.integers(1, max_run + 1)), int(rng.integers(0, hi or nd))
            if hi and not vals:
                v = hi - 1
            out += bytes([c << 1]) + v.to_bytes(4, "little")[:nb]
            vals += [v] * c
        return bytes(out), len(vals)

    def builder():
        cb = pg.ChunkBuilder(gap=int(rng.integers(0, 5)), lead=int(rng.integers(0, 3)))
        if ptype == pg.BYTE_ARRAY:
            cb.dict_page(pg.plain_byte_array(rand_strings(rng, nd, 0, 20)), nd)
        else:
            cb.dict_page(rand_dict(rng, ptype, nd), nd)
        return cb

    def page(cb, n, levels, tail=b"", short=0, max_run=3, hi=None):
        max_def, max_rep, style = levels
        dl = rng.integers(0, max_def + 1, n) if max_def else np.zeros(n, np.int64)
        rl = rng.integers(0, max_rep + 1, n) if max_rep else np.zeros(n, np.int64)
        rl[0] = 0
        nn = int((dl == max_def).sum())
        body, have = runs(max(nn - short, 1) if short else nn + int(rng.integers(0, 3)), max_run, hi)
        cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, n, bytes([w]) + body + tail,
                        pg.level_block_v1(rl, max_rep, rng, style) if max_rep else b"",
                        pg.level_block_v1(dl, max_def, rng, style) if max_def else b"")

    # (level streams written as RLE runs only, layout/page.go:234-238, take the same route when every run is short)
    for max_def in ((0, 0, "mixed"), (2, 0, "rle"), (2, 1, "rle"), (3, 1, "mixed")):
        cb = builder()
        for k, n in enumerate([1, 2, 300, 2048, 2048, 1500, 7000]):
            page(cb, n, max_def, max_run=63 if k == 6 else 3)
        chunk, pages = cb.finish()
        g, o = check_same(dec, oracle, chunk, pages, ptype, 0, max_def[0], max_def[1])
        assert o.status == 0
        # pages that stay on the general path, each behind two good ones: a trailing byte, an empty run, a two-byte
        # header, a bit-packed run, too few values; and an index beyond the dictionary
        for tail, short, hi in ((b"\x00", 0, None), (b"\x00" + bytes(nb), 0, None), (pg.varint(100 << 1) + bytes(nb), 0, None),
                                (pg.varint(3) + pg.bitpack([0] * 8, w), 0, None), (b"", 5, None), (b"", 0, nd + 1)):
            cb = builder()
            page(cb, 500, max_def)
            page(cb, 40, max_def)
            page(cb, 600, max_def, tail, short, 3, hi)
            chunk, pages = cb.finish()
            g, o = check_same(dec, oracle, chunk, pages, ptype, 0, max_def[0], max_def[1])
            if hi and nd < (1 << (8 * nb)):
                assert o.status == oracle.PANIC and o.bad_page == 3


def _runs_body(rng, w, nd, n_target, mix):
    """Index stream of runs drawn from `mix` (kind -> weight) until it holds n_target values; returns (bytes, values)."""
    nb = (w + 7) // 8
    kinds = list(mix)
    p = np.array([mix[k] for k in kinds], float)
    p /= p.sum()
    out, vals = bytearray(), []

    def rle(c):
        v = int(rng.integers(0, nd))
        out.extend(pg.varint(c << 1) + v.to_bytes(4, "little")[:nb])
        vals.extend([v] * c)

    def bp(g):
        x = rng.integers(0, nd, 8 * g)
        out.extend(pg.varint((g << 1) | 1) + pg.bitpack(x, w))
        vals.extend(x.tolist())

    while len(vals) < n_target:
        k = kinds[int(rng.choice(len(kinds), p=p))]
        if k == "bp":            # one-byte header, up to 63 groups
            bp(int(rng.integers(1, 64)))
        elif k == "bp2":         # two-byte header: a payload of many KiB, crossing several windows of the walk
            bp(int(rng.integers(64, min(8192, 64 + 40000 // max(w, 1)))))
        elif k == "rle":
            rle(int(rng.integers(1, 64)))
        elif k == "rle2":        # two-byte header
            rle(int(rng.integers(64, 8192)))
        elif k == "short":       # a few values per run: more runs in a window than one step of the walk takes
            if rng.integers(0, 2):
                rle(int(rng.integers(1, 4)))
            else:
                bp(1)
        elif k == "rle3":        # three-byte header
            rle(int(rng.integers(8192, 9000)))
        elif k == "empty":
            out.extend(b"\x00" + bytes(nb))
        elif k == "emptybp":
            out.extend(b"\x01")
    return bytes(out), vals


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("ptype,w,nd", [(pg.INT32, 1, 2), (pg.INT32, 3, 7), (pg.INT32, 8, 256), (pg.INT32, 12, 4096), (pg.INT64, 12, 3000),
                                        (pg.DOUBLE, 5, 32), (pg.INT32, 16, 65536), (pg.INT32, 20, 65536), (pg.INT64, 15, 30000),
                                        (pg.INT32, 32, 5000), (pg.FLOAT, 17, 20000)])
def test_large_pages_walked_and_expanded_in_one_pass(dec, oracle, ptype, w, nd, fused):
    # Index streams of pages large enough for the one-pass kernel (a workgroup per page: a wavefront follows the run
    # headers through a ring of page bytes in LDS, the others expand out of it): pages of many windows, runs with one-
    # and two-byte headers, bit-packed runs that span several windows, stretches of more runs per window than one step
    # takes, a count that ends inside the last run, two dictionaries in one chunk, a nullable column; and the pages
    # the kernel must leave to the general path - a three-byte header, empty runs, bytes behind the last run, too few
    # values - each behind good pages; an index beyond the dictionary.  Same results with the kernel on and off.
    dec.set_option(7, int(fused))            # PQD_OPT_FUSED
    rng = np.random.default_rng(7700 + w + ptype)
    common = {"bp": 60, "rle": 20, "rle2": 5, "bp2": 3}
    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 9)), lead=int(rng.integers(0, 5)))
    cb.dict_page(rand_dict(rng, ptype, nd), nd)
    nvals = (120000, 40000, 9000, 200) if w < 16 else (60000, 20000, 300)
    for k, n_target in enumerate(nvals):
        body, vals = _runs_body(rng, w, nd, n_target, common if k != 1 else {"bp": 30, "rle": 10, "short": 60})
        n = len(vals) - int(rng.integers(0, min(8, len(vals))))      # the count may end inside the last run
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, bytes([w]) + body)
    # a second, smaller dictionary for the pages behind it
    nd2 = max(1, nd // 3)
    cb.dict_page(rand_dict(rng, ptype, nd2), nd2)
    body, vals = _runs_body(rng, w, nd2, 30000, common)
    cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, len(vals), bytes([w]) + body)
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, ptype)
    assert o.status == 0
    # with the kernel on, every index stream of this chunk was finished by it (none fell back to the general path)
    dec.decode(chunk, pages, ptype)
    assert dec.direct_streams() == (len(nvals) + 1 if fused == "1" else 0)
    # nullable and nested: the values of a page are fewer than its slots and land behind those of the pages before it
    cb = pg.ChunkBuilder(gap=2)
    cb.dict_page(rand_dict(rng, ptype, nd), nd)
    for n in (50000, 30000):
        dl, rl = _levels(rng, n, 2, 1)
        nn = int((dl == 2).sum())
        body, vals = _runs_body(rng, w, nd, nn, common)
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, bytes([w]) + body, pg.level_block_v1(rl, 1, rng, "mixed"), pg.level_block_v1(dl, 2, rng, "mixed"))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, ptype, 0, 2, 1)
    assert o.status == 0
    # pages for the general path, each behind a good page
    nb = (w + 7) // 8
    for what in ("rle3", "empty", "emptybp", "tail", "short", "index"):
        cb = pg.ChunkBuilder(gap=1)
        cb.dict_page(rand_dict(rng, ptype, nd), nd)
        body, vals = _runs_body(rng, w, nd, 30000, common)
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, len(vals), bytes([w]) + body)
        body, vals = _runs_body(rng, w, nd, 20000, common)
        if what in ("rle3", "empty", "emptybp"):
            if what == "rle3":
                odd, odd_vals = _runs_body(rng, w, nd, 1, {what: 1})
            else:
                odd, odd_vals = (b"\x00" + bytes(nb) if what == "empty" else b"\x01"), []
            more, more_vals = _runs_body(rng, w, nd, 20000, common)
            body, vals = body + odd + more, vals + odd_vals + more_vals
        n = len(vals)
        if what == "tail":
            body += b"\x02" + bytes(nb)          # a run behind the one that completes the count
        elif what == "short":
            n += 5                               # the stream holds too few values
        elif what == "index":
            if nd >= (1 << min(w, 31)):
                continue
            v = (1 << w) - 1
            body += pg.varint(9 << 1) + v.to_bytes(4, "little")[:nb]   # dict.Values[index] out of range
            n += 9
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, bytes([w]) + body)
        chunk, pages = cb.finish()
        g, o = check_same(dec, oracle, chunk, pages, ptype)
        if what in ("short", "index"):
            assert o.status != 0 and o.bad_page == 2
        # the good page is finished by the kernel, the odd one is left to the general path (an index beyond the dictionary
        # is no concern of the walk: that page is finished too, and fails as the reference's gather does)
        dec.decode(chunk, pages, ptype)
        assert dec.direct_streams() == ((2 if what == "index" else 1) if fused == "1" else 0), what


@pytest.mark.parametrize("variant", ["wave", "wide"])
@pytest.mark.parametrize("w", [1, 2, 7, 8, 9, 16, 20, 32])
def test_dense_rle_stretches_taken_a_stride_at_a_time(dec, oracle, w, variant):
    # long stretches of one-byte-header RLE runs (what WriteRLE emits, encodingwrite.go:149-172) broken up by runs of
    # 64 or more values (two-byte header), bit-packed runs and empty runs; pages whose count ends inside a stretch,
    # stretches that cross tile and window boundaries; levels written the same way
    dec.set_option(0, 1 if variant == "wide" else 0)   # PQD_OPT_PRESCAN
    rng = np.random.default_rng(900 + w)
    nd = min(1 << w, 5000)

    def dense_runs(n_target):
        out, vals = bytearray(), []
        nb = (w + 7) // 8
        while len(vals) < n_target:
            kind = rng.integers(0, 200)
            if kind == 0:      # two-byte header
                c, v = int(rng.integers(64, 400)), int(rng.integers(0, nd))
                out += pg.varint(c << 1) + v.to_bytes(4, "little")[:nb]
                vals += [v] * c
            elif kind == 1:    # bit-packed groups
                g = int(rng.integers(1, 4))
                x = rng.integers(0, nd, 8 * g)
                out += pg.varint((g << 1) | 1) + pg.bitpack(x, w)
                vals += [int(t) for t in x]
            elif kind == 2:    # empty run
                out += b"\x00" + bytes(nb)
            else:
                c, v = int(rng.integers(1, 64 if kind % 3 else 3)), int(rng.integers(0, nd))
                out += bytes([c << 1]) + v.to_bytes(4, "little")[:nb]
                vals += [v] * c
        return bytes(out), vals

    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 7)), lead=int(rng.integers(0, 3)))
    cb.dict_page(rand_dict(rng, pg.INT32, nd), nd)
    for n_target in (50, 4096, 9000, 40000):
        body, vals = dense_runs(n_target)
        n = len(vals) - int(rng.integers(0, min(40, len(vals))))   # the page's count may end inside a run or a stretch
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, bytes([w]) + body)
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.INT32)
    # levels as the reference writes them: RLE runs only
    cb = pg.ChunkBuilder(gap=1)
    for n in (300, 20000):
        dl, rl = _levels(rng, n, 3, 2)
        nn = int((dl == 3).sum())
        cb.data_page_v1(pg.ENC_PLAIN, n, rng.integers(-2**31, 2**31, nn, dtype=np.int64).astype(np.int32).tobytes(),
                        pg.level_block_v1(rl, 2, rng, "rle"), pg.level_block_v1(dl, 3, rng, "rle"))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.INT32, 0, 3, 2)


@pytest.mark.parametrize("jump", ["1", "0"])
@pytest.mark.parametrize("w", [1, 2, 3, 5, 8, 9, 16, 20, 32])
def test_short_runs_resolved_a_span_at_a_time(dec, oracle, w, jump):
    # Arrow-style streams: a few values per run, bit-packed groups and RLE runs mixed, so that the headers follow no
    # fixed stride; now and then a run with a two-byte header, one long enough to leave the span or the window, an
    # empty run of either kind and a three-byte header (none of which the successor table covers); pages whose
    # count ends inside a span; the same as definition / repetition levels
    dec.set_option(0, 0)                     # PQD_OPT_PRESCAN: a wavefront per stream
    dec.set_option(2, int(jump))             # PQD_OPT_JUMP
    rng = np.random.default_rng(4400 + w)
    nd = min(1 << w, 5000)
    nb = (w + 7) // 8

    def short_runs(n_target, odd_every):
        out, vals = bytearray(), []
        while len(vals) < n_target:
            kind = int(rng.integers(0, odd_every))
            if kind == 0:      # two-byte header, RLE
                c, v = int(rng.integers(64, 400)), int(rng.integers(0, nd))
                out += pg.varint(c << 1) + v.to_bytes(4, "little")[:nb]
                vals += [v] * c
            elif kind == 1:    # two-byte header, bit-packed: up to a few KiB of payload
                g = int(rng.integers(64, 64 + 2048 // w))
                x = rng.integers(0, nd, 8 * g)
                out += pg.varint((g << 1) | 1) + pg.bitpack(x, w)
                vals += x.tolist()
            elif kind == 2:    # empty RLE run
                out += b"\x00" + bytes(nb)
            elif kind == 3:    # empty bit-packed run
                out += b"\x01"
            elif kind == 4:    # three-byte header
                c, v = int(rng.integers(8192, 9000)), int(rng.integers(0, nd))
                out += pg.varint(c << 1) + v.to_bytes(4, "little")[:nb]
                vals += [v] * c
            elif kind % 2:
                g = int(rng.integers(1, 4))
                x = rng.integers(0, nd, 8 * g)
                out += pg.varint((g << 1) | 1) + pg.bitpack(x, w)
                vals += x.tolist()
            else:
                c, v = int(rng.integers(1, 20)), int(rng.integers(0, nd))
                out += bytes([c << 1]) + v.to_bytes(4, "little")[:nb]
                vals += [v] * c
        return bytes(out), vals

    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 7)), lead=int(rng.integers(0, 3)))
    cb.dict_page(rand_dict(rng, pg.INT32, nd), nd)
    for n_target, odd_every in ((50, 40), (5000, 10**9), (30000, 2000), (120000, 300), (20000, 30)):
        body, vals = short_runs(n_target, odd_every)
        n = len(vals) - int(rng.integers(0, min(40, len(vals))))
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, bytes([w]) + body)
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.INT32)
    assert o.status == 0
    if w <= 3:
        max_def = (1 << w) - 1
        block = lambda body: len(body).to_bytes(4, "little") + body   # v1 level section (layout/page.go:234-238)
        cb = pg.ChunkBuilder(gap=1)
        for n in (300, 20000, 150000):
            dl = rng.integers(0, max_def + 1, n)
            rl = rng.integers(0, 2, n)
            rl[0] = 0
            for r in rng.integers(0, n, n // 40):
                k = int(rng.integers(8, 60))
                dl[r:r + k] = dl[r]
                rl[r + 1:r + k] = 1
            nn = int((dl == max_def).sum())
            cb.data_page_v1(pg.ENC_PLAIN, n, rng.integers(-2**31, 2**31, nn, dtype=np.int64).astype(np.int32).tobytes(),
                            block(pg.hybrid_encode(rl, 1, rng, "arrow", max_groups=2)),
                            block(pg.hybrid_encode(dl, w, rng, "arrow", max_groups=3)))
        chunk, pages = cb.finish()
        g, o = check_same(dec, oracle, chunk, pages, pg.INT32, 0, max_def, 1)
        assert o.status == 0


@pytest.mark.parametrize("ptype,tl", [(pg.INT32, 0), (pg.INT64, 0), (pg.FLOAT, 0), (pg.DOUBLE, 0), (pg.INT96, 0),
                                      (pg.FIXED_LEN_BYTE_ARRAY, 7), (pg.BOOLEAN, 0)])
def test_plain_fixed(dec, oracle, ptype, tl):
    rng = np.random.default_rng(20 + ptype)
    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 7)), lead=3)
    for n in (1, 15, 16, 17, 4099, 100003):
        if ptype == pg.BOOLEAN:
            body = oracle.write_plain_boolean(rng.integers(0, 2, n, dtype=np.uint8))
        else:
            body = rand_dict(rng, ptype, n, tl)
        cb.data_page_v1(pg.ENC_PLAIN, n, body)
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, ptype, tl)


@pytest.mark.parametrize("dtype,ptype", [(np.int32, pg.INT32), (np.int64, pg.INT64)])
def test_delta_binary_packed(dec, oracle, dtype, ptype):
    rng = np.random.default_rng(30 + ptype)
    info = np.iinfo(dtype)
    cb = pg.ChunkBuilder(gap=5, lead=1)
    for n in (1, 2, 33, 128, 129, 130, 257, 5000, 40001):
        for spread, trim in ((0, False), (3, True), (1000, False), (1 << 20, True), (None, False)):
            if spread is None:
                vals = rng.integers(info.min, info.max, n, dtype=dtype, endpoint=True)
            else:
                vals = (np.cumsum(rng.integers(-spread, spread + 1, n)) + 1_700_000_000).astype(dtype)
            enc = pg.delta_encode(vals, dtype, trim=trim) if n < 6000 else pg.delta_encode_fast(vals, dtype)
            cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, n, enc)
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, ptype)


@pytest.mark.parametrize("blocks", [1, 2, 0])
@pytest.mark.parametrize("n_pages,page_values", [(3, 70001), (600, 1500), (2100, 700)])
def test_delta_block_parallel_and_wave_per_page_agree_with_the_oracle(dec, oracle, n_pages, page_values, blocks):
    # standard 128 / 4 streams go through a table of block starts and are decoded by one, four or sixteen wavefronts
    # per page, depending on how many pages there are; widths below and above 24 bits, width 0, 63-bit deltas that
    # wrap around, INT32; last blocks that are partial; and the same pages with the table switched off
    from gen import pqgen
    dec.set_option(6, blocks)                # PQD_OPT_DELTA_BLOCKS
    for w, esz, ptype in ((12, 8, pg.INT64), (0, 8, pg.INT64), (24, 8, pg.INT64), (25, 8, pg.INT64), (40, 8, pg.INT64), (63, 8, pg.INT64),
                          (12, 4, pg.INT32), (31, 4, pg.INT32)):
        if n_pages > 100 and w in (0, 24, 40):
            continue
        chunk, pages = pqgen.delta_chunk(0xD0 + w, n_pages * page_values - 5, w, page_values, esz, 1_700_000_000_000_000 if esz == 8 else 7)
        g = dec.decode(chunk, pages, ptype)
        o = oracle.read_chunk(chunk, pages, ptype)
        assert g.status == o.status == 0, (w, esz, g.status, g.bad_page)
        assert np.array_equal(g.values, o.values), (w, esz)
    # a stream cut inside a block, a block header cut short: the general path's cases
    chunk, pages = pqgen.delta_chunk(0xD9, 3 * 5000, 12, 5000)
    for cut in (7, 300, 4001):
        pp = pages.copy()
        pp["length"][1] = cut
        check_same(dec, oracle, chunk, pp, pg.INT64)


def test_delta_page_of_constant_blocks_with_a_few_of_the_widest_kind(dec, oracle):
    # Pages that spend next to nothing per value (blocks of width 0: five bytes each) get the smallest window over a
    # unit's packed bytes; the unit that holds blocks of the widest kind (ten-byte minimum delta, 64-bit deltas: 1038
    # bytes each) is then decoded through a window that moves over its blocks - two such blocks must fit it.
    rng = np.random.default_rng(20782028245)
    cb = pg.ChunkBuilder(gap=7, lead=3)
    info = np.iinfo(np.int64)
    for n, at in ((58183, 74), (120000, 3), (30000, 130)):
        d = np.full(n, 3, np.int64)
        d[128 * at + 1:128 * at + 385] = rng.integers(info.min, info.max, 384)     # three such blocks next to each other
        with np.errstate(over="ignore"):
            vals = np.cumsum(d.astype(np.uint64)).view(np.int64)
        cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, n, pg.delta_encode_fast(vals, np.int64))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.INT64)
    assert o.status == 0


@pytest.mark.parametrize("blocks", [1, 2])
@pytest.mark.parametrize("dtype,ptype", [(np.int64, pg.INT64), (np.int32, pg.INT32)])
def test_delta_blocks_of_one_length_but_for_a_few(dec, oracle, dtype, ptype, blocks):
    # Pages whose blocks all have one length skip the walk of their block headers: block k is taken to start k lengths
    # after the first, 64 blocks spread over the page are tried, and the wavefronts that decode the blocks check the rest.
    # Here every block is 198 bytes long, except (a) none, (b) two neighbours that are 4 bytes longer and 4 bytes shorter
    # - every block behind them starts where it would anyway, so the 64 trials see nothing -, (c) the same pair inside the
    # page's first 64 blocks and inside its last ones, (d) one longer block (everything behind it moves), (e) a last
    # block of wider deltas than the rest.  encodingread.go:329-383.
    dec.set_option(6, blocks)                # PQD_OPT_DELTA_BLOCKS
    rng = np.random.default_rng(4100 + ptype)
    cb = pg.ChunkBuilder(gap=3, lead=2)

    def page(n_blocks, pairs=(), longer=(), wide_tail=False, tail=17):
        d = rng.integers(1000, 1000 + 4096, n_blocks * 128 + tail).astype(np.int64)
        for b in range(n_blocks):
            d[b * 128] = 1000                                   # every block's minimum
            d[b * 128 + 1:b * 128 + 128:32] = 1000 + 4095       # twelve bits in every miniblock
        for a in pairs:
            d[a * 128 + 5] = 1000 + 5000                        # thirteen bits in block a's first miniblock
            d[(a + 1) * 128:(a + 1) * 128 + 32] = rng.integers(1000, 1000 + 2048, 32)
            d[(a + 1) * 128] = 1000
            d[(a + 1) * 128 + 1] = 1000 + 2047                  # eleven in the next block's
        for a in longer:
            d[a * 128 + 70] = 1000 + 9000
        if wide_tail:
            d[n_blocks * 128:] = rng.integers(1000, 1000 + (1 << 20), tail)
        vals = (7 + np.concatenate([[0], np.cumsum(d)])).astype(dtype)
        cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, len(vals), pg.delta_encode_fast(vals, dtype))

    page(200)
    page(200, pairs=(76,))
    page(400, pairs=(10, 395))
    page(300, longer=(150,))
    page(260, wide_tail=True)
    page(64, pairs=(30,), tail=1)
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, ptype)
    assert o.status == 0


def test_delta_reference_writer_and_other_geometries(dec, oracle):
    rng = np.random.default_rng(33)
    cb = pg.ChunkBuilder()
    # restated parquet-go writer, incl. the wrap-around inputs of encodingread_test.go:153
    cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, 5, oracle.write_delta_int64(
        [9223372036854775807, -9223372036854775808, -15654523568543623, 4354365463543632, 0]))
    for block, n_mb in ((256, 8), (128, 2), (64, 2), (512, 4), (128, 16)):
        vals = np.cumsum(rng.integers(-5000, 5000, 3000)).astype(np.int64)
        cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, len(vals), pg.delta_encode(vals, np.int64, block, n_mb))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.INT64)


@pytest.mark.parametrize("blocks", [-1, 0])
@pytest.mark.parametrize("where", ["first value", "minimum delta"])
def test_delta_varint_of_more_than_ten_bytes_is_refused(dec, oracle, where, blocks):
    # ReadUnsignedVarInt (encodingread.go:127-143) reads on as long as the continuation bit is set and drops what is
    # shifted beyond 64 bits: a varint padded with 0x80 bytes decodes to the same number there.  The library does not
    # follow varints beyond ten bytes: it refuses the page (PQD_ERR_REFUSED, -13) and must not call it malformed -
    # that would claim a panic the reference does not have.  The pages in front of it decode as usual.
    dec.set_option(6, blocks)                # PQD_OPT_DELTA_BLOCKS
    rng = np.random.default_rng(77)

    def varint_end(b, i):
        while b[i] & 0x80:
            i += 1
        return i + 1

    vals = np.cumsum(rng.integers(-300, 300, 1000)).astype(np.int32)
    good = pg.delta_encode(vals, np.int32)
    i = 0
    for _ in range(3 if where == "first value" else 4):
        i = varint_end(good, i)
    j = varint_end(good, i)
    padded = bytearray(good[i:j])
    padded[-1] |= 0x80
    padded += b"\x80" * (12 - len(padded) - 1) + b"\x00"
    bad = good[:i] + bytes(padded) + good[j:]
    cb = pg.ChunkBuilder()
    cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, len(vals), good)
    cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, len(vals), bad)
    chunk, pages = cb.finish()
    o = oracle.read_chunk(chunk, pages, pg.INT32, 0, 0, 0)
    assert o.status == 0 and np.array_equal(np.ascontiguousarray(o.values).view(np.int32).ravel(), np.concatenate([vals, vals]))
    g = dec.decode(chunk, pages, pg.INT32, 0, 0, 0)
    assert g.status == -13 and g.bad_page == 1


def _levels(rng, n, max_def, max_rep):
    dl = rng.integers(0, max_def + 1, n) if max_def else np.zeros(n, np.int64)
    # long stretches of equal levels, as real columns have
    for r in rng.integers(0, n, max(1, n // 200)):
        dl[r:r + int(rng.integers(5, 300))] = max_def
    rl = np.zeros(n, np.int64)
    if max_rep:
        rl = rng.integers(0, max_rep + 1, n)
        rl[0] = 0
    return dl, rl


@pytest.mark.parametrize("max_def,max_rep", [(1, 0), (3, 1), (2, 2), (7, 0)])
@pytest.mark.parametrize("enc", ["plain", "dict", "delta"])
def test_levels_and_nulls(dec, oracle, max_def, max_rep, enc):
    rng = np.random.default_rng(40 + max_def * 10 + max_rep)
    cb = pg.ChunkBuilder(gap=2)
    nd = 300
    dict_vals = rng.integers(-2**62, 2**62, nd, dtype=np.int64)
    if enc == "dict":
        cb.dict_page(dict_vals.tobytes(), nd)
    for n, v2 in ((1, False), (100, True), (4096, False), (9000, True), (50000, False)):
        dl, rl = _levels(rng, n, max_def, max_rep)
        nn = int((dl == max_def).sum())
        if enc == "plain":
            body, e = rng.integers(-2**62, 2**62, nn, dtype=np.int64).tobytes(), pg.ENC_PLAIN
        elif enc == "dict":
            body, e = pg.dict_indices_body(rng.integers(0, nd, nn), 9, rng), pg.ENC_RLE_DICTIONARY
        else:
            body = pg.delta_encode_fast(np.cumsum(rng.integers(0, 1000, nn)).astype(np.int64)) if nn else b""
            e = pg.ENC_DELTA_BINARY_PACKED
        if v2:
            cb.data_page_v2(e, n, body, pg.hybrid_encode(rl, int(max_rep).bit_length(), rng) if max_rep else b"",
                            pg.hybrid_encode(dl, int(max_def).bit_length(), rng))
        else:
            cb.data_page_v1(e, n, body, pg.level_block_v1(rl, max_rep, rng) if max_rep else b"",
                            pg.level_block_v1(dl, max_def, rng, "arrow"))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.INT64, 0, max_def, max_rep)


@pytest.mark.parametrize("tables", [1, 0, -1])
@pytest.mark.parametrize("shape", ["nested-mixed", "plain-nullable", "flat-dict", "rep-without-def"])
def test_stream_tables_by_kernel_and_by_host(dec, oracle, shape, tables):
    # The per-stream tables of a call (run-table ranges, tile prefixes, tile -> stream maps) come from a kernel for
    # chunks of thousands of pages - what the reference's writer produces, 8 KiB a page (writer/writer.go:63-65) - and
    # from the host otherwise: the same chunk both ways (and as the library chooses), with several dictionaries, pages
    # of no values, pages of more than one tile, v1 and v2 pages and PLAIN / Encoding_RLE pages between the indexed ones.
    dec.set_option(9, tables)                # PQD_OPT_DEVICE_TABLES
    rng = np.random.default_rng(7700 + len(shape))
    max_def, max_rep = {"nested-mixed": (3, 1), "plain-nullable": (1, 0), "flat-dict": (0, 0), "rep-without-def": (0, 2)}[shape]
    cb = pg.ChunkBuilder(gap=int(rng.integers(0, 5)))
    n_pages = 5000
    sizes = rng.integers(1, 60, n_pages)
    sizes[rng.integers(0, n_pages, 40)] = 0
    sizes[rng.integers(0, n_pages, 6)] = rng.integers(4097, 15000, 6)
    nd = 0
    for k, n in enumerate(int(x) for x in sizes):
        if shape != "plain-nullable" and (k % 1300 == 0 or (k == 2 and shape == "nested-mixed")):
            nd = int(rng.integers(1, 400))
            cb.dict_page(rng.integers(-2**62, 2**62, nd, dtype=np.int64).tobytes(), nd)
        dl, rl = _levels(rng, n, max_def, max_rep) if n else (np.zeros(0, np.int64), np.zeros(0, np.int64))
        nn = int((dl == max_def).sum()) if max_def else n
        kind = "plain" if shape == "plain-nullable" else ("plain" if k % 17 == 5 else "delta" if k % 23 == 7 else "dict")
        if kind == "plain":
            body, e = rng.integers(-2**62, 2**62, nn, dtype=np.int64).tobytes(), pg.ENC_PLAIN
        elif kind == "delta":
            body, e = (pg.delta_encode_fast(np.cumsum(rng.integers(0, 1000, nn)).astype(np.int64)) if nn else b""), pg.ENC_DELTA_BINARY_PACKED
        else:
            body, e = pg.dict_indices_body(rng.integers(0, nd, nn), max(1, int(nd - 1).bit_length()), rng), pg.ENC_RLE_DICTIONARY
        if nn == 0:
            body = b""
        if k % 3 == 1 and n:
            cb.data_page_v2(e, n, body, pg.hybrid_encode(rl, int(max_rep).bit_length(), rng) if max_rep else b"",
                            pg.hybrid_encode(dl, int(max_def).bit_length(), rng) if max_def else b"")
        else:
            cb.data_page_v1(e, n, body, pg.level_block_v1(rl, max_rep, rng) if max_rep else b"",
                            pg.level_block_v1(dl, max_def, rng) if max_def else b"")
    chunk, pages = cb.finish()
    # (twice: from the second call on the handle's page-table block is large enough for the page entries to go to the
    #  device slice by slice while the host still walks the rest of the pages)
    for _ in range(2):
        g, o = check_same(dec, oracle, chunk, pages, pg.INT64, 0, max_def, max_rep)
        assert o.status == 0


@pytest.mark.parametrize("how", ["bits", "levels", "levels-direct", "copy"])
@pytest.mark.parametrize("ptype", [pg.INT32, pg.INT64, pg.FLOAT, pg.DOUBLE])
def test_nullable_plain_scattered_from_the_pages(dec, oracle, ptype, how):
    # nullable column, PLAIN pages only: the scatter reads the value sections of the pages (no dense copy): many small
    # pages per 4096 slots, pages that are all null, pages without slots, a page boundary inside a thread's
    # slots; then the same chunk with one page's value section a few bytes short (binaryread.go:12-18: ReadFull fails).
    # Which slots hold a value comes from the validity bits the level expansion leaves per tile ("bits"), or from the
    # definition levels read again ("levels"; "levels-direct": with the RLE-only level expansion tried first, which
    # leaves no bits), or the values go through a dense copy first ("copy").
    if how == "bits":
        dec.set_option(1, 0)                 # PQD_OPT_DIRECT_RLE off: every level stream through the tile expansion
    elif how == "levels":
        dec.set_option(1, 0)
        dec.set_option(3, 2)                 # PQD_OPT_PLAIN_SCATTER: read the levels again
    elif how == "levels-direct":
        dec.set_option(1, 1)
    else:
        dec.set_option(3, 0)                 # PQD_OPT_PLAIN_SCATTER off
    rng = np.random.default_rng(6100 + ptype)
    esz = np.dtype(PHYS_DT[ptype]).itemsize
    for max_def, max_rep in ((1, 0), (3, 1)):
        cb = pg.ChunkBuilder(gap=int(rng.integers(0, 7)), lead=int(rng.integers(0, 3)))
        sizes = [1, 2, 3, 0, 5, 4090, 0, 0, 7, 1021, 3000, 11, 13, 20000, 2, 4096, 1, 9001] + [int(x) for x in rng.integers(1, 40, 60)]
        for k, n in enumerate(sizes):
            dl, rl = _levels(rng, n, max_def, max_rep) if n else (np.zeros(0, np.int64), np.zeros(0, np.int64))
            if k % 5 == 4:
                dl[:] = 0           # nothing but nulls
            if k % 7 == 6:
                dl[:] = max_def     # no nulls
            nn = int((dl == max_def).sum())
            body = rng.integers(0, 256, nn * esz, dtype=np.uint8).tobytes()
            if k % 2:
                cb.data_page_v2(pg.ENC_PLAIN, n, body, pg.hybrid_encode(rl, 1, rng) if max_rep else b"",
                                pg.hybrid_encode(dl, int(max_def).bit_length(), rng))
            else:
                cb.data_page_v1(pg.ENC_PLAIN, n, body, pg.level_block_v1(rl, max_rep, rng) if max_rep else b"",
                                pg.level_block_v1(dl, max_def, rng))
        chunk, pages = cb.finish()
        g, o = check_same(dec, oracle, chunk, pages, ptype, 0, max_def, max_rep)
        assert o.status == 0 and o.n_nulls > 0
        short = pages.copy()
        short["length"][13] -= 3
        g, o = check_same(dec, oracle, chunk, short, ptype, 0, max_def, max_rep)
        assert o.status != 0 and o.bad_page == 13


@pytest.mark.parametrize("ptype,dt", [(pg.FLOAT, np.float32), (pg.DOUBLE, np.float64)])
def test_byte_stream_split(dec, oracle, ptype, dt):
    rng = np.random.default_rng(61 + ptype)
    cb = pg.ChunkBuilder(gap=3, lead=1)
    esz = np.dtype(dt).itemsize
    for n, nullable in ((1, False), (3, False), (4, False), (5, False), (4097, True), (70001, False), (16384, True)):
        dl = (rng.random(n) < 0.8).astype(np.int64) if nullable else np.ones(n, np.int64)
        nn = int(dl.sum())
        vals = rng.standard_normal(nn).astype(dt)
        body = vals.view(np.uint8).reshape(nn, esz).T.tobytes()   # plane k = byte k of every value
        cb.data_page_v1(9, n, body, b"", pg.level_block_v1(dl, 1, rng))
    # mixed with a PLAIN page, and a short section
    cb.data_page_v1(pg.ENC_PLAIN, 7, rng.standard_normal(7).astype(dt).tobytes(), b"", pg.level_block_v1(np.ones(7, np.int64), 1))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, ptype, 0, 1, 0)
    assert o.status == 0
    cb.data_page_v1(9, 10, b"\0" * (10 * esz - 1), b"", pg.level_block_v1(np.ones(10, np.int64), 1))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, ptype, 0, 1, 0)
    assert o.status == oracle.UNEXPECTED_EOF
    # not defined for integer columns (layout/page.go:713)
    cb = pg.ChunkBuilder()
    cb.data_page_v1(9, 4, b"\0" * 16)
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.INT32)
    assert o.status == oracle.UNSUPPORTED


@pytest.mark.parametrize("ptype", [pg.INT64, pg.BYTE_ARRAY])
def test_levels_only_skip_path(dec, oracle, ptype):
    # what SkipRows needs (layout/page.go:433-559): levels, nulls and rows per page, values untouched
    rng = np.random.default_rng(77 + ptype)
    cb = pg.ChunkBuilder(gap=4)
    for n, v2 in ((1, False), (3000, True), (70000, False), (9, True)):
        dl, rl = _levels(rng, n, 3, 1)
        nn = int((dl == 3).sum())
        if ptype == pg.INT64:
            body = rng.integers(-2**62, 2**62, nn, dtype=np.int64).tobytes()
        else:
            body = pg.plain_byte_array([b"x" * int(k) for k in rng.integers(0, 9, nn)])
        if v2:
            cb.data_page_v2(pg.ENC_PLAIN, n, body, pg.hybrid_encode(rl, 1, rng), pg.hybrid_encode(dl, 2, rng))
        else:
            cb.data_page_v1(pg.ENC_PLAIN, n, body, pg.level_block_v1(rl, 1, rng), pg.level_block_v1(dl, 3, rng, "arrow"))
    chunk, pages = cb.finish()
    o = oracle.read_chunk(chunk, pages, ptype, 0, 3, 1)
    g = dec.decode_levels(chunk, pages, ptype, 0, 3, 1)
    assert g.status == o.status == 0 and g.kind == "levels"
    assert (g.n_slots, g.n_rows, g.n_nulls) == (o.n_slots, o.n_rows, o.n_nulls)
    assert np.array_equal(g.def_levels, o.def_levels) and np.array_equal(g.rep_levels, o.rep_levels)
    # the handle still decodes values afterwards
    check_same(dec, oracle, chunk, pages, ptype, 0, 3, 1)
    # a broken level section is reported the same way
    cb.data_page_v1(pg.ENC_PLAIN, 10, b"", b"\x02\x00\x00\x00\x03\x00", b"\x05\x00")
    chunk, pages = cb.finish()
    o = oracle.read_chunk(chunk, pages, ptype, 0, 3, 1)
    g = dec.decode_levels(chunk, pages, ptype, 0, 3, 1)
    assert g.status == o.status != 0 and g.bad_page == o.bad_page


def test_chunk_read_into_the_handles_input_buffer(dec, oracle):
    # pqd_input_buffer: the caller fills the handle's page-locked buffer and passes that pointer (no staging copy);
    # a larger request may move the buffer, a smaller one reuses it
    rng = np.random.default_rng(77)
    for n in (5000, 200000, 300):
        cb = pg.ChunkBuilder(gap=3)
        cb.dict_page(rand_dict(rng, pg.INT64, 500), 500)
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, pg.dict_indices_body(rng.integers(0, 500, n), 9, rng, "mixed"))
        cb.data_page_v1(pg.ENC_PLAIN, n, rng.integers(-2**62, 2**62, n, dtype=np.int64).tobytes())
        chunk, pages = cb.finish()
        raw = np.frombuffer(chunk, np.uint8)
        buf = dec.input_buffer(raw.size)
        buf[:] = raw
        dec.submit(buf, pages, pg.INT64)
        g = dec.fetch_pinned()
        o = oracle.read_chunk(chunk, pages, pg.INT64, 0, 0, 0)
        assert g.status == o.status == 0 and np.array_equal(g.values, o.values)


def test_int96_cut_short_keeps_the_previous_values_tail(dec, oracle):
    # ReadPlainINT96 reads every value into one 12-byte buffer (encodingread.go:71-82): a last value cut short keeps
    # the tail of the value before it (zeros when it is the first); the same for an INT96 dictionary page
    rng = np.random.default_rng(96)
    for n, cut in ((5, 7), (1, 5), (300, 1), (300, 11)):
        body = rng.integers(1, 256, n * 12, dtype=np.uint8).tobytes()
        cb = pg.ChunkBuilder(gap=3)
        cb.data_page_v1(pg.ENC_PLAIN, n, body[:len(body) - cut])
        chunk, pages = cb.finish()
        g, o = check_same(dec, oracle, chunk, pages, pg.INT96)
        assert o.status == 0
        if n > 1:
            assert bytes(g.values[-1][12 - cut:]) == body[(n - 1) * 12 - cut:(n - 1) * 12]
    cb = pg.ChunkBuilder()
    body = rng.integers(1, 256, 40 * 12, dtype=np.uint8).tobytes()
    cb.dict_page(body[:-5], 40)
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 500, pg.dict_indices_body(rng.integers(0, 40, 500), 6, rng))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.INT96)


def test_handles_on_several_threads(oracle):
    # one handle per thread, as the reference runs one goroutine per column (reader/reader.go:306-327): decodes
    # overlap on the device, results must not depend on who else is running
    import threading
    import parquet_go_amd as pq
    rng = np.random.default_rng(4711)
    jobs = []
    for t in range(6):
        w = [3, 12, 16, 1, 8, 20][t]
        nd = min(1 << w, 70000)
        cb = pg.ChunkBuilder(gap=t)
        cb.dict_page(rand_dict(rng, pg.INT32, nd), nd)
        for n in (30000, 100000, 5):
            cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, pg.dict_indices_body(rng.integers(0, nd, n), w, rng))
        chunk, pages = cb.finish()
        jobs.append((chunk, pages, oracle.read_chunk(chunk, pages, pg.INT32)))
    errors = []

    def work(k):
        try:
            d = pq.Decoder(0)
            chunk, pages, o = jobs[k]
            for _ in range(8):
                g = d.decode(chunk, pages, pg.INT32)
                if g.status != 0 or not np.array_equal(g.values, o.values):
                    errors.append("thread %d: result differs" % k)
                    break
            d.close()
        except Exception as e:   # surfaces in the main thread below
            errors.append("thread %d: %r" % (k, e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_delta_byte_array_length_classes(dec, oracle):
    # pages of short strings (one 64-byte row of lanes), of strings up to 1 KiB (previous string kept in LDS) and with
    # longer ones (parent-chain kernels), batches that mix the first two, and a prefix that overruns its predecessor
    rng = np.random.default_rng(4242)

    def keys(n, lo, hi):
        out, prev = [], b""
        for _ in range(n):
            ln = int(rng.integers(lo, hi + 1))
            pl = int(rng.integers(0, min(len(prev), ln) + 1))
            cur = prev[:pl] + bytes(rng.integers(97, 123, ln - pl, dtype=np.uint8))
            out.append(cur)
            prev = cur
        return out

    cb = pg.ChunkBuilder(gap=2)
    for strings in (keys(3000, 0, 64), keys(1000, 0, 300), keys(700, 900, 1024), keys(300, 0, 64) + keys(200, 60, 70) + keys(300, 1, 9),
                    keys(200, 0, 3000), keys(1, 5, 5), keys(65, 64, 64)):
        cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, len(strings), pg.dba_encode(strings))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
    assert o.status == 0
    # prefix longer than the previous string: the reference panics on the slice (encodingread.go:430)
    bad = pg.delta_encode(np.array([0, 3, 9], np.int32), np.int32) + pg.dlba_encode([b"abc", b"de", b"f"])
    cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, 3, bad)
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
    assert o.status == oracle.PANIC


@pytest.mark.parametrize("ptype,max_rep", [(pg.INT64, 1), (pg.BYTE_ARRAY, 2), (pg.INT32, 0)])
def test_column_buffer_read_rows(dec, oracle, ptype, max_rep):
    # ReadRows / Table.Pop over several chunks (reader/columnbuffer.go:232-262, layout/table.go:73-103): rows that
    # continue across a chunk boundary, requests larger than what is left, requests of zero rows
    import parquet_go_amd as pq
    rng = np.random.default_rng(300 + ptype + max_rep)
    max_def = 2
    chunks, tables = [], []
    for c in range(3):
        cb = pg.ChunkBuilder(gap=1)
        for n in (1500, 1, 8000):
            dl, rl = _levels(rng, n, max_def, max_rep)
            if max_rep:
                rl[rng.random(n) < 0.7] = rng.integers(1, max_rep + 1)   # long rows
                if c == 0:
                    rl[0] = 0
                # (chunks after the first may start in the middle of a row)
            nn = int((dl == max_def).sum())
            if ptype == pg.BYTE_ARRAY:
                body = pg.plain_byte_array([b"v%d" % k for k in rng.integers(0, 1000, nn)])
            else:
                body = rng.integers(0, 2**31 - 1, nn).astype(np.int64 if ptype == pg.INT64 else np.int32).tobytes()
            cb.data_page_v1(pg.ENC_PLAIN, n, body, pg.level_block_v1(rl, max_rep, rng) if max_rep else b"",
                            pg.level_block_v1(dl, max_def, rng))
        chunk, pages = cb.finish()
        chunks.append((chunk, pages))
        tables.append(oracle.read_chunk(chunk, pages, ptype, 0, max_def, max_rep))
    model = oracle.ColumnBufferModel(tables)
    whole_def = np.concatenate([t.def_levels for t in tables])
    whole_rep = np.concatenate([t.rep_levels for t in tables])
    cbuf = pq.ColumnBuffer(dec, chunks, ptype, 0, max_def, max_rep)
    pos = 0
    for num in [0, 1, 7, 300, 2, 5000, 1, 10**9, 5]:
        want_slots, want_rows = model.read_rows(num)
        got, rows = cbuf.read_rows(num)
        assert rows == want_rows, num
        assert got.def_levels.size == want_slots, num
        assert np.array_equal(got.def_levels, whole_def[pos:pos + want_slots])
        assert np.array_equal(got.rep_levels, whole_rep[pos:pos + want_slots])
        # values of the slice against the whole-table decode
        if ptype == pg.BYTE_ARRAY:
            lens = np.diff(got.offsets)
            ref_lens = np.concatenate([np.diff(t.offsets) for t in tables])[pos:pos + want_slots]
            assert np.array_equal(lens, ref_lens)
            ref_bytes = np.concatenate([t.bytes for t in tables])
            ref_off = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(t.offsets) for t in tables]))])
            assert np.array_equal(got.bytes, ref_bytes[ref_off[pos]:ref_off[pos + want_slots]])
        else:
            ref_vals = np.concatenate([t.values for t in tables])
            assert np.array_equal(got.values, ref_vals[pos:pos + want_slots])
        pos += want_slots
    assert pos == whole_def.size


def test_empty_and_all_null(dec, oracle):
    cb = pg.ChunkBuilder()
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk or b"\0", pages, pg.INT32)
    cb = pg.ChunkBuilder()
    cb.data_page_v1(pg.ENC_PLAIN, 0, b"")
    cb.data_page_v1(pg.ENC_PLAIN, 500, b"", b"", pg.level_block_v1(np.zeros(500, np.int64), 1))   # all null
    cb.data_page_v1(pg.ENC_PLAIN, 3, np.arange(3, dtype=np.int32).tobytes(), b"", pg.level_block_v1(np.ones(3, np.int64), 1))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.INT32, 0, 1, 0)


def test_error_classes(dec, oracle):
    def one(build, ptype=pg.INT32, **kw):
        cb = pg.ChunkBuilder(lead=2)
        good = np.arange(4, dtype=np.int32 if ptype == pg.INT32 else np.int64).tobytes()
        cb.data_page_v1(pg.ENC_PLAIN, 4, good, b"", pg.level_block_v1(np.ones(4, np.int64), 1) if kw.get("max_def") else b"")
        build(cb)
        chunk, pages = cb.finish()
        g, o = check_same(dec, oracle, chunk, pages, ptype, **kw)
        assert o.status != 0
        return o.status

    d4 = np.arange(4, dtype=np.int32).tobytes()
    # dictionary index out of range
    assert one(lambda cb: (cb.dict_page(d4, 4), cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, 8, pg.dict_indices_body(np.array([0, 1, 2, 3, 4, 0, 0, 0]), 3, style="bp")))) == oracle.PANIC
    # fewer indices than the page needs
    assert one(lambda cb: (cb.dict_page(d4, 4), cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, 9, pg.dict_indices_body(np.zeros(8, np.int64), 2, style="bp")))) == oracle.PANIC
    # PLAIN section too short / empty
    assert one(lambda cb: cb.data_page_v1(pg.ENC_PLAIN, 10, b"\0" * 39)) == oracle.UNEXPECTED_EOF
    assert one(lambda cb: cb.data_page_v1(pg.ENC_PLAIN, 10, b"")) == oracle.EOF
    # bit-packed run whose payload is missing entirely
    assert one(lambda cb: (cb.dict_page(d4, 4), cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 8, bytes([2, 3])))) == oracle.EOF
    # truncated level section
    assert one(lambda cb: cb.data_page_v1(pg.ENC_PLAIN, 10, b"", b"", b"\x05\x00"), max_def=1) == oracle.UNEXPECTED_EOF
    # delta stream announcing fewer values than the page holds
    assert one(lambda cb: cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, 10, pg.delta_encode(np.arange(5), np.int64)), ptype=pg.INT64) == oracle.PANIC
    # delta stream cut in the middle of a block
    enc = pg.delta_encode(np.cumsum(np.arange(300) % 7 * 1000), np.int64)
    assert one(lambda cb: cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, 300, enc[:len(enc) // 2]), ptype=pg.INT64) == oracle.EOF
    # unsupported encodings
    assert one(lambda cb: cb.data_page_v1(4, 10, b"\0" * 40)) == oracle.UNSUPPORTED


def test_truncated_bit_packed_run_is_zero_filled(dec, oracle):
    # a final bit-packed run cut short by the end of the page decodes to zeros where bytes are missing
    # (encodingread.go:188-191) as long as one byte is left
    cb = pg.ChunkBuilder()
    cb.dict_page(np.arange(8, dtype=np.int32).tobytes(), 8)
    body = pg.dict_indices_body(np.arange(16) % 8, 3, style="bp")
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 16, body[:-2])
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.INT32)
    assert o.status == 0


def test_v2_page_without_values_whose_level_sections_outgrow_it(dec, oracle):
    # ReadPage cuts a v2 page into its sections before it reads anything (layout/page.go:736-773): level sections
    # longer than the page are a slice out of range even when the page announces no values
    rng = np.random.default_rng(343099)
    for nv, rep_len, def_len, body_len, ok in ((0, 2, 2, 2, False), (0, 2, 2, 4, True), (0, 0, 9, 3, False), (1, 2, 2, 3, False)):
        cb = pg.ChunkBuilder(gap=2)
        cb.dict_page(rng.integers(0, 256, 8, dtype=np.uint8).tobytes(), 2)
        cb.data_page_v2(pg.ENC_RLE_DICTIONARY, nv, b"", b"", b"")
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 100, pg.dict_indices_body(rng.integers(0, 2, 60), 1, rng),
                        pg.level_block_v1(np.concatenate([[0], rng.integers(0, 2, 99)]), 1, rng),
                        pg.level_block_v1(np.array([2] * 60 + [1] * 40), 2, rng))
        chunk, pages = cb.finish()
        chunk = bytearray(chunk)
        pages = pages.copy()
        pages["rep_len"][1], pages["def_len"][1], pages["length"][1] = rep_len, def_len, body_len
        g, o = check_same(dec, oracle, bytes(chunk), pages, pg.INT32, 0, 2, 1)
        assert (o.status == 0) == ok, (nv, rep_len, def_len, body_len, o.status)


@pytest.mark.parametrize("w", [1, 8, 12])
def test_one_long_bit_packed_run_cut_short_tiles_behind_the_last_byte(dec, oracle, w):
    # One bit-packed run announces tens of thousands of values and the page ends a few hundred bytes into it: the
    # tiles further on lie wholly behind the stream's last byte, and all their values are the zero fill
    # (encodingread.go:188-191) - in the index stream of a dictionary page and in a definition-level stream alike.
    rng = np.random.default_rng(1620 + w)
    nd = min(1 << w, 300)
    n = 20000
    groups = 35334
    for lead_runs in (0, 700):
        cb = pg.ChunkBuilder(gap=3, lead=2)
        cb.dict_page(rng.integers(-2**31, 2**31 - 1, nd, dtype=np.int32).tobytes(), nd)
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, pg.dict_indices_body(rng.integers(0, nd, n), w, rng))
        head = pg.hybrid_encode(rng.integers(0, nd, lead_runs), w, rng, "rle") if lead_runs else b""
        body = bytes([w]) + head + pg.varint((groups << 1) | 1) + pg.bitpack(rng.integers(0, nd, 512), w)[:509 * w // 8 + 1]
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, body)
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, pg.dict_indices_body(rng.integers(0, nd, n), w, rng))
        chunk, pages = cb.finish()
        g, o = check_same(dec, oracle, chunk, pages, pg.INT32)
        assert o.status == 0
    # the same in a definition-level stream (v2 page: the level bytes are the run itself)
    cb = pg.ChunkBuilder(gap=1)
    runs = pg.varint((groups << 1) | 1) + pg.bitpack(rng.integers(0, 2, 512), 1)[:40]
    cb.data_page_v2(pg.ENC_PLAIN, n, rng.integers(0, 256, 8 * n, dtype=np.uint8).tobytes(), b"", runs)
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.INT64, 0, 1, 0)


@pytest.mark.parametrize("w", [1, 5, 12, 20])
def test_bit_packed_payload_cut_far_short(dec, oracle, w):
    # headers announce long bit-packed runs but the page ends after a fraction of the payload: everything the
    # missing bytes would have held decodes as index 0 (encodingread.go:188-191), over many tiles
    rng = np.random.default_rng(90 + w)
    nd = min(1 << w, 3000)
    cb = pg.ChunkBuilder(gap=1)
    cb.dict_page(rng.integers(-2**31, 2**31 - 1, nd, dtype=np.int32).tobytes(), nd)
    n = 40000
    idx = rng.integers(0, nd, n)
    body = pg.dict_indices_body(idx, w, style="bp")
    for keep in (0.9, 0.31, 0.02):
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, body[:max(3, int(len(body) * keep))])
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.INT32)


def test_device_resident_entry_point(dec, oracle):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(77)
    cb = pg.ChunkBuilder(lead=16)
    nd = 4096
    cb.dict_page(rand_dict(rng, pg.INT32, nd), nd)
    idx = rng.integers(0, nd, 250000)
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, len(idx), pg.dict_indices_body(idx, 12, rng, "arrow"))
    chunk, pages = cb.finish()
    t = torch.frombuffer(bytearray(chunk), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    dec._is_bytes = False
    dec.submit_device(t.data_ptr(), t.numel(), pages, pg.INT32)
    g = dec.fetch()
    del t
    o = oracle.read_chunk(chunk, pages, oracle.INT32)
    assert g.status == 0 and np.array_equal(g.values, o.values)


# ---------------------------------------------------------------------------------------------
# BYTE_ARRAY
# ---------------------------------------------------------------------------------------------
def rand_strings(rng, n, lo=0, hi=40, alphabet=b"abcdefghijklmnopqrstuvwxyz0123456789"):
    lens = rng.integers(lo, hi + 1, n)
    pool = rng.integers(0, len(alphabet), int(lens.sum()) + 1)
    ab = np.frombuffer(alphabet, np.uint8)
    out, pos = [], 0
    for ln in lens:
        out.append(ab[pool[pos:pos + ln]].tobytes())
        pos += ln
    return out


def sorted_keys(rng, n, lo=8, hi=64):
    """Sorted keys sharing prefixes with their predecessor (prefix chains)."""
    keys = sorted(set(rand_strings(rng, n, lo, hi, b"abc")))
    return keys


def test_byte_array_plain(dec, oracle):
    rng = np.random.default_rng(60)
    cb = pg.ChunkBuilder(gap=3, lead=1)
    for n in (1, 2, 100, 5000):
        cb.data_page_v1(pg.ENC_PLAIN, n, pg.plain_byte_array(rand_strings(rng, n)))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)


def test_byte_array_dictionary(dec, oracle):
    rng = np.random.default_rng(61)
    words = rand_strings(rng, 700, 0, 30)
    cb = pg.ChunkBuilder(gap=2)
    cb.dict_page(pg.plain_byte_array(words), len(words))
    for n, style in ((1, "rle"), (4097, "mixed"), (30000, "arrow")):
        cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, n, pg.dict_indices_body(rng.integers(0, len(words), n), 10, rng, style))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
    # the flat example's "name" column: one dictionary entry, width-32 RLE-only indices (example/local_flat.go)
    cb = pg.ChunkBuilder()
    cb.dict_page(pg.plain_byte_array([b"StudentName"]), 1)
    cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, 100, bytes([32]) + oracle.write_rle_int32(np.zeros(100, np.int32), 32))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
    assert g.bytes.tobytes() == b"StudentName" * 100


@pytest.mark.parametrize("fused", ["1", "0"])
def test_byte_array_dictionary_large_pages(dec, oracle, fused):
    # dictionary-encoded strings in pages large enough for the one-pass kernel: the dictionary entries it looks up are
    # the descriptors of the strings; with nulls; same result with the kernel on and off
    dec.set_option(7, int(fused))            # PQD_OPT_FUSED
    rng = np.random.default_rng(6100)
    words = rand_strings(rng, 900, 0, 24)
    mix = {"bp": 60, "rle": 20, "rle2": 5, "bp2": 3}
    cb = pg.ChunkBuilder(gap=5)
    cb.dict_page(pg.plain_byte_array(words), len(words))
    for n_target in (50000, 20000, 100):
        body, vals = _runs_body(rng, 10, len(words), n_target, mix)
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, len(vals) - 3, bytes([10]) + body)
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
    assert o.status == 0
    dec.decode(chunk, pages, pg.BYTE_ARRAY)
    assert dec.direct_streams() == (3 if fused == "1" else 0)
    cb = pg.ChunkBuilder(gap=1)
    cb.dict_page(pg.plain_byte_array(words), len(words))
    for n in (40000, 9000):
        dl, rl = _levels(rng, n, 1, 0)
        body, vals = _runs_body(rng, 10, len(words), int((dl == 1).sum()), mix)
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, bytes([10]) + body, b"", pg.level_block_v1(dl, 1, rng, "mixed"))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY, 0, 1, 0)
    assert o.status == 0


def test_byte_array_dictionary_page_announcing_more_entries_than_it_holds(dec, oracle):
    # the dictionary page fails part way through (encodingread.go:98-125: EOF at the entry after the last one), the
    # data pages are expanded all the same and use indices of entries that were never read: those must read as
    # empty strings, whatever the work buffer held before; a larger dictionary first, so that the buffer is reused
    rng = np.random.default_rng(71)
    big = rand_strings(rng, 3000, 1, 30)
    cb = pg.ChunkBuilder()
    cb.dict_page(pg.plain_byte_array(big), len(big))
    cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, 5000, pg.dict_indices_body(rng.integers(0, len(big), 5000), 12, rng, "bp"))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
    words = rand_strings(rng, 40, 0, 12)
    for announced in (51, 64, 4000):
        cb = pg.ChunkBuilder(lead=1)
        cb.dict_page(pg.plain_byte_array(words), announced)
        w = max(6, int(announced - 1).bit_length())
        cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, 4000, pg.dict_indices_body(rng.integers(0, announced, 4000), w, rng, "mixed"))
        chunk, pages = cb.finish()
        g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
        assert o.status == oracle.EOF and o.bad_page == 0


@pytest.mark.parametrize("trim", [False, True])
@pytest.mark.parametrize("offsets", [1, 0])
def test_delta_length_byte_array(dec, oracle, trim, offsets):
    dec.set_option(8, offsets)               # PQD_OPT_BYTE_OFFSETS
    rng = np.random.default_rng(62)
    cb = pg.ChunkBuilder(gap=1, lead=2)
    cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, 4, oracle.write_delta_length_byte_array([b"Hello", b"World", b"Foobar", b"ABCDEF"]))
    for n in (1, 2, 129, 3000, 20000):
        cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, n, pg.dlba_encode(rand_strings(rng, n, 0, 64), trim))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)


@pytest.mark.parametrize("trim", [False, True])
@pytest.mark.parametrize("offsets", [1, 0])
def test_delta_byte_array_prefix_chains(dec, oracle, trim, offsets):
    dec.set_option(8, offsets)               # PQD_OPT_BYTE_OFFSETS
    rng = np.random.default_rng(63)
    cb = pg.ChunkBuilder(gap=1)
    cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, 4, oracle.write_delta_byte_array([b"Hello", b"World", b"Foobar", b"ABCDEF"]))
    for n in (1, 2, 300, 5000, 30000):
        keys = sorted_keys(rng, n)
        cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, len(keys), pg.dba_encode(keys, trim))
    # adversarial chains: constant prefix lengths, strictly growing, strictly shrinking
    same = [b"prefix-" + bytes([97 + i % 26]) for i in range(2000)]
    grow = [b"x" * i for i in range(1, 300)]
    shrink = [b"y" * i for i in range(300, 0, -1)]
    for keys in (same, grow, shrink):
        cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, len(keys), pg.dba_encode(keys, trim))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)


@pytest.mark.parametrize("offsets", [1, 0])
@pytest.mark.parametrize("enc", ["plain", "dict", "dlba", "dba"])
def test_byte_array_with_nulls(dec, oracle, enc, offsets):
    dec.set_option(8, offsets)               # PQD_OPT_BYTE_OFFSETS
    rng = np.random.default_rng(64)
    cb = pg.ChunkBuilder()
    words = rand_strings(rng, 50, 1, 12)
    if enc == "dict":
        cb.dict_page(pg.plain_byte_array(words), len(words))
    for n in (10, 4096, 9001):
        dl, rl = _levels(rng, n, 2, 1)
        nn = int((dl == 2).sum())
        if enc == "plain":
            body, e = pg.plain_byte_array(rand_strings(rng, nn)), pg.ENC_PLAIN
        elif enc == "dict":
            body, e = pg.dict_indices_body(rng.integers(0, len(words), nn), 6, rng), pg.ENC_RLE_DICTIONARY
        elif enc == "dlba":
            body, e = pg.dlba_encode(rand_strings(rng, nn)), pg.ENC_DELTA_LENGTH_BYTE_ARRAY
        else:
            body, e = pg.dba_encode(sorted(rand_strings(rng, nn, 0, 20, b"ab"))), pg.ENC_DELTA_BYTE_ARRAY
        if nn == 0:
            body = b""
        cb.data_page_v1(e, n, body, pg.level_block_v1(rl, 1, rng), pg.level_block_v1(dl, 2, rng))
    chunk, pages = cb.finish()
    check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY, 0, 2, 1)


@pytest.mark.parametrize("offsets", [1, 0])
def test_byte_array_errors(dec, oracle, offsets):
    dec.set_option(8, offsets)               # PQD_OPT_BYTE_OFFSETS
    # PLAIN value starting exactly at the end -> io.EOF
    cb = pg.ChunkBuilder()
    cb.data_page_v1(pg.ENC_PLAIN, 3, pg.plain_byte_array([b"ab", b"c"]))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
    assert o.status == oracle.EOF
    # DELTA_BYTE_ARRAY prefix longer than the previous string -> the reference panics
    body = pg.delta_encode(np.array([0, 5], np.int32), np.int32) + pg.dlba_encode([b"abc", b"d"])
    cb = pg.ChunkBuilder()
    cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, 2, body)
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
    assert o.status == oracle.PANIC
    # fewer strings than the page holds
    cb = pg.ChunkBuilder()
    cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, 5, pg.dlba_encode([b"a", b"b"]))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
    assert o.status == oracle.PANIC


def _fixed_keys(rng, n, tl):
    """n strings of tl bytes each, sorted, sharing prefixes with their predecessors."""
    ab = np.frombuffer(b"abc", np.uint8)
    return sorted(ab[rng.integers(0, 3, tl)].tobytes() for _ in range(n))


@pytest.mark.parametrize("tl", [1, 6, 16, 100, 1500])
def test_flba_under_delta_byte_array_encodings(dec, oracle, tl):
    # layout/page.go:684-706: a FIXED_LEN_BYTE_ARRAY column whose pages are DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY
    # (what the reference's writer emits for such a column tagged that way, layout/page.go:179-183), next to PLAIN
    # and dictionary pages of the same column
    rng = np.random.default_rng(90 + tl)
    cb = pg.ChunkBuilder(gap=1, lead=3)
    sizes = (1, 2, 130, 3000) if tl <= 100 else (1, 2, 70)
    for n in sizes:
        cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, n, pg.dlba_encode([rng.integers(0, 256, tl, dtype=np.uint8).tobytes() for _ in range(n)]))
        cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, n, pg.dba_encode(_fixed_keys(rng, n, tl)))
    cb.data_page_v1(pg.ENC_PLAIN, 50, rand_dict(rng, pg.FIXED_LEN_BYTE_ARRAY, 50, tl))
    cb.dict_page(rand_dict(rng, pg.FIXED_LEN_BYTE_ARRAY, 40, tl), 40)
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 777, pg.dict_indices_body(rng.integers(0, 40, 777), 6, rng))
    cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, 500 if tl <= 100 else 40, pg.dba_encode(_fixed_keys(rng, 500 if tl <= 100 else 40, tl)))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.FIXED_LEN_BYTE_ARRAY, tl)
    assert o.status == 0 and o.kind == "fixed" and o.values.shape[1] == tl


@pytest.mark.parametrize("version", [1, 2])
@pytest.mark.parametrize("enc", ["dlba", "dba"])
def test_flba_delta_with_nulls(dec, oracle, enc, version):
    rng = np.random.default_rng(97)
    tl = 12
    cb = pg.ChunkBuilder(lead=1)
    for n in (10, 4096, 9001):
        dl, rl = _levels(rng, n, 2, 1)
        nn = int((dl == 2).sum())
        if enc == "dlba":
            body, e = pg.dlba_encode([rng.integers(0, 256, tl, dtype=np.uint8).tobytes() for _ in range(nn)]), pg.ENC_DELTA_LENGTH_BYTE_ARRAY
        else:
            body, e = pg.dba_encode(_fixed_keys(rng, nn, tl)), pg.ENC_DELTA_BYTE_ARRAY
        if nn == 0:
            body = b""
        if version == 1:
            cb.data_page_v1(e, n, body, pg.level_block_v1(rl, 1, rng), pg.level_block_v1(dl, 2, rng))
        else:
            cb.data_page_v2(e, n, body, pg.hybrid_encode(rl, 1, rng), pg.hybrid_encode(dl, 2, rng))
    chunk, pages = cb.finish()
    g, o = check_same(dec, oracle, chunk, pages, pg.FIXED_LEN_BYTE_ARRAY, tl, 2, 1)
    assert o.status == 0 and o.n_nulls > 0


@pytest.mark.parametrize("enc", ["dlba", "dba"])
def test_flba_delta_string_of_another_length_is_refused(dec, oracle, enc):
    # The reference hands a delta-decoded string on whatever its length (layout/page.go:689-693, :701-705); the
    # column's fixed-width image cannot hold it: the page is refused, and the pages before it decode.
    rng = np.random.default_rng(98)
    tl = 8
    good = _fixed_keys(rng, 300, tl)
    odd = list(good)
    odd[137] = odd[137] + b"x"
    encode, e = (pg.dlba_encode, pg.ENC_DELTA_LENGTH_BYTE_ARRAY) if enc == "dlba" else (pg.dba_encode, pg.ENC_DELTA_BYTE_ARRAY)
    cb = pg.ChunkBuilder()
    cb.data_page_v1(e, 300, encode(good))
    cb.data_page_v1(e, 300, encode(odd))
    cb.data_page_v1(e, 300, encode(good))
    chunk, pages = cb.finish()
    g = dec.decode(chunk, pages, pg.FIXED_LEN_BYTE_ARRAY, tl)
    o = oracle.read_chunk(chunk, pages, pg.FIXED_LEN_BYTE_ARRAY, tl)
    assert o.status == oracle.FLBA_LENGTH and o.bad_page == 1
    assert g.status == -13 and g.bad_page == 1
    # a column of another physical type still refuses these encodings
    g = dec.decode(chunk, pages, pg.INT64)
    assert g.status == -4 and g.bad_page == 0


def test_string_lengths_that_do_not_fit_32_bits(dec, oracle):
    # chunks of one delta string encoding decode their lengths with 32-bit arithmetic where the block headers prove that
    # to be what the reference's 64-bit arithmetic gives; lengths and prefix lengths whose low words look harmless but
    # whose 64-bit values are not (2^32 + 2 where 2 bytes are there, steps of 2^32), and honest lengths with large
    # jumps, must come out as the oracle has them on either path
    rng = np.random.default_rng(909)
    good = rand_strings(rng, 300, 0, 40)
    strs = rand_strings(rng, 200, 1, 30)
    lens = np.array([len(x) for x in strs], np.int64)
    payload = b"".join(strs)

    def dlba_chunk(body, n):
        cb = pg.ChunkBuilder(gap=1, lead=3)
        cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, len(good), pg.dlba_encode(good))
        cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, n, body)
        cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, len(good), pg.dlba_encode(good))
        return cb.finish()

    cases = []
    for at, add in ((100, 1 << 32), (0, 1 << 32), (150, -(1 << 32)), (7, 1 << 40)):
        bad = lens.copy(); bad[at] += add
        cases.append(dlba_chunk(pg.delta_encode(bad, np.int64) + payload, len(strs)))
    # every length 2^32 too large: the steps are small, the first value is not
    cases.append(dlba_chunk(pg.delta_encode(lens + (1 << 32), np.int64) + payload, len(strs)))
    # honest lengths with jumps of a megabyte
    big = [b"x" * (1 << 20) if i % 50 == 7 else strs[i] for i in range(len(strs))]
    cases.append(dlba_chunk(pg.dlba_encode(big), len(big)))
    # DELTA_BYTE_ARRAY: a prefix length of 2^32 + 1, suffix lengths 2^32 too large
    keys = sorted(rand_strings(rng, 200, 4, 30))
    body = pg.dba_encode(keys)
    pre, suf, prev = [], [], b""
    for x in keys:
        k = 0
        while k < min(len(prev), len(x)) and prev[k] == x[k]:
            k += 1
        pre.append(k); suf.append(x[k:]); prev = x
    for which in (0, 1):
        p64 = np.array(pre, np.int64); s64 = np.array([len(x) for x in suf], np.int64)
        if which == 0: p64[120] += 1 << 32
        else: s64[33] += 1 << 32
        cb = pg.ChunkBuilder(gap=1, lead=3)
        cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, len(keys), body)
        cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, len(keys), pg.delta_encode(p64, np.int64) + pg.delta_encode(s64, np.int64) + b"".join(suf))
        cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, len(keys), body)
        cases.append(cb.finish())
    n_ok = 0
    for chunk, pages in cases:
        o = oracle.read_chunk(chunk, pages, pg.BYTE_ARRAY, 0, 0, 0)
        res = []
        for offsets in (1, 0):
            dec.set_option(8, offsets)
            if o.status == 0:
                g, _ = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
                n_ok += 1
            else:
                # (a length of 16 MiB or more is refused where the reference reads on to the end of the page: the page and
                #  the verdict must be the same on both paths, and the reference's where the library has no rule of its own)
                g = dec.decode(chunk, pages, pg.BYTE_ARRAY)
                assert g.bad_page == o.bad_page == 1 and g.status in (o.status, -13)
            res.append((g.status, g.bad_page))
        assert res[0] == res[1]
    assert n_ok == 2     # (the honest one, on both paths)


def test_delta_length_pages_with_bad_lengths_and_short_payloads(dec, oracle):
    # a chunk of DELTA_LENGTH_BYTE_ARRAY pages only takes its offsets straight from the decoded lengths; what the
    # per-value path checks must still be found, page by page: a negative length (make([]byte, n) panics in the
    # reference, encodingread.go:400), a payload shorter than its lengths say (the reference pads: refused), trailing
    # bytes behind the last string (ignored by the reference), a page of empty strings, each behind good pages
    rng = np.random.default_rng(6262)
    good = [rand_strings(rng, n, 0, 50) for n in (3000, 1, 700)]

    def build(bad_body, bad_n):
        cb = pg.ChunkBuilder(gap=2, lead=1)
        for strs in good[:2]:
            cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, len(strs), pg.dlba_encode(strs))
        cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, bad_n, bad_body)
        cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, len(good[2]), pg.dlba_encode(good[2]))
        return cb.finish()

    strs = rand_strings(rng, 500, 1, 30)
    lens = np.array([len(x) for x in strs], np.int64)
    payload = b"".join(strs)
    neg = lens.copy(); neg[250] = -5
    for body, n, ok in ((pg.delta_encode(lens, np.int64) + payload + b"trailing", 500, True),
                        (pg.delta_encode(neg, np.int64) + payload, 500, False),
                        (pg.delta_encode(np.zeros(40, np.int64), np.int64), 40, True)):
        chunk, pages = build(body, n)
        for offsets in (1, 0):
            dec.set_option(8, offsets)
            g, o = check_same(dec, oracle, chunk, pages, pg.BYTE_ARRAY)
            assert (o.status == 0) == ok and (ok or o.bad_page == 2)
    # a payload shorter than its lengths say: the reference zero pads the cut string (bytes.Reader.Read is lenient), the
    # library refuses the page, on either path
    chunk, pages = build(pg.delta_encode(lens, np.int64) + payload[:-7], 500)
    for offsets in (1, 0):
        dec.set_option(8, offsets)
        g = dec.decode(chunk, pages, pg.BYTE_ARRAY)
        assert g.status == -13 and g.bad_page == 2
