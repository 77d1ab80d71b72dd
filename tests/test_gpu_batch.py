"""GPU parity of what one decode call may hold beyond a plain chunk: several chunks of a column in one call,
dictionary pages in any position, Encoding_RLE values, BOOLEAN dictionaries, byte outputs that outgrow their
estimate, calls on a handle whose previous call was never waited for.  Checked against the CPU oracle."""
import struct

import numpy as np
import pytest

import pagegen as pg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dec():
    import parquet_go_amd as pq
    d = pq.Decoder(0)
    yield d
    d.close()


def same_table(g, o):
    assert g.status == o.status == 0, (g.status, g.bad_page, o.status, o.bad_page)
    assert (g.n_slots, g.n_rows, g.n_nulls) == (o.n_slots, o.n_rows, o.n_nulls)
    assert np.array_equal(g.def_levels, o.def_levels) and np.array_equal(g.rep_levels, o.rep_levels)
    if o.kind == "fixed":
        assert np.array_equal(g.values, o.values)
    else:
        assert np.array_equal(g.offsets, o.offsets) and np.array_equal(g.bytes, o.bytes)


def concat_oracle(oracle, parts, ptype, tl=0, md=0, mr=0):
    """The chunks read one after the other and merged, as ColumnBufferType does across row groups."""
    tabs = [oracle.read_chunk(c, p, ptype, tl, md, mr) for c, p in parts]
    assert all(t.status == 0 for t in tabs)
    out = oracle.Table()
    out.status, out.kind = 0, tabs[0].kind
    out.n_slots = sum(t.n_slots for t in tabs)
    out.n_rows = sum(t.n_rows for t in tabs)
    out.n_nulls = sum(t.n_nulls for t in tabs)
    out.def_levels = np.concatenate([t.def_levels for t in tabs])
    out.rep_levels = np.concatenate([t.rep_levels for t in tabs])
    if out.kind == "fixed":
        out.values = np.concatenate([t.values for t in tabs])
    else:
        offs, base = [np.zeros(1, np.int64)], 0
        for t in tabs:
            offs.append(t.offsets[1:] + base)
            base += int(t.offsets[-1])
        out.offsets = np.concatenate(offs)
        out.bytes = np.concatenate([t.bytes for t in tabs])
    return out


def int32_dict_chunk(rng, n_pages, page_values, w, nd, lead=0, style="mixed"):
    cb = pg.ChunkBuilder(lead=lead, gap=int(rng.integers(0, 5)))
    cb.dict_page(rng.integers(-2**31, 2**31 - 1, nd, dtype=np.int64).astype(np.int32).tobytes(), nd)
    for _ in range(n_pages):
        n = int(rng.integers(1, page_values + 1))
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, pg.dict_indices_body(rng.integers(0, nd, n), w, rng, style))
    return cb.finish()


@pytest.mark.parametrize("device_side", [False, True])
def test_many_chunks_of_a_column_in_one_call(dec, oracle, device_side):
    import torch
    rng = np.random.default_rng(515)
    # every chunk has its own dictionary (of its own size), page counts and sizes differ, one chunk is empty of pages
    parts = [int32_dict_chunk(rng, int(rng.integers(1, 6)), 9000, w, nd, lead=int(rng.integers(0, 9)),
                              style=("rle" if k % 3 == 0 else "mixed"))
             for k, (w, nd) in enumerate([(3, 8), (12, 4096), (1, 2), (16, 40000), (7, 100), (20, 70000), (5, 32)])]
    parts.insert(3, (b"", np.zeros(0, pg.PAGE_DESC_DTYPE)))
    o = concat_oracle(oracle, [p for p in parts if len(p[1])], pg.INT32)
    if device_side:
        keep = []
        for c, _ in parts:
            t = torch.zeros(len(c) + 64, dtype=torch.uint8, device="cuda")
            if len(c):
                t[:len(c)].copy_(torch.from_numpy(np.frombuffer(c, np.uint8).copy()))
            keep.append(t)
        torch.cuda.synchronize()
        dec.submit_chunks_device([t.data_ptr() for t in keep], [len(c) for c, _ in parts], [p for _, p in parts], pg.INT32)
    else:
        dec.submit_chunks([c for c, _ in parts], [p for _, p in parts], pg.INT32)
    g = dec.fetch()
    same_table(g, o)
    # per-page counts cover the pages of all chunks in order
    n_pages = sum(len(p) for _, p in parts)
    assert g.page_counts.shape == (n_pages, 3)
    assert int(g.page_counts[:, 2].sum()) == o.n_slots


@pytest.mark.parametrize("tables", [1, 0])
def test_chunks_of_thousands_of_small_pages_in_one_call(dec, oracle, tables):
    # row groups as the reference's writer cuts them - hundreds of pages of a few dozen values per chunk, a dictionary
    # per chunk - with the per-stream tables of the call worked out by a kernel (and the page entries sent ahead in
    # slices while the host walks the rest: from the second call on) or by the host
    dec.set_option(9, tables)                # PQD_OPT_DEVICE_TABLES
    try:
        rng = np.random.default_rng(919)
        parts = [int32_dict_chunk(rng, 700, 60, w, nd, lead=int(rng.integers(0, 9)), style=("rle" if k % 2 else "mixed"))
                 for k, (w, nd) in enumerate([(3, 8), (12, 4096), (1, 2), (9, 300), (7, 100), (32, 500)])]
        o = concat_oracle(oracle, parts, pg.INT32)
        for _ in range(2):
            dec.submit_chunks([c for c, _ in parts], [p for _, p in parts], pg.INT32)
            same_table(dec.fetch(), o)
    finally:
        dec.set_option(9, -1)


def test_many_small_nested_chunks_in_one_call(dec, oracle):
    # row groups of a nested column (optional list of optional int64: levels + PLAIN values), a few pages each
    from gen import pqgen
    parts = []
    for k in range(12):
        chunk, pages = pqgen.list_chunk(0xC5000 + k, 20000 + 777 * k, 4096)
        parts.append((chunk, pages))
    o = concat_oracle(oracle, parts, pg.INT64, 0, 3, 1)
    dec.submit_chunks([c for c, _ in parts], [p for _, p in parts], pg.INT64, 0, 3, 1)
    g = dec.fetch()
    same_table(g, o)
    # levels only, same call shape (skip path)
    dec.submit_chunks([c for c, _ in parts], [p for _, p in parts], pg.INT64, 0, 3, 1, levels_only=True)
    r = dec.wait()
    assert r.status == 0 and (r.num_slots, r.num_rows, r.num_nulls) == (o.n_slots, o.n_rows, o.n_nulls)


def test_byte_array_chunks_in_one_call_and_an_output_that_outgrows_its_estimate(dec, oracle):
    rng = np.random.default_rng(99)
    parts = []
    # dictionary of long strings referenced many times: the bytes handed out are a multiple of the page bytes, so the
    # first sizing of the byte output is too small and the copies are done again when the call is waited for
    words = [bytes(rng.integers(97, 123, int(rng.integers(200, 400)), dtype=np.uint8)) for _ in range(16)]
    cb = pg.ChunkBuilder()
    cb.dict_page(pg.plain_byte_array(words), len(words))
    for _ in range(3):
        n = 5000
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, pg.dict_indices_body(rng.integers(0, 16, n), 4, rng, "arrow"))
    parts.append(cb.finish())
    # sorted keys with long shared prefixes: DELTA_BYTE_ARRAY expands far beyond its page bytes too
    keys = sorted(b"tenant-0001/region-eu/" + b"%012d" % int(x) for x in rng.integers(0, 10**9, 4000))
    cb = pg.ChunkBuilder(lead=3)
    cb.data_page_v1(pg.ENC_DELTA_BYTE_ARRAY, len(keys), pg.dba_encode(keys))
    cb.data_page_v1(pg.ENC_DELTA_LENGTH_BYTE_ARRAY, 100, pg.dlba_encode(keys[:100]))
    cb.data_page_v1(pg.ENC_PLAIN, 50, pg.plain_byte_array(keys[100:150]))
    parts.append(cb.finish())
    import parquet_go_amd as pq
    fresh = pq.Decoder(0)          # a new handle: no byte buffer grown by earlier tests
    try:
        for sub in (parts[:1], parts[1:], parts):
            o = concat_oracle(oracle, sub, pg.BYTE_ARRAY)
            fresh.submit_chunks([c for c, _ in sub], [p for _, p in sub], pg.BYTE_ARRAY)
            g = fresh.fetch()
            same_table(g, o)
            assert int(g.offsets[-1]) > sum(len(c) for c, _ in sub[:1])
    finally:
        fresh.close()


def test_dictionary_pages_follow_the_page_order(dec, oracle):
    rng = np.random.default_rng(7)
    d1 = rng.integers(0, 1000, 50).astype(np.int64)
    d2 = rng.integers(10**12, 10**13, 300).astype(np.int64)
    cb = pg.ChunkBuilder()
    cb.dict_page(d1.tobytes(), 50)
    cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, 5000, pg.dict_indices_body(rng.integers(0, 50, 5000), 6, rng))
    cb.data_page_v1(pg.ENC_PLAIN, 10, rng.integers(0, 9, 10).astype(np.int64).tobytes())
    cb.dict_page(d2.tobytes(), 300)                     # replaces the first one (reader/columnbuffer.go:132-135)
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 9000, pg.dict_indices_body(rng.integers(0, 300, 9000), 9, rng, "arrow"))
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 3, pg.dict_indices_body([299, 0, 17], 9, rng, "rle"))
    chunk, pages = cb.finish()
    o = oracle.read_chunk(chunk, pages, pg.INT64)
    g = dec.decode(chunk, pages, pg.INT64)
    same_table(g, o)
    # an index that only the other dictionary holds: out of range (layout/page.go:161 panics)
    cb = pg.ChunkBuilder()
    cb.dict_page(d2.tobytes(), 300)
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 4, pg.dict_indices_body([1, 2, 3, 4], 9, rng, "rle"))
    cb.dict_page(d1.tobytes(), 50)
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 4, pg.dict_indices_body([1, 2, 60, 4], 9, rng, "rle"))
    chunk, pages = cb.finish()
    o = oracle.read_chunk(chunk, pages, pg.INT64)
    g = dec.decode(chunk, pages, pg.INT64)
    assert o.status == oracle.PANIC and g.status == o.status and g.bad_page == o.bad_page == 3
    # dictionary-encoded pages in front of the dictionary page: Page.Decode(nil) leaves the indices in place of the
    # values (layout/page.go:139-143), which has no typed result; both sides name that page
    cb = pg.ChunkBuilder()
    cb.data_page_v1(pg.ENC_PLAIN, 10, rng.integers(0, 9, 10).astype(np.int64).tobytes())
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 4, pg.dict_indices_body([1, 2, 3, 4], 9, rng, "rle"))
    cb.dict_page(d1.tobytes(), 50)
    chunk, pages = cb.finish()
    o = oracle.read_chunk(chunk, pages, pg.INT64)
    g = dec.decode(chunk, pages, pg.INT64)
    assert o.status == oracle.UNSUPPORTED and o.bad_page == 1
    assert g.status == -11 and g.bad_page == 1


def test_boolean_dictionary(dec, oracle):
    rng = np.random.default_rng(11)
    cb = pg.ChunkBuilder()
    cb.dict_page(bytes([0b0110]), 3)                       # false, true, true (encodingread.go:35-55)
    idx = rng.integers(0, 3, 20000)
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, len(idx), pg.dict_indices_body(idx, 2, rng, "arrow"))
    dl = (rng.random(7000) < 0.8).astype(np.int64)
    chunk, pages = cb.finish()
    same_table(dec.decode(chunk, pages, pg.BOOLEAN), oracle.read_chunk(chunk, pages, pg.BOOLEAN))
    # nullable, and a dictionary page with fewer bytes than entries ask for (zero filled)
    cb = pg.ChunkBuilder()
    cb.dict_page(bytes([0xA5]), 12)
    cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, len(dl), pg.dict_indices_body(rng.integers(0, 12, int(dl.sum())), 4, rng), b"",
                    pg.level_block_v1(dl, 1, rng, "arrow"))
    chunk, pages = cb.finish()
    same_table(dec.decode(chunk, pages, pg.BOOLEAN, 0, 1, 0), oracle.read_chunk(chunk, pages, pg.BOOLEAN, 0, 1, 0))
    # an empty dictionary page body: io.EOF from ReadBitPacked
    cb = pg.ChunkBuilder()
    cb.dict_page(b"", 3)
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 4, pg.dict_indices_body([0, 1, 2, 0], 2, rng, "rle"))
    chunk, pages = cb.finish()
    o = oracle.read_chunk(chunk, pages, pg.BOOLEAN)
    g = dec.decode(chunk, pages, pg.BOOLEAN)
    assert o.status == oracle.EOF and g.status == o.status and g.bad_page == o.bad_page == 0


@pytest.mark.parametrize("ptype,tl", [(pg.INT32, 7), (pg.INT64, 20), (pg.INT32, 32), (pg.BOOLEAN, 1), (pg.BOOLEAN, 0), (pg.INT64, 0)])
def test_rle_encoded_values(dec, oracle, ptype, tl):
    # Encoding_RLE data pages (layout/page.go:657-667): a length-prefixed hybrid stream whose bit width is the
    # column's type length - 0 for a BOOLEAN column whose schema leaves the length unset, which makes the reference
    # read every run as zeros without consuming value bytes
    rng = np.random.default_rng(31 + tl + ptype)
    cb = pg.ChunkBuilder(lead=5)
    for n in (1, 700, 10000):
        vals = rng.integers(0, 1 << tl, n) if tl else np.zeros(n, np.int64)
        body = pg.hybrid_encode(vals, tl, rng, "mixed" if tl else "rle")
        cb.data_page_v1(pg.ENC_RLE, n, struct.pack("<i", len(body)) + body)
    nulls = (rng.random(3000) < 0.7).astype(np.int64)
    chunk, pages = cb.finish()
    same_table(dec.decode(chunk, pages, ptype, tl), oracle.read_chunk(chunk, pages, ptype, tl))
    cb = pg.ChunkBuilder()
    vals = rng.integers(0, 1 << tl, int(nulls.sum())) if tl else np.zeros(int(nulls.sum()), np.int64)
    body = pg.hybrid_encode(vals, tl, rng, "arrow" if tl else "rle")
    cb.data_page_v2(pg.ENC_RLE, len(nulls), struct.pack("<i", len(body)) + body, b"", pg.hybrid_encode(nulls, 1, rng, "arrow"))
    # a length prefix that runs past the page, and a stream that holds too few values
    cb.data_page_v1(pg.ENC_RLE, 5, b"", b"", pg.level_block_v1(np.ones(5, np.int64), 1, rng, "rle"))
    chunk, pages = cb.finish()
    o = oracle.read_chunk(chunk, pages, ptype, tl, 1, 0)
    g = dec.decode(chunk, pages, ptype, tl, 1, 0)
    assert o.status != 0 and g.status == o.status and g.bad_page == o.bad_page == 1
    o = oracle.read_chunk(chunk, pages[:1], ptype, tl, 1, 0)
    g = dec.decode(chunk, pages[:1], ptype, tl, 1, 0)
    same_table(g, o)
    # a column type without a typed image of such values
    g = dec.decode(chunk, pages[:1], pg.DOUBLE, tl, 1, 0)
    o = oracle.read_chunk(chunk, pages[:1], pg.DOUBLE, tl, 1, 0)
    assert g.status == o.status == -6 and g.bad_page == o.bad_page == 0


def test_a_call_on_a_handle_that_was_never_waited_for(oracle):
    import parquet_go_amd as pq
    rng = np.random.default_rng(5)
    d = pq.Decoder(0)
    try:
        a = int32_dict_chunk(rng, 4, 50000, 12, 4096)
        b = int32_dict_chunk(rng, 2, 3000, 3, 8)
        d.submit(a[0], a[1], pg.INT32)
        d.submit(b[0], b[1], pg.INT32)      # waits for the first call, whose result is dropped
        g = d.fetch()
        same_table(g, oracle.read_chunk(b[0], b[1], pg.INT32))
        # a call that fails its argument check leaves nothing to wait for
        bad = b[1].copy()
        bad["offset"][1] = len(b[0]) + 5
        with pytest.raises(pq.DecodeError):
            d.submit(b[0], bad, pg.INT32)
        with pytest.raises(pq.DecodeError):
            d.wait()
        d.submit(a[0], a[1], pg.INT32)
        same_table(d.fetch(), oracle.read_chunk(a[0], a[1], pg.INT32))
    finally:
        d.close()


@pytest.mark.parametrize("page_values", [5000, 70000])
def test_delta_byte_array_pages_without_a_restart_are_cut_anyway(dec, oracle, page_values):
    # sorted keys (a zero-padded counter and a suffix): no string inside a page has prefix length 0, so the only place
    # where a chain restarts is the page's first string; pages of 4096 strings and more are cut into ranges at
    # arbitrary strings, each range working out what its first string shares with the string before it
    from gen import pqgen
    chunk, pages = pqgen.strings_chunk(0xBEEF, 3 * page_values + 17, page_values, 3)
    g = dec.decode(chunk, pages, pg.BYTE_ARRAY)
    o = oracle.read_chunk(chunk, pages, pg.BYTE_ARRAY)
    assert g.status == o.status == 0
    assert np.array_equal(g.offsets, o.offsets)
    assert np.array_equal(g.bytes, o.bytes)


def test_delta_pages_ending_chunks_that_end_their_allocations(dec, oracle):
    # The walk of the block headers asks for 20 bytes at every guessed header.  A call over several device chunks spans
    # from the lowest to the highest address; reads must stay within each chunk's length rounded up to 16 all the same.
    # Every chunk here ends exactly where its own device allocation ends, with a block header in its last bytes.
    import ctypes as C
    import torch
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so")
    rng = np.random.default_rng(516)
    parts, ptrs, allocs = [], [], []
    size = 2 << 20
    for k in range(5):
        cb = pg.ChunkBuilder(lead=int(rng.integers(0, 16)))
        for n in (130 + 128 * k, 1000 + k, 4000):
            # (the last block holds few values: its header sits within 5 to 19 bytes of the page's end)
            vals = np.cumsum(rng.integers(0, 3, n)).astype(np.int64)
            cb.data_page_v1(pg.ENC_DELTA_BINARY_PACKED, n, pg.delta_encode(vals, np.int64, trim=True))
        chunk, pages = cb.finish()
        parts.append((chunk, pages))
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(size)) == 0
        allocs.append(p)
        at = p.value + size - (len(chunk) + 15) // 16 * 16
        assert hip.hipMemcpy(C.c_void_p(at), chunk, C.c_size_t(len(chunk)), C.c_int(1)) == 0
        ptrs.append(at)
    try:
        o = concat_oracle(oracle, parts, pg.INT64)
        dec.submit_chunks_device(ptrs, [len(c) for c, _ in parts], [p for _, p in parts], pg.INT64)
        g = dec.fetch()
        same_table(g, o)
    finally:
        for p in allocs:
            hip.hipFree(p)
