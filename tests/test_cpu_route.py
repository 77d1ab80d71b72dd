"""Chunks the GPU path does not take go to the host's own page functions, as integration/go/layout_gpu/page_read_gpu.go
hands them to the reference's ReadPage / ReadPageRawData (kept in the package under private names): a build with the GPU
path must not lose files the reference reads.  The route here is a stand-in that records what it is given and answers
with the oracle's decode of the pages (test side only: the product has no CPU decoder)."""
import numpy as np
import pytest

import pagegen as pg


class RecordingRoute(object):
    """cpu_route factory + layout object: read_page() hands out the pages from header_offset on, decoded by the oracle."""

    def __init__(self, oracle, col):
        self.oracle, self.col = oracle, col
        self.calls = []

    def __call__(self, chunk, header_offset, dict_header_offset):
        import parquet_go_amd as pq
        self.calls.append((header_offset, dict_header_offset))
        self.chunk = np.asarray(chunk)
        # page headers are not thrift here (pagegen chunks carry descriptors beside them): the test gives them to the route
        self.cursor = [i for i, d in enumerate(self.pages) if self._hdr(i) == header_offset][0]
        self.dict_index = None if dict_header_offset is None else [i for i, d in enumerate(self.pages) if self._hdr(i) == dict_header_offset][0]
        return self

    def _hdr(self, i):
        return 0 if i == 0 else int(self.pages[i - 1]["offset"] + self.pages[i - 1]["length"])

    def read_dictionary(self):
        t = self.oracle.read_chunk(self.chunk, self.pages[self.dict_index:self.dict_index + 1], *self.col)
        return ("dictionary table of page", self.dict_index, t.status)

    def read_page(self):
        if self.cursor >= len(self.pages):
            raise EOFError
        k = self.cursor
        self.cursor += 1
        lead = [] if self.dict_index is None or self.pages[k]["page_type"] == pg.DICTIONARY_PAGE else [self.pages[self.dict_index]]
        sel = np.array(lead + [self.pages[k]], dtype=self.pages.dtype)
        return ("cpu page", k, self.oracle.read_chunk(self.chunk, sel, *self.col))


def test_compressed_chunk_goes_to_the_cpu_route_without_touching_the_gpu(oracle):
    # parquet.CompressionCodec_SNAPPY = 1: what the reference's writer produces by default (writer/writer.go:63-65)
    from parquet_go_amd.layout_gpu import ChunkPager
    col = (pg.INT32, 0, 0, 0)
    route = RecordingRoute(oracle, col)
    cb = pg.ChunkBuilder()
    cb.data_page_v1(pg.ENC_PLAIN, 5, np.arange(5, dtype=np.int32).tobytes())
    chunk, pages = cb.finish()
    route.pages = pages
    pager = ChunkPager(None, chunk, *col, codec=1, cpu_route=route)       # no decoder: nothing may reach the library
    assert route.calls == [(0, None)]
    tag, k, t = pager.read_page()
    assert tag == "cpu page" and k == 0 and t.status == 0
    with pytest.raises(EOFError):
        pager.read_page()
    # without a route the chunk is refused with the reference's "unsupported" class of error, not decoded as if it were plain
    import parquet_go_amd as pq
    with pytest.raises(pq.binding.DecodeError) as e:
        ChunkPager(None, chunk, *col, codec=1)
    assert e.value.code == -4


@pytest.mark.gpu
def test_chunk_goes_over_to_the_cpu_route_at_the_page_the_library_does_not_take(oracle):
    # Encoding_RLE values of a FLOAT column (layout/page.go:657-667: the reference hands out int64 numbers, which the typed
    # result cannot hold: PQD_ERR_UNSUPPORTED_TYPE) behind a dictionary page and two good pages: those come from the GPU,
    # the third data page and everything behind it from the route, which is also asked for the dictionary's table.
    import parquet_go_amd as pq
    from parquet_go_amd.layout_gpu import ChunkPager
    rng = np.random.default_rng(5)
    col = (pg.FLOAT, 0, 0, 0)
    cb = pg.ChunkBuilder(lead=3)
    cb.dict_page(rng.random(16, dtype=np.float32).tobytes(), 16)
    for _ in range(2):
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 100, pg.dict_indices_body(rng.integers(0, 16, 100), 4, rng))
    body = pg.hybrid_encode(rng.integers(0, 2, 64), 32, rng, "rle")
    cb.data_page_v1(pg.ENC_RLE, 64, len(body).to_bytes(4, "little") + body)
    cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 50, pg.dict_indices_body(rng.integers(0, 16, 50), 4, rng))
    chunk, pages = cb.finish()
    route = RecordingRoute(oracle, col)
    route.pages = pages
    dec = pq.Decoder(0)
    try:
        pager = ChunkPager(dec, b"", *col, cpu_route=route)     # (pagegen chunks carry no thrift headers: the descriptors are given)
        pager.chunk, pager.pages = np.frombuffer(chunk, np.uint8), pages
        page, n, rows = pager.read_page()
        assert page.page_type == pg.DICTIONARY_PAGE and page.data_table is None
        dict_page = page
        o = oracle.read_chunk(chunk, pages[:3], *col)
        for k in (1, 2):
            page, n, rows = pager.read_page()
            assert n == 100 and np.array_equal(page.data_table.values, o.values[(k - 1) * 100:k * 100])
        assert route.calls == []
        tag, k, t = pager.read_page()
        hdr3 = int(pages[2]["offset"] + pages[2]["length"])
        assert route.calls == [(hdr3, 0)] and tag == "cpu page" and k == 3
        assert t.status == oracle.UNSUPPORTED_TYPE          # (the checker's stand-in for "int64 numbers in a FLOAT column")
        assert dict_page.data_table == ("dictionary table of page", 0, 0)
        tag, k, t = pager.read_page()
        assert k == 4 and t.status == 0 and t.n_slots == 50
        with pytest.raises(EOFError):
            pager.read_page()
        # without a route the failing page raises the library's code, as before
        pager = ChunkPager(dec, b"", *col)
        pager.chunk, pager.pages = np.frombuffer(chunk, np.uint8), pages
        for _ in range(3):
            pager.read_page()
        with pytest.raises(pq.binding.DecodeError) as e:
            pager.read_page()
        assert e.value.code == -6 and e.value.page == 3
    finally:
        dec.close()
