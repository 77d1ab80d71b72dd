"""The entry point that allocates nothing (pqd_decode_chunk_device_noalloc) against pqd_decode_chunk_device: same
chunks, bit-exact outputs, two calls in a row with the second one smaller, buffers sized exactly (a canary behind
each), and the refusals (work space exceeded, output too small, BYTE_ARRAY)."""
import numpy as np
import pytest

import pagegen as pg

pytestmark = pytest.mark.gpu

CANARY = 0xA7


def dev_bytes(torch, n, fill=None):
    t = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
    t.fill_(CANARY if fill is None else fill)
    return t


def run_both(torch, pq, dec, chunk, pages, ptype, tl=0, md=0, mr=0):
    a = np.frombuffer(chunk, np.uint8)
    d_in = torch.zeros(a.size + 64, dtype=torch.uint8, device="cuda")
    d_in[:a.size].copy_(torch.from_numpy(a.copy()))
    torch.cuda.synchronize()
    dec.submit_device(d_in.data_ptr(), a.size, pages, ptype, tl, md, mr)
    want = dec.fetch()
    slots = int(sum(max(int(p["num_values"]), 0) for p in pages if p["page_type"] != pg.DICTIONARY_PAGE))
    esz = {pg.BOOLEAN: 1, pg.INT32: 4, pg.INT64: 8, pg.FLOAT: 4, pg.DOUBLE: 8, pg.INT96: 12}.get(ptype, tl)
    vals = dev_bytes(torch, slots * esz)
    dl = dev_bytes(torch, slots * 4) if md else None
    rl = dev_bytes(torch, slots * 4) if mr else None
    rc, r = pq.binding.decode_chunk_device_noalloc(0, 0, d_in.data_ptr(), a.size, pages, ptype, tl, md, mr, vals.data_ptr(), slots * esz,
                                                   dl.data_ptr() if md else 0, rl.data_ptr() if mr else 0, slots)
    torch.cuda.synchronize()
    assert rc == r.status == want.status, (rc, r.status, want.status, pq.lib().pqd_last_error())
    if want.status != 0:
        assert r.bad_page == want.bad_page
        return
    assert (r.num_slots, r.num_rows, r.num_nulls) == (want.n_slots, want.n_rows, want.n_nulls)
    got = vals.cpu().numpy()
    assert np.array_equal(got[:slots * esz].reshape(slots, esz), want.values)
    assert (got[slots * esz:] == CANARY).all(), "bytes written behind the value buffer"
    for t, lv in ((dl, want.def_levels), (rl, want.rep_levels)):
        if t is not None:
            g = t.cpu().numpy()
            assert np.array_equal(g[:slots * 4].view("<i4"), lv)
            assert (g[slots * 4:] == CANARY).all(), "bytes written behind a level buffer"


def test_no_alloc_entry_point_matches_the_allocating_one():
    import torch
    import parquet_go_amd as pq
    from gen import pqgen
    rng = np.random.default_rng(77)
    dec = pq.Decoder(0)
    try:
        # two calls in a row, the second with a smaller input
        chunk, pages = pqgen.dict_chunk(0x77, 300000, 12, 4096, 65536)
        run_both(torch, pq, dec, chunk, pages, pg.INT32)
        chunk, pages = pqgen.dict_chunk(0x78, 5000, 3, 8, 2048)
        run_both(torch, pq, dec, chunk, pages, pg.INT32)
        # nested nullable column: levels, dense values and the scatter all come out of the work space
        chunk, pages = pqgen.list_chunk(0x79, 200000, 16384)
        run_both(torch, pq, dec, chunk, pages, pg.INT64, 0, 3, 1)
        chunk, pages = pqgen.list_chunk(0x7A, 3000, 1024)
        run_both(torch, pq, dec, chunk, pages, pg.INT64, 0, 3, 1)
        # delta, and a chunk with a failing page
        chunk, pages = pqgen.delta_chunk(0x7B, 100000, 12, 8192)
        run_both(torch, pq, dec, chunk, pages, pg.INT64)
        cb = pg.ChunkBuilder()
        cb.dict_page(rng.integers(0, 100, 16).astype(np.int32).tobytes(), 16)
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 100, pg.dict_indices_body(rng.integers(0, 16, 100), 4, rng))
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 100, pg.dict_indices_body(rng.integers(0, 16, 100), 4, rng)[:9])
        chunk, pages = cb.finish()
        run_both(torch, pq, dec, chunk, pages, pg.INT32)
        chunk, pages = pqgen.dict_chunk(0x7C, 70000, 16, 40000, 8192)
        run_both(torch, pq, dec, chunk, pages, pg.INT32)

        # value buffers whose sizes are no multiple of 16 bytes (a 16-byte store that spilled past the last value would
        # land in the canary): INT32 dictionary, PLAIN, a nullable PLAIN column, BOOLEAN
        for n in (4097, 1001, 13):
            cb = pg.ChunkBuilder(lead=3)
            cb.dict_page(rng.integers(0, 1000, 50).astype(np.int32).tobytes(), 50)
            cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, pg.dict_indices_body(rng.integers(0, 50, n), 6, rng))
            chunk, pages = cb.finish()
            run_both(torch, pq, dec, chunk, pages, pg.INT32)
            cb = pg.ChunkBuilder()
            cb.data_page_v1(pg.ENC_PLAIN, n, rng.integers(-9, 9, n).astype(np.int32).tobytes())
            chunk, pages = cb.finish()
            run_both(torch, pq, dec, chunk, pages, pg.INT32)
            dl = rng.integers(0, 2, n)
            cb = pg.ChunkBuilder()
            cb.data_page_v1(pg.ENC_PLAIN, n, rng.integers(-9, 9, int(dl.sum())).astype(np.int32).tobytes(), b"", pg.level_block_v1(dl, 1, rng))
            chunk, pages = cb.finish()
            run_both(torch, pq, dec, chunk, pages, pg.INT32, 0, 1, 0)
            cb = pg.ChunkBuilder()
            cb.data_page_v1(pg.ENC_PLAIN, n, np.packbits(rng.integers(0, 2, n).astype(np.uint8), bitorder="little").tobytes())
            chunk, pages = cb.finish()
            run_both(torch, pq, dec, chunk, pages, pg.BOOLEAN)
        # a chunk without level slots needs no value buffer
        cb = pg.ChunkBuilder()
        cb.dict_page(rng.integers(0, 1000, 50).astype(np.int32).tobytes(), 50)
        chunk, pages = cb.finish()
        a = np.frombuffer(chunk, np.uint8)
        d_in = torch.zeros(a.size + 64, dtype=torch.uint8, device="cuda")
        d_in[:a.size].copy_(torch.from_numpy(a.copy()))
        torch.cuda.synchronize()
        rc, r = pq.binding.decode_chunk_device_noalloc(0, 0, d_in.data_ptr(), a.size, pages, pg.INT32, 0, 0, 0, 0, 0, 0, 0, 0)
        assert rc == 0 and r.num_slots == 0, (rc, pq.lib().pqd_last_error())

        # the property the entry point is named for: a warmed-up call leaves the device's free memory as it was
        chunk, pages = pqgen.dict_chunk(0x77, 300000, 12, 4096, 65536)
        a = np.frombuffer(chunk, np.uint8)
        d_in = torch.zeros(a.size + 64, dtype=torch.uint8, device="cuda")
        d_in[:a.size].copy_(torch.from_numpy(a.copy()))
        vals = dev_bytes(torch, 300000 * 4)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for _ in range(3):
            rc, r = pq.binding.decode_chunk_device_noalloc(0, 0, d_in.data_ptr(), a.size, pages, pg.INT32, 0, 0, 0, vals.data_ptr(), 300000 * 4, 0, 0, 0)
            assert rc == 0
        assert torch.cuda.mem_get_info()[0] == free_before, "device memory was allocated or freed during the calls"

        # refusals: more work space than the array holds, an output that is too small, BYTE_ARRAY
        chunk, pages = pqgen.list_chunk(0x7D, 6 << 20, 1 << 20)
        a = np.frombuffer(chunk, np.uint8)
        d_in = torch.zeros(a.size + 64, dtype=torch.uint8, device="cuda")
        d_in[:a.size].copy_(torch.from_numpy(a.copy()))
        slots = 6 << 20
        vals, dl, rl = dev_bytes(torch, slots * 8), dev_bytes(torch, slots * 4), dev_bytes(torch, slots * 4)
        rc, r = pq.binding.decode_chunk_device_noalloc(0, 0, d_in.data_ptr(), a.size, pages, pg.INT64, 0, 3, 1, vals.data_ptr(), slots * 8,
                                                       dl.data_ptr(), rl.data_ptr(), slots)
        assert rc == -10, rc
        assert (vals.cpu().numpy() == CANARY).all()
        chunk, pages = pqgen.dict_chunk(0x7E, 5000, 3, 8, 2048)
        a = np.frombuffer(chunk, np.uint8)
        d_in = torch.zeros(a.size + 64, dtype=torch.uint8, device="cuda")
        d_in[:a.size].copy_(torch.from_numpy(a.copy()))
        vals = dev_bytes(torch, 5000 * 4)
        rc, r = pq.binding.decode_chunk_device_noalloc(0, 0, d_in.data_ptr(), a.size, pages, pg.INT32, 0, 0, 0, vals.data_ptr(), 4999 * 4, 0, 0, 0)
        assert rc == -8, rc
        rc, r = pq.binding.decode_chunk_device_noalloc(0, 0, d_in.data_ptr(), a.size, pages, pg.BYTE_ARRAY, 0, 0, 0, vals.data_ptr(), 5000 * 4, 0, 0, 0)
        assert rc == -8, rc
        # and the entry point still works after them
        run_both(torch, pq, dec, chunk, pages, pg.INT32)
    finally:
        dec.close()
