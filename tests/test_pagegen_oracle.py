"""Oracle vs the spec-conformant page builders: encodings and shapes the reference's own tests never
touch (bit-packed runs, widths 1..32, multi-block deltas, trimmed blocks, nulls, v2 pages)."""
import numpy as np
import pytest

import pagegen as pg


@pytest.mark.parametrize("w", [0, 1, 2, 3, 5, 7, 8, 9, 12, 16, 17, 20, 24, 31, 32])
@pytest.mark.parametrize("style", ["mixed", "rle", "bp", "arrow"])
def test_hybrid_builder_roundtrip(oracle, w, style):
    rng = np.random.default_rng(w * 7 + len(style))
    n = 3000
    hi = (1 << w) if w else 1
    vals = rng.integers(0, hi, n, dtype=np.int64)
    # runs of repeats so RLE has something to do
    rep = rng.integers(0, n, 20)
    for r in rep:
        vals[r:r + int(rng.integers(2, 100))] = vals[r]
    enc = pg.hybrid_encode(vals, w, rng, style)
    out, rc, used = oracle.read_rle_bit_packed_hybrid(enc, w, len(enc))
    assert rc == 0 and used == len(enc)
    assert len(out) >= n and np.array_equal(out[:n], vals)
    assert not out[n:].any()


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
@pytest.mark.parametrize("trim", [False, True])
def test_delta_builder_roundtrip(oracle, dtype, trim):
    rng = np.random.default_rng(5)
    info = np.iinfo(dtype)
    for n in (1, 2, 31, 32, 33, 128, 129, 130, 1000):
        for spread in (0, 1, 1000, 1 << 20, None):
            if spread is None:
                vals = rng.integers(info.min, info.max, n, dtype=dtype, endpoint=True)
            else:
                vals = np.cumsum(rng.integers(-spread, spread + 1, n)).astype(dtype)
            enc = pg.delta_encode(vals, dtype, trim=trim)
            rd = oracle.read_delta_binary_packed_int32 if dtype == np.int32 else oracle.read_delta_binary_packed_int64
            out, rc, used = rd(enc + b"\x55\x55")
            assert rc == 0 and used == len(enc), (n, spread, trim)
            assert np.array_equal(out, vals)
            # (full-range data: parquet-go sign-extends the 32-bit zigzag to a 10-byte varint and sizes miniblock
            #  widths from a signed maximum; the spec encoder does neither, so only the bytes of the
            #  small-delta cases are compared)
            if not trim and spread is not None:
                ref = (oracle.write_delta_int32 if dtype == np.int32 else oracle.write_delta_int64)(vals)
                assert ref == enc, "numpy builder differs from the restated parquet-go writer"
                assert pg.delta_encode_fast(vals, dtype) == enc


def test_chunk_dict_int32_required(oracle):
    rng = np.random.default_rng(1)
    dict_vals = rng.integers(-2**31, 2**31 - 1, 300, dtype=np.int64).astype(np.int32)
    cb = pg.ChunkBuilder(gap=3, lead=5)
    cb.dict_page(dict_vals.tobytes(), len(dict_vals))
    expect = []
    for n in (1, 7, 8, 1000, 5000):
        idx = rng.integers(0, len(dict_vals), n)
        cb.data_page_v1(pg.ENC_RLE_DICTIONARY, n, pg.dict_indices_body(idx, 9, rng))
        expect.append(dict_vals[idx])
    chunk, pages = cb.finish()
    t = oracle.read_chunk(chunk, pages, oracle.INT32)
    assert t.status == 0
    exp = np.concatenate(expect)
    assert t.n_slots == len(exp) and t.n_rows == len(exp) and t.n_nulls == 0
    assert np.array_equal(t.values.view("<i4").reshape(-1), exp)
    assert not t.def_levels.any() and not t.rep_levels.any()


def test_chunk_optional_plain_int64_v1_v2(oracle):
    rng = np.random.default_rng(2)
    cb = pg.ChunkBuilder()
    exp_vals, exp_def = [], []
    for v2 in (False, True):
        n = 777
        dl = (rng.random(n) < 0.8).astype(np.int64)
        vals = rng.integers(-2**62, 2**62, int(dl.sum()), dtype=np.int64)
        if v2:
            cb.data_page_v2(pg.ENC_PLAIN, n, vals.tobytes(), b"", pg.hybrid_encode(dl, 1, rng))
        else:
            cb.data_page_v1(pg.ENC_PLAIN, n, vals.tobytes(), b"", pg.level_block_v1(dl, 1, rng))
        slot = np.zeros(n, np.int64)
        slot[dl == 1] = vals
        exp_vals.append(slot)
        exp_def.append(dl)
    chunk, pages = cb.finish()
    t = oracle.read_chunk(chunk, pages, oracle.INT64, max_def=1)
    assert t.status == 0
    assert np.array_equal(t.def_levels, np.concatenate(exp_def))
    assert np.array_equal(t.values.view("<i8").reshape(-1), np.concatenate(exp_vals))
    assert t.n_nulls == int((np.concatenate(exp_def) == 0).sum())


def test_chunk_error_classes(oracle):
    # dictionary index out of range -> the reference panics
    cb = pg.ChunkBuilder()
    cb.dict_page(np.arange(4, dtype=np.int32).tobytes(), 4)
    cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, 8, pg.dict_indices_body(np.array([0, 1, 2, 3, 4, 0, 0, 0]), 3, style="bp"))
    chunk, pages = cb.finish()
    t = oracle.read_chunk(chunk, pages, oracle.INT32)
    assert t.status == oracle.PANIC and t.bad_page == 1
    # PLAIN page too short -> ErrUnexpectedEOF
    cb = pg.ChunkBuilder()
    cb.data_page_v1(pg.ENC_PLAIN, 10, b"\0" * 39)
    chunk, pages = cb.finish()
    assert oracle.read_chunk(chunk, pages, oracle.INT32).status == oracle.UNEXPECTED_EOF
    # fewer indices than values -> slice past the decoded values
    cb = pg.ChunkBuilder()
    cb.dict_page(np.arange(4, dtype=np.int32).tobytes(), 4)
    cb.data_page_v1(pg.ENC_PLAIN_DICTIONARY, 9, pg.dict_indices_body(np.zeros(8, np.int64), 2, style="bp"))
    chunk, pages = cb.finish()
    assert oracle.read_chunk(chunk, pages, oracle.INT32).status == oracle.PANIC
