"""Pins the CPU oracle against the reference's own known-answer vectors.

Vectors: tests/golden/reference_kat.json, transcribed from
encoding/encodingwrite_test.go (literal bytes) and encoding/encodingread_test.go
(round-trip inputs; these go through the oracle's restated encoders, exactly as
the reference's tests go through the package's own writers).
"""
import json
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kat.json")))


def test_varint_kat(oracle):
    for c in KAT["varint"]["cases"]:
        assert oracle.write_unsigned_varint(c["num"]) == bytes(c["bytes"])
        v, used = oracle.read_unsigned_varint(bytes(c["bytes"]))
        assert (v, used) == (c["num"], len(c["bytes"]))
    # all ten concatenated, as the reference test builds them
    blob = b"".join(bytes(c["bytes"]) for c in KAT["varint"]["cases"])
    pos = 0
    for c in KAT["varint"]["cases"]:
        v, used = oracle.read_unsigned_varint(blob[pos:])
        assert v == c["num"]
        pos += used
    assert pos == len(blob)


def test_varint_roundtrip(oracle):
    for n in KAT["varint_roundtrip"]["nums"]:
        v, _ = oracle.read_unsigned_varint(oracle.write_unsigned_varint(n))
        assert v == n


def test_varint_never_errors_on_truncation(oracle):
    # encodingread.go:132 shadows err: a cut-off varint yields the partial value
    v, used = oracle.read_unsigned_varint(bytes([0x80, 0x80]))
    assert (v, used) == (0, 2)
    v, used = oracle.read_unsigned_varint(b"")
    assert (v, used) == (0, 0)


def test_rle_kat(oracle):
    for c in KAT["rle"]["cases"]:
        enc = oracle.write_rle_int64(c["nums"], c["bit_width"])
        assert enc == bytes(c["bytes"])
        # decode direction: a hybrid stream of exactly these runs
        vals, rc, used = oracle.read_rle_bit_packed_hybrid(enc, c["bit_width"], len(enc))
        assert rc == 0 and used == len(enc)
        assert vals.tolist() == c["nums"]


def test_bit_packed_kat(oracle):
    for c in KAT["bit_packed"]["cases"]:
        enc = oracle.write_bit_packed(c["nums"], c["bit_width"], True)
        assert enc == bytes(c["bytes"])
        vals, rc, used = oracle.read_bit_packed(enc[1:], enc[0], c["bit_width"])
        assert rc == 0 and vals.tolist() == c["nums"] and used == len(enc) - 1
        vals, rc, _ = oracle.read_rle_bit_packed_hybrid(enc, c["bit_width"], len(enc))
        assert rc == 0 and vals.tolist() == c["nums"]


def test_bit_packed_roundtrip(oracle):
    for c in KAT["bit_packed_roundtrip"]["cases"]:
        ln = len(c["nums"])
        header = ((ln // 8) << 1) | 1
        enc = oracle.write_bit_packed(c["nums"], c["bit_width"], False)
        vals, rc, _ = oracle.read_bit_packed(enc, header, c["bit_width"])
        assert rc == 0 and vals.tolist() == c["nums"]


def test_hybrid_roundtrip(oracle):
    for c in KAT["hybrid_roundtrip"]["cases"]:
        enc = oracle.write_rle_bit_packed_hybrid_int64(c["nums"], c["bit_width"])
        vals, rc, used = oracle.read_rle_bit_packed_hybrid(enc, c["bit_width"], 0)
        assert rc == 0 and used == len(enc) and vals.tolist() == c["nums"]


def test_plain_boolean(oracle):
    for c in KAT["plain_boolean"]["cases"]:
        assert oracle.write_plain_boolean(c["vals"]) == bytes(c["bytes"])
        if c["vals"]:
            (_, v, _), rc, _ = oracle.read_plain(bytes(c["bytes"]), oracle.BOOLEAN, len(c["vals"]))
            assert rc == 0 and v[:, 0].tolist() == c["vals"]
    for vals in KAT["plain_boolean_roundtrip"]["cases"]:
        (_, v, _), rc, _ = oracle.read_plain(oracle.write_plain_boolean(vals), oracle.BOOLEAN, len(vals))
        assert rc == 0 and v[:, 0].tolist() == vals


def test_plain_fixed(oracle):
    for key, typ, fmt in (("plain_int32", 1, "<i4"), ("plain_int64", 2, "<i8")):
        for c in KAT[key]["cases"]:
            (_, v, _), rc, used = oracle.read_plain(bytes(c["bytes"]), typ, len(c["vals"]))
            assert rc == 0 and used == len(c["bytes"])
            assert v.view(fmt).reshape(-1).tolist() == c["vals"]
    for c in KAT["plain_int96"]["cases"]:
        (_, v, _), rc, _ = oracle.read_plain(bytes(c["bytes"]), oracle.INT96, len(c["vals"]))
        assert rc == 0 and v.tolist() == c["vals"]
    for vals in KAT["plain_float_roundtrip"]["cases"]:
        raw = np.asarray(vals, dtype="<f4").tobytes()
        (_, v, _), rc, _ = oracle.read_plain(raw, oracle.FLOAT, len(vals))
        assert rc == 0 and v.tobytes() == raw
    for vals in KAT["plain_double_roundtrip"]["cases"]:
        raw = np.asarray(vals, dtype="<f8").tobytes()
        (_, v, _), rc, _ = oracle.read_plain(raw, oracle.DOUBLE, len(vals))
        assert rc == 0 and v.tobytes() == raw


def test_plain_fixed_short_buffer(oracle):
    # io.ReadFull: nothing there -> EOF, fewer than needed -> ErrUnexpectedEOF (binaryread.go:12-18)
    _, rc, _ = oracle.read_plain(b"", oracle.INT32, 1)
    assert rc == oracle.EOF
    _, rc, _ = oracle.read_plain(b"\x01\x02\x03", oracle.INT32, 1)
    assert rc == oracle.UNEXPECTED_EOF
    _, rc, _ = oracle.read_plain(b"", oracle.INT32, 0)
    assert rc == 0


def test_plain_byte_arrays(oracle):
    for c in KAT["plain_byte_array"]["cases"]:
        vals = [s.encode() for s in c["vals"]]
        assert oracle.write_plain_byte_array(vals) == bytes(c["bytes"])
        col, rc, _ = oracle.read_plain(bytes(c["bytes"]), oracle.BYTE_ARRAY, len(vals))
        assert rc == 0 and oracle.col_strings(col) == vals
    for vals in KAT["plain_byte_array_roundtrip"]["cases"]:
        vals = [s.encode() for s in vals]
        col, rc, _ = oracle.read_plain(oracle.write_plain_byte_array(vals), oracle.BYTE_ARRAY, len(vals))
        assert rc == 0 and oracle.col_strings(col) == vals
    for c in KAT["plain_flba"]["cases"]:
        (_, v, _), rc, _ = oracle.read_plain(bytes(c["bytes"]), oracle.FIXED_LEN_BYTE_ARRAY, len(c["vals"]), c["len"])
        assert rc == 0 and [bytes(r) for r in v] == [s.encode() for s in c["vals"]]
    for vals in KAT["plain_flba_roundtrip"]["cases"]:
        raw = "".join(vals).encode()
        (_, v, _), rc, _ = oracle.read_plain(raw, oracle.FIXED_LEN_BYTE_ARRAY, len(vals), len(vals[0]))
        assert rc == 0 and [bytes(r) for r in v] == [s.encode() for s in vals]


def test_delta_kat(oracle):
    for c in KAT["delta_int32"]["cases"]:
        assert oracle.write_delta_int32(c["nums"]) == bytes(c["bytes"])
        v, rc, used = oracle.read_delta_binary_packed_int32(bytes(c["bytes"]))
        assert rc == 0 and v.tolist() == c["nums"]
    for c in KAT["delta_int64"]["cases"]:
        assert oracle.write_delta_int64(c["nums"]) == bytes(c["bytes"])
        v, rc, used = oracle.read_delta_binary_packed_int64(bytes(c["bytes"]))
        assert rc == 0 and v.tolist() == c["nums"]


def test_delta_roundtrip_wrap(oracle):
    for nums in KAT["delta_int64_roundtrip"]["cases"]:
        v, rc, _ = oracle.read_delta_binary_packed_int64(oracle.write_delta_int64(nums))
        assert rc == 0 and v.tolist() == nums
    for nums in KAT["delta_int32_roundtrip"]["cases"]:
        v, rc, _ = oracle.read_delta_binary_packed_int32(oracle.write_delta_int32(nums))
        assert rc == 0 and v.tolist() == nums


def test_zigzag32(oracle):
    # encodingread_test.go:169-188 — zigzag through a varint, decoded with the INT32 rule:
    # a one-value INT32 delta stream carries exactly that first value
    for n in KAT["zigzag32_roundtrip"]["nums"]:
        v, rc, _ = oracle.read_delta_binary_packed_int32(oracle.write_delta_int32([n]))
        assert rc == 0 and v.tolist() == [n]


def test_delta_unneeded_miniblocks_not_consumed(oracle):
    # {1..5}: one block, only miniblock 0 is read (encodingread.go:372); the cursor stops after it
    enc = bytes(KAT["delta_int64"]["cases"][0]["bytes"])
    v, rc, used = oracle.read_delta_binary_packed_int64(enc + b"\xAA\xBB")
    assert rc == 0 and used == len(enc)
    enc2 = bytes(KAT["delta_int64"]["cases"][1]["bytes"])   # widths 2,0,0,0: 8 payload bytes then nothing
    v, rc, used = oracle.read_delta_binary_packed_int64(enc2 + b"\xAA")
    assert rc == 0 and used == len(enc2)


def test_delta_byte_arrays(oracle):
    for key, rd, wr in (("delta_length_byte_array", "read_delta_length_byte_array", "write_delta_length_byte_array"),
                        ("delta_byte_array", "read_delta_byte_array", "write_delta_byte_array")):
        for c in KAT[key]["cases"]:
            vals = [s.encode() for s in c["vals"]]
            assert getattr(oracle, wr)(vals) == bytes(c["bytes"])
            col, rc, used = getattr(oracle, rd)(bytes(c["bytes"]))
            assert rc == 0 and used == len(c["bytes"]) and oracle.col_strings(col) == vals
    for vals in KAT["delta_byte_array_roundtrip"]["cases"]:
        vals = [s.encode() for s in vals]
        col, rc, _ = oracle.read_delta_byte_array(oracle.write_delta_byte_array(vals))
        assert rc == 0 and oracle.col_strings(col) == vals
        col, rc, _ = oracle.read_delta_length_byte_array(oracle.write_delta_length_byte_array(vals))
        assert rc == 0 and oracle.col_strings(col) == vals


def test_hybrid_quirks(oracle):
    # RLE value is the low 32 bits of the padded bytes and is NOT masked to the width (encodingread.go:157-160)
    vals, rc, _ = oracle.read_rle_bit_packed_hybrid(bytes([4, 0xFF]), 2, 2)
    assert rc == 0 and vals.tolist() == [255, 255]
    # bit-packed run cut short by the end of the buffer is zero-filled when at least one byte is left (:189)
    vals, rc, _ = oracle.read_rle_bit_packed_hybrid(bytes([3, 0x88]), 3, 2)
    assert rc == 0 and vals.tolist() == [0, 1, 2, 0, 0, 0, 0, 0]
    # ... and an EOF error when nothing is left (:189-191)
    vals, rc, _ = oracle.read_rle_bit_packed_hybrid(bytes([3]), 3, 1)
    assert rc == oracle.EOF
    # width 0 bit-packed consumes nothing (:182-187)
    vals, rc, used = oracle.read_rle_bit_packed_hybrid(bytes([3, 3]), 0, 2)
    assert rc == 0 and vals.tolist() == [0] * 16 and used == 2
    # 4-byte length prefix when length == 0 (:233-239)
    vals, rc, used = oracle.read_rle_bit_packed_hybrid(struct.pack("<i", 2) + bytes([6, 1, 99]), 1, 0)
    assert rc == 0 and vals.tolist() == [1, 1, 1] and used == 6


def test_flba_column_under_delta_byte_array_encodings(oracle):
    """layout/page.go:684-706 hands delta-decoded strings on for a FIXED_LEN_BYTE_ARRAY column too.  The reference's
    own byte vectors (encodingwrite_test.go:256-286: "Hello", "World", "Foobar", "ABCDEF") as pages of such a column:
    with type_length 5 the first two strings fill the fixed-width image, and "Foobar" has no place in it."""
    import pagegen as pg
    for name, enc in (("delta_length_byte_array", pg.ENC_DELTA_LENGTH_BYTE_ARRAY), ("delta_byte_array", pg.ENC_DELTA_BYTE_ARRAY)):
        body = bytes(KAT[name]["cases"][0]["bytes"])
        cb = pg.ChunkBuilder()
        cb.data_page_v1(enc, 2, body)     # values[:cnt] with cnt = 2
        chunk, pages = cb.finish()
        t = oracle.read_chunk(chunk, pages, oracle.FIXED_LEN_BYTE_ARRAY, 5)
        assert t.status == 0 and t.kind == "fixed"
        assert [bytes(r) for r in t.values] == [b"Hello", b"World"]
        cb = pg.ChunkBuilder()
        cb.data_page_v1(enc, 4, body)
        chunk, pages = cb.finish()
        t = oracle.read_chunk(chunk, pages, oracle.FIXED_LEN_BYTE_ARRAY, 5)
        assert t.status == oracle.FLBA_LENGTH and t.bad_page == 0
        t = oracle.read_chunk(chunk, pages, oracle.BYTE_ARRAY)
        assert t.status == 0 and t.bytes.tobytes() == b"HelloWorldFoobarABCDEF"
