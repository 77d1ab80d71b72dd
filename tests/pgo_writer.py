"""Test-side writer of Parquet files laid out the way the reference's own writer lays them out (flat schemas).

What is reproduced, from reading the reference: page cuts by accumulated element size (layout/page.go:71-136,
layout/dictpage.go:74-152: a page is closed once its values reach PageSize), v1 data pages `rep | def | values`
with length-prefixed RLE-only level blocks (layout/page.go:194-281 -> encoding/encodingwrite.go:149-214),
PLAIN_DICTIONARY index pages as `[bit width = 32][RLE runs]` with the dictionary page in front of the chunk's data
pages (layout/dictpage.go:155-225, writer/writer.go:323-324), page headers and the footer in thrift compact protocol,
`PAR1` framing (writer/writer.go:142-225).  The encoders are the oracle's restatements of encodingwrite.go.

The reference itself cannot run here (no Go toolchain), so byte-for-byte identity with a file it would write is not
claimed: statistics are left out of the page headers and the footer carries only what a reader needs.
"""
import struct

import numpy as np

BOOLEAN, INT32, INT64, INT96, FLOAT, DOUBLE, BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY = range(8)
ENC_PLAIN, ENC_PLAIN_DICTIONARY, ENC_RLE = 0, 2, 3
REQUIRED, OPTIONAL, REPEATED = 0, 1, 2


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _zz(v):
    return (v << 1) ^ (v >> 63) if v >= 0 else ((-v) << 1) - 1


class Thrift(object):
    """Writer of the thrift compact protocol (structs of i32 / i64 / binary / list / struct fields)."""

    def __init__(self):
        self.b = bytearray()
        self.last = [0]

    def _field(self, fid, ftype):
        delta = fid - self.last[-1]
        if 0 < delta <= 15:
            self.b.append((delta << 4) | ftype)
        else:
            self.b.append(ftype)
            self.b += _varint(_zz(fid))
        self.last[-1] = fid

    def i32(self, fid, v):
        self._field(fid, 5)
        self.b += _varint(_zz(int(v)))

    def i64(self, fid, v):
        self._field(fid, 6)
        self.b += _varint(_zz(int(v)))

    def binary(self, fid, data):
        self._field(fid, 8)
        self.b += _varint(len(data)) + bytes(data)

    def begin(self, fid):
        self._field(fid, 12)
        self.last.append(0)

    def end(self):
        self.b.append(0)
        self.last.pop()

    def list_begin(self, fid, etype, n):
        self._field(fid, 9)
        if n < 15:
            self.b.append((n << 4) | etype)
        else:
            self.b.append(0xF0 | etype)
            self.b += _varint(n)

    def elem_begin(self):
        self.last.append(0)

    def elem_i32(self, v):
        self.b += _varint(_zz(int(v)))

    def elem_binary(self, data):
        self.b += _varint(len(data)) + bytes(data)


def page_header(page_type, size, num_values, encoding):
    """parquet.PageHeader (parquet/parquet.go:6329-6338) with a DataPageHeader or DictionaryPageHeader."""
    t = Thrift()
    t.i32(1, page_type)
    t.i32(2, size)
    t.i32(3, size)
    if page_type == 0:
        t.begin(5)
        t.i32(1, num_values)
        t.i32(2, encoding)
        t.i32(3, ENC_RLE)
        t.i32(4, ENC_RLE)
        t.end()
    else:
        t.begin(7)
        t.i32(1, num_values)
        t.i32(2, encoding)
        t.end()
    t.b.append(0)
    return bytes(t.b)


def _elem_sizes(col):
    pt = col["type"]
    if pt == BYTE_ARRAY:
        return [len(v) for v in col["values"]]
    return [{BOOLEAN: 1, INT32: 4, INT64: 8, FLOAT: 4, DOUBLE: 8, INT96: 12}.get(pt, col.get("type_length", 0))] * len(col["values"])


def _plain(orc, pt, vals):
    if pt == BYTE_ARRAY:
        return orc.write_plain_byte_array(list(vals))
    if pt == BOOLEAN:
        return orc.write_plain_boolean(np.asarray(vals, np.uint8))
    if pt in (INT96, FIXED_LEN_BYTE_ARRAY):
        return b"".join(vals)
    dt = {INT32: "<i4", INT64: "<i8", FLOAT: "<f4", DOUBLE: "<f8"}[pt]
    return np.asarray(vals, dtype=dt).tobytes()


def _column_chunk(orc, col, lo, hi, page_size):
    """Pages of rows [lo, hi) of one flat column -> chunk bytes, and whether a dictionary page leads them.
    col: name, type, values (non-null values of all rows, in order), optional 'valid' (bool per row: OPTIONAL
    column), 'encoding'."""
    pt = col["type"]
    valid = col.get("valid")
    n_rows = hi - lo
    if valid is None:
        v_lo, v_hi = lo, hi
        defs = None
    else:
        valid = np.asarray(valid, bool)
        v_lo, v_hi = int(valid[:lo].sum()), int(valid[:hi].sum())
        defs = valid[lo:hi].astype(np.int32)
    vals = col["values"][v_lo:v_hi]
    sizes = _elem_sizes({"type": pt, "values": vals, "type_length": col.get("type_length", 0)})
    dict_enc = col.get("encoding", ENC_PLAIN) == ENC_PLAIN_DICTIONARY
    out = bytearray()
    dict_slice, dict_map = [], {}
    pages = []
    i = 0        # row
    vi = 0       # value
    while i < n_rows:
        j, size, v0 = i, 0, vi
        while j < n_rows and size < page_size:       # layout/page.go:89, layout/dictpage.go:94
            if defs is None or defs[j] == 1:
                size += sizes[vi]
                vi += 1
            j += 1
        pvals = vals[v0:vi]
        levels = b"" if defs is None else orc.write_rle_bit_packed_hybrid_int32(defs[i:j], 1)
        if dict_enc:
            idx = []
            for v in pvals:
                key = bytes(v) if pt in (BYTE_ARRAY, INT96, FIXED_LEN_BYTE_ARRAY) else v
                if key not in dict_map:
                    dict_map[key] = len(dict_slice)
                    dict_slice.append(v)
                idx.append(dict_map[key])
            body = levels + bytes([32]) + orc.write_rle_int32(np.asarray(idx, np.int32), 32)
            enc = ENC_PLAIN_DICTIONARY
        else:
            body = levels + _plain(orc, pt, pvals)
            enc = ENC_PLAIN
        pages.append(page_header(0, len(body), j - i, enc) + body)
        i = j
    if dict_enc:
        dbody = _plain(orc, pt, dict_slice)
        out += page_header(2, len(dbody), len(dict_slice), ENC_PLAIN)
        out += dbody
    first_data = len(out)
    for p in pages:
        out += p
    return bytes(out), dict_enc, first_data, len(pages)


def write_file(orc, columns, n_rows, rows_per_group=None, page_size=8192):
    """-> (file bytes, info): info['chunks'][row group][column] = (file offset, length, data pages)."""
    rows_per_group = rows_per_group or max(n_rows, 1)
    f = bytearray(b"PAR1")
    groups = []
    for lo in range(0, max(n_rows, 1), rows_per_group):
        hi = min(n_rows, lo + rows_per_group)
        cols = []
        for col in columns:
            chunk, has_dict, first_data, n_pages = _column_chunk(orc, col, lo, hi, page_size)
            cols.append({"offset": len(f), "len": len(chunk), "dict": has_dict, "data_off": len(f) + first_data, "n_pages": n_pages,
                         "num_values": hi - lo})
            f += chunk
        groups.append({"rows": hi - lo, "cols": cols})
        if n_rows == 0:
            break
    t = Thrift()
    t.i32(1, 1)                                   # version (writer/writer.go:75)
    t.list_begin(2, 12, 1 + len(columns))         # schema
    t.elem_begin()
    t.i32(3, REQUIRED)
    t.binary(4, b"Parquet_go_root")               # schema/schemahandler.go:247
    t.i32(5, len(columns))
    t.end()
    for col in columns:
        t.elem_begin()
        t.i32(1, col["type"])
        if col["type"] == FIXED_LEN_BYTE_ARRAY:
            t.i32(2, col["type_length"])
        t.i32(3, OPTIONAL if col.get("valid") is not None else REQUIRED)
        t.binary(4, col["name"].encode())
        if "converted_type" in col:
            t.i32(6, col["converted_type"])
        t.end()
    t.i64(3, n_rows)
    t.list_begin(4, 12, len(groups))
    for g in groups:
        t.elem_begin()
        t.list_begin(1, 12, len(columns))
        for col, c in zip(columns, g["cols"]):
            t.elem_begin()
            t.i64(2, c["offset"])
            t.begin(3)                            # ColumnMetaData (parquet/parquet.go:7451-7466)
            t.i32(1, col["type"])
            encs = [ENC_RLE, ENC_PLAIN] + ([ENC_PLAIN_DICTIONARY] if c["dict"] else [])
            t.list_begin(2, 5, len(encs))
            for e in encs:
                t.elem_i32(e)
            t.list_begin(3, 8, 1)
            t.elem_binary(col["name"].encode())
            t.i32(4, 0)                           # UNCOMPRESSED
            t.i64(5, c["num_values"])
            t.i64(6, c["len"])
            t.i64(7, c["len"])
            t.i64(9, c["data_off"])
            if c["dict"]:
                t.i64(11, c["offset"])
            t.end()
            t.end()
        t.i64(2, sum(c["len"] for c in g["cols"]))
        t.i64(3, g["rows"])
        t.end()
    t.binary(6, b"parquet-go version latest")
    t.b.append(0)
    f += t.b
    f += struct.pack("<I", len(t.b)) + b"PAR1"
    return bytes(f), {"groups": groups}


def local_flat_columns(num=100, day=19000):
    """The six columns example/local_flat.go:42-50 writes."""
    i = np.arange(num)
    return [
        {"name": "name", "type": BYTE_ARRAY, "values": [b"StudentName"] * num, "encoding": ENC_PLAIN_DICTIONARY, "converted_type": 0},
        {"name": "age", "type": INT32, "values": (20 + i % 5).astype(np.int32)},
        {"name": "id", "type": INT64, "values": i.astype(np.int64)},
        {"name": "weight", "type": FLOAT, "values": (np.float32(50.0) + i.astype(np.float32) * np.float32(0.1)).astype(np.float32)},
        {"name": "sex", "type": BOOLEAN, "values": (i % 2 == 0).astype(np.uint8)},
        {"name": "day", "type": INT32, "values": np.full(num, day, np.int32), "converted_type": 6},
    ]
