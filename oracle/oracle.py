"""ctypes binding of the CPU oracle (oracle/pq_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py; the product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, EOF, UNEXPECTED_EOF, PANIC, UNSUPPORTED, NOMEM, UNSUPPORTED_TYPE, FLBA_LENGTH = 0, -1, -2, -3, -4, -5, -6, -7

BOOLEAN, INT32, INT64, INT96, FLOAT, DOUBLE, BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY = range(8)
ENC_PLAIN, ENC_PLAIN_DICTIONARY, ENC_RLE, ENC_BIT_PACKED = 0, 2, 3, 4
ENC_DELTA_BINARY_PACKED, ENC_DELTA_LENGTH_BYTE_ARRAY, ENC_DELTA_BYTE_ARRAY = 5, 6, 7
ENC_RLE_DICTIONARY, ENC_BYTE_STREAM_SPLIT = 8, 9
DATA_PAGE, INDEX_PAGE, DICTIONARY_PAGE, DATA_PAGE_V2 = 0, 1, 2, 3

PAGE_DESC_DTYPE = np.dtype([
    ("page_type", "<i4"), ("encoding", "<i4"), ("num_values", "<i4"), ("rep_len", "<i4"),
    ("def_len", "<i4"), ("reserved", "<i4"), ("offset", "<i8"), ("length", "<i8")])
assert PAGE_DESC_DTYPE.itemsize == 40


class _Rd(C.Structure):
    _fields_ = [("s", C.c_void_p), ("n", C.c_int64), ("i", C.c_int64)]


class _I64v(C.Structure):
    _fields_ = [("v", C.POINTER(C.c_int64)), ("n", C.c_int64), ("cap", C.c_int64)]


class _Buf(C.Structure):
    _fields_ = [("b", C.POINTER(C.c_uint8)), ("n", C.c_int64), ("cap", C.c_int64)]


class _Col(C.Structure):
    _fields_ = [("kind", C.c_int32), ("elem_size", C.c_int32), ("n", C.c_int64),
                ("fixed", _Buf), ("off", _I64v), ("bytes", _Buf)]


class _ColDesc(C.Structure):
    _fields_ = [("physical_type", C.c_int32), ("type_length", C.c_int32),
                ("max_def", C.c_int32), ("max_rep", C.c_int32)]


class _Table(C.Structure):
    _fields_ = [("n_slots", C.c_int64), ("n_rows", C.c_int64), ("n_nulls", C.c_int64),
                ("def_", C.POINTER(C.c_int32)), ("rep", C.POINTER(C.c_int32)),
                ("values", _Col), ("status", C.c_int32), ("bad_page", C.c_int32)]


def build(force=False):
    """Compile liboracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "pq_oracle.c")
    hdr = os.path.join(_HERE, "pq_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_read_chunk.restype = C.c_int
    return _LIB


def _as_u8(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    return a


def _rd(a):
    return _Rd(a.ctypes.data if a.size else None, a.size, 0)


def _take_i64(v):
    out = np.ctypeslib.as_array(v.v, shape=(v.n,)).copy() if v.n else np.zeros(0, np.int64)
    lib().orc_i64v_free(C.byref(v))
    return out


def _take_buf(b):
    out = np.ctypeslib.as_array(b.b, shape=(b.n,)).copy() if b.n else np.zeros(0, np.uint8)
    lib().orc_buf_free(C.byref(b))
    return out


def _col_to_py(c, free=True):
    """-> ('fixed', ndarray uint8 [n, elem]) or ('bytes', offsets int64[n+1], bytes uint8)"""
    n = c.n
    if c.kind == 0:
        raw = np.ctypeslib.as_array(c.fixed.b, shape=(n * c.elem_size,)).copy() if n * c.elem_size else np.zeros(0, np.uint8)
        res = ("fixed", raw.reshape(n, c.elem_size) if c.elem_size else raw.reshape(n, 0), None)
    else:
        off = np.ctypeslib.as_array(c.off.v, shape=(n + 1,)).copy()
        nb = int(off[n])
        by = np.ctypeslib.as_array(c.bytes.b, shape=(nb,)).copy() if nb else np.zeros(0, np.uint8)
        res = ("bytes", off, by)
    if free:
        lib().orc_col_free(C.byref(c))
    return res


def col_strings(col):
    """list of bytes objects from a ('bytes', off, by) triple."""
    _, off, by = col
    b = by.tobytes()
    return [b[off[i]:off[i + 1]] for i in range(len(off) - 1)]


# ---- decoders -----------------------------------------------------------
def read_unsigned_varint(data):
    a = _as_u8(data); r = _rd(a); out = C.c_uint64(0)
    lib().orc_read_unsigned_varint(C.byref(r), C.byref(out))
    return out.value, r.i


def _i64_call(fn, data, *args):
    a = _as_u8(data); r = _rd(a); v = _I64v()
    rc = fn(C.byref(r), *args, C.byref(v))
    return _take_i64(v), rc, r.i


def read_rle(data, header, bit_width):
    return _i64_call(lib().orc_read_rle, data, C.c_uint64(header), C.c_uint64(bit_width))


def read_bit_packed(data, header, bit_width):
    return _i64_call(lib().orc_read_bit_packed, data, C.c_uint64(header), C.c_uint64(bit_width))


def read_rle_bit_packed_hybrid(data, bit_width, length):
    return _i64_call(lib().orc_read_rle_bit_packed_hybrid, data, C.c_uint64(bit_width), C.c_uint64(length))


def read_delta_binary_packed_int32(data):
    v, rc, pos = _i64_call(lib().orc_read_delta_binary_packed_int32, data)
    return v.astype(np.int32), rc, pos


def read_delta_binary_packed_int64(data):
    return _i64_call(lib().orc_read_delta_binary_packed_int64, data)


def _col_call(fn, data, *args):
    a = _as_u8(data); r = _rd(a); c = _Col()
    rc = fn(C.byref(r), *args, C.byref(c))
    return _col_to_py(c), rc, r.i


def read_delta_length_byte_array(data):
    return _col_call(lib().orc_read_delta_length_byte_array, data)


def read_delta_byte_array(data):
    return _col_call(lib().orc_read_delta_byte_array, data)


def read_plain(data, data_type, cnt, bit_width=0):
    return _col_call(lib().orc_read_plain, data, C.c_int32(data_type), C.c_uint64(cnt), C.c_uint64(bit_width))


def read_byte_stream_split(data, data_type, cnt):
    return _col_call(lib().orc_read_byte_stream_split, data, C.c_int32(data_type), C.c_uint64(cnt))


class Table:
    """Typed image of layout.Table for one column chunk."""

    def __init__(self):
        self.status = 0
        self.bad_page = -1
        self.n_slots = self.n_rows = self.n_nulls = 0
        self.def_levels = self.rep_levels = None
        self.kind = None        # 'fixed' | 'bytes'
        self.values = None      # fixed: uint8 [n_slots, elem]
        self.offsets = None     # bytes: int64 [n_slots + 1]
        self.bytes = None


def read_chunk(chunk, pages, physical_type, type_length=0, max_def=0, max_rep=0):
    a = _as_u8(chunk)
    pg = np.ascontiguousarray(pages, dtype=PAGE_DESC_DTYPE)
    cd = _ColDesc(physical_type, type_length, max_def, max_rep)
    t = _Table()
    rc = lib().orc_read_chunk(C.c_void_p(a.ctypes.data if a.size else None), C.c_int64(a.size),
                              C.c_void_p(pg.ctypes.data if pg.size else None), C.c_int32(pg.size),
                              C.byref(cd), C.byref(t))
    out = Table()
    out.status, out.bad_page = rc, t.bad_page
    if rc == OK:
        n = t.n_slots
        out.n_slots, out.n_rows, out.n_nulls = n, t.n_rows, t.n_nulls
        out.def_levels = np.ctypeslib.as_array(t.def_, shape=(n,)).copy() if n else np.zeros(0, np.int32)
        out.rep_levels = np.ctypeslib.as_array(t.rep, shape=(n,)).copy() if n else np.zeros(0, np.int32)
        kind, x, y = _col_to_py(t.values, free=False)
        out.kind = kind
        if kind == "fixed":
            out.values = x
        else:
            out.offsets, out.bytes = x, y
    lib().orc_table_free(C.byref(t))
    return out


def table_pop(rep_levels, num_rows):
    """layout/table.go:73-103: number of leading slots Table.Pop(numRows) hands out (the loop, literally)."""
    num = -1
    i = 0
    ln = len(rep_levels)
    while i < ln:
        if rep_levels[i] == 0:
            num += 1
            if num >= num_rows:
                break
        i += 1
    return i


class ColumnBufferModel:
    """reader/columnbuffer.go:105-156, :232-262 over oracle tables, one ReadPage batch per chunk: DataTable as growing
    arrays, DataTableNumRows one below the row starts seen until the end of the file, ReadRows = Pop."""

    def __init__(self, tables):
        self.tables = list(tables)      # decoded chunks (Table), in order
        self.k = 0
        self.rep = np.zeros(0, np.int32)
        self.dl = np.zeros(0, np.int32)
        self.num_rows = -1              # columnbuffer.go:44
        self.eof = False

    def _read(self):
        if self.k >= len(self.tables):
            self.eof = True
            self.num_rows += 1          # columnbuffer.go:58, on every call past the last row group: a ReadRows on a
            return False                # drained column therefore reports one row and hands out an empty table
        t = self.tables[self.k]
        self.k += 1
        self.rep = np.concatenate([self.rep, t.rep_levels])
        self.dl = np.concatenate([self.dl, t.def_levels])
        self.num_rows += t.n_rows
        return True

    def read_rows(self, num):
        """-> (slots handed out, rows handed out)"""
        while self.num_rows < num and self._read():
            pass
        if self.num_rows < 0:
            self.num_rows = 0
        num = min(num, self.num_rows)
        end = table_pop(self.rep, num)
        self.rep, self.dl = self.rep[end:], self.dl[end:]
        self.num_rows -= num
        return end, num


    def skip_rows(self, num):
        """reader/columnbuffer.go:194-230 as seen from outside: the rows SkipRows drops are the rows ReadRows would
        have handed out (it reads the same pages, levels first, and pops the same table); -> (slots, rows) dropped."""
        return self.read_rows(num)


def read_chunk_timed(chunk, pages, physical_type, type_length=0, max_def=0, max_rep=0):
    """Decode and discard (no copies back to numpy): used for CPU-baseline timing.  -> (rc, n_slots)"""
    a = _as_u8(chunk)
    pg = np.ascontiguousarray(pages, dtype=PAGE_DESC_DTYPE)
    cd = _ColDesc(physical_type, type_length, max_def, max_rep)
    t = _Table()
    rc = lib().orc_read_chunk(C.c_void_p(a.ctypes.data), C.c_int64(a.size), C.c_void_p(pg.ctypes.data),
                              C.c_int32(pg.size), C.byref(cd), C.byref(t))
    n = t.n_slots
    lib().orc_table_free(C.byref(t))
    return rc, n


# ---- encoders -----------------------------------------------------------
def _enc(fn, *args):
    b = _Buf()
    fn(*args, C.byref(b))
    return _take_buf(b).tobytes()


def write_unsigned_varint(num):
    return _enc(lib().orc_write_unsigned_varint, C.c_uint64(num))


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def write_rle_int64(vals, bit_width):
    a = np.ascontiguousarray(vals, dtype=np.int64)
    return _enc(lib().orc_write_rle_int64, _i64p(a), C.c_int64(a.size), C.c_int32(bit_width))


def write_rle_int32(vals, bit_width):
    a = np.ascontiguousarray(vals, dtype=np.int32)
    return _enc(lib().orc_write_rle_int32, a.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(a.size), C.c_int32(bit_width))


def write_rle_bit_packed_hybrid_int64(vals, bit_width):
    a = np.ascontiguousarray(vals, dtype=np.int64)
    return _enc(lib().orc_write_rle_bit_packed_hybrid_int64, _i64p(a), C.c_int64(a.size), C.c_int32(bit_width))


def write_rle_bit_packed_hybrid_int32(vals, bit_width):
    a = np.ascontiguousarray(vals, dtype=np.int32)
    return _enc(lib().orc_write_rle_bit_packed_hybrid_int32, a.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(a.size), C.c_int32(bit_width))


def write_bit_packed(vals, bit_width, if_header):
    a = np.ascontiguousarray(vals, dtype=np.int64)
    return _enc(lib().orc_write_bit_packed, _i64p(a), C.c_int64(a.size), C.c_int64(bit_width), C.c_int(1 if if_header else 0))


def write_delta_int32(nums):
    a = np.ascontiguousarray(nums, dtype=np.int32)
    return _enc(lib().orc_write_delta_int32, a.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(a.size))


def write_delta_int64(nums):
    a = np.ascontiguousarray(nums, dtype=np.int64)
    return _enc(lib().orc_write_delta_int64, _i64p(a), C.c_int64(a.size))


def _pack_strings(strings):
    off = np.zeros(len(strings) + 1, np.int64)
    np.cumsum([len(s) for s in strings], out=off[1:])
    by = np.frombuffer(b"".join(strings) + b"\0", dtype=np.uint8).copy()
    return off, by


def write_delta_length_byte_array(strings):
    off, by = _pack_strings(strings)
    return _enc(lib().orc_write_delta_length_byte_array, _i64p(off), by.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(strings)))


def write_delta_byte_array(strings):
    off, by = _pack_strings(strings)
    return _enc(lib().orc_write_delta_byte_array, _i64p(off), by.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(strings)))


def write_plain_boolean(vals):
    a = np.ascontiguousarray(vals, dtype=np.uint8)
    return _enc(lib().orc_write_plain_boolean, a.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(a.size))


def write_plain_byte_array(strings):
    off, by = _pack_strings(strings)
    return _enc(lib().orc_write_plain_byte_array, _i64p(off), by.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(strings)))
