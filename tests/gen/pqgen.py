"""ctypes wrapper of tests/gen/pqgen.c (synthetic chunk generator; benchmark / test infrastructure)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PAGE_DESC_DTYPE = np.dtype([
    ("page_type", "<i4"), ("encoding", "<i4"), ("num_values", "<i4"), ("rep_len", "<i4"),
    ("def_len", "<i4"), ("reserved", "<i4"), ("offset", "<i8"), ("length", "<i8")])


def build():
    so = os.path.join(_HERE, "libpqgen.so")
    src = os.path.join(_HERE, "pqgen.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libpqgen.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.pqgen_dict_slot_bytes.restype = C.c_int64
        L.pqgen_dict_slot_bytes.argtypes = [C.c_int64, C.c_int]
        L.pqgen_dict_slot_bytes2.restype = C.c_int64
        L.pqgen_dict_slot_bytes2.argtypes = [C.c_int64, C.c_int, C.c_int]
        L.pqgen_delta_slot_bytes.restype = C.c_int64
        L.pqgen_delta_slot_bytes.argtypes = [C.c_int64, C.c_int]
        L.pqgen_dict_chunk.restype = C.c_int32
        L.pqgen_dict_chunk.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int32, C.c_int64, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        L.pqgen_delta_chunk.restype = C.c_int32
        L.pqgen_delta_chunk.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        L.pqgen_list_slot_bytes.restype = C.c_int64
        L.pqgen_list_slot_bytes.argtypes = [C.c_int64]
        L.pqgen_list_chunk.restype = C.c_int32
        L.pqgen_list_chunk.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        L.pqgen_strings_slot_bytes.restype = C.c_int64
        L.pqgen_strings_slot_bytes.argtypes = [C.c_int64]
        L.pqgen_strings_chunk.restype = C.c_int32
        L.pqgen_strings_chunk.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        _LIB = L
    return _LIB


def dict_chunk(seed, n_values, w, dict_n, page_values, style=0, rep_max=4, out=None):
    """-> (chunk uint8 array, page descriptors).  RLE_DICTIONARY INT32 chunk: dictionary page + v1 data pages."""
    n_pages = (n_values + page_values - 1) // page_values
    cap = (dict_n * 4 + 255) // 256 * 256 + n_pages * lib().pqgen_dict_slot_bytes2(page_values, w, style) + 64
    buf = out if out is not None else np.zeros(cap, np.uint8)
    assert buf.size >= cap
    pages = np.zeros(n_pages + 1, PAGE_DESC_DTYPE)
    n = lib().pqgen_dict_chunk(seed, n_values, w, dict_n, page_values, style, rep_max, buf.ctypes.data, buf.size,
                               pages.ctypes.data, pages.size)
    assert n == n_pages + 1, n
    return buf[:cap], pages


def delta_chunk(seed, n_values, w, page_values, elem_size=8, start=1_700_000_000_000_000, base_step=1000, out=None):
    """-> (chunk uint8 array, page descriptors).  DELTA_BINARY_PACKED chunk of v1 data pages."""
    n_pages = (n_values + page_values - 1) // page_values
    cap = n_pages * lib().pqgen_delta_slot_bytes(page_values, w) + 64
    buf = out if out is not None else np.zeros(cap, np.uint8)
    assert buf.size >= cap
    pages = np.zeros(n_pages, PAGE_DESC_DTYPE)
    n = lib().pqgen_delta_chunk(seed, n_values, w, page_values, elem_size, start, base_step, buf.ctypes.data, buf.size,
                                pages.ctypes.data, pages.size)
    assert n == n_pages, n
    return buf[:cap], pages


def list_chunk(seed, n_slots, page_slots):
    """-> (chunk, pages): optional list<optional int64> column chunk (max_def 3, max_rep 1), PLAIN INT64 values."""
    n_pages = (n_slots + page_slots - 1) // page_slots
    cap = n_pages * lib().pqgen_list_slot_bytes(page_slots) + 64
    buf = np.zeros(cap, np.uint8)
    pages = np.zeros(n_pages, PAGE_DESC_DTYPE)
    n = lib().pqgen_list_chunk(seed, n_slots, page_slots, buf.ctypes.data, buf.size, pages.ctypes.data, pages.size)
    assert n == n_pages, n
    return buf, pages


def strings_chunk(seed, n_values, page_values, mode):
    """-> (chunk, pages): BYTE_ARRAY chunk of key-like strings.  mode 0 DELTA_LENGTH_BYTE_ARRAY, 1 DELTA_BYTE_ARRAY, 2 PLAIN."""
    n_pages = (n_values + page_values - 1) // page_values
    cap = n_pages * lib().pqgen_strings_slot_bytes(page_values) + 64
    buf = np.zeros(cap, np.uint8)
    pages = np.zeros(n_pages, PAGE_DESC_DTYPE)
    n = lib().pqgen_strings_chunk(seed, n_values, page_values, mode, buf.ctypes.data, buf.size, pages.ctypes.data, pages.size)
    assert n == n_pages, n
    return buf, pages
