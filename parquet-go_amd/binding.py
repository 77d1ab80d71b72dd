"""ctypes binding of libpqdecode.so (include/pqdecode.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BOOLEAN, INT32, INT64, INT96, FLOAT, DOUBLE, BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY = range(8)
ENC_PLAIN, ENC_PLAIN_DICTIONARY, ENC_RLE, ENC_BIT_PACKED = 0, 2, 3, 4
ENC_DELTA_BINARY_PACKED, ENC_DELTA_LENGTH_BYTE_ARRAY, ENC_DELTA_BYTE_ARRAY = 5, 6, 7
ENC_RLE_DICTIONARY, ENC_BYTE_STREAM_SPLIT = 8, 9
DATA_PAGE, INDEX_PAGE, DICTIONARY_PAGE, DATA_PAGE_V2 = 0, 1, 2, 3

PAGE_DESC_DTYPE = np.dtype([
    ("page_type", "<i4"), ("encoding", "<i4"), ("num_values", "<i4"), ("rep_len", "<i4"),
    ("def_len", "<i4"), ("reserved", "<i4"), ("offset", "<i8"), ("length", "<i8")])
assert PAGE_DESC_DTYPE.itemsize == 40

EXPORTS = [
    "pqd_version", "pqd_last_error", "pqd_strerror", "pqd_device_count", "pqd_handle_create",
    "pqd_handle_destroy", "pqd_set_profiling", "pqd_scan_pages", "pqd_decode_chunk",
    "pqd_decode_chunk_device", "pqd_wait", "pqd_result_device", "pqd_copy_out", "pqd_page_counts",
    "pqd_kernel_times", "pqd_decode_levels", "pqd_decode_levels_device", "pqd_row_starts", "pqd_result_host",
    "pqd_input_buffer", "pqd_set_option", "pqd_decode_chunks", "pqd_decode_chunks_device",
    "pqd_footer_locate", "pqd_footer_parse", "pqd_footer_free", "pqd_footer_num_rows", "pqd_footer_num_row_groups",
    "pqd_footer_num_leaves", "pqd_footer_num_chunks", "pqd_footer_leaf", "pqd_footer_chunk",
    "pqd_decode_chunk_device_noalloc", "pqd_direct_streams",
]

OPT_PRESCAN, OPT_DIRECT_RLE, OPT_JUMP, OPT_PLAIN_SCATTER, OPT_SPLIT, OPT_POISON, OPT_DELTA_BLOCKS, OPT_FUSED, OPT_BYTE_OFFSETS, OPT_DEVICE_TABLES = range(10)
LEVELS_ONLY = 1
ERR_CAPACITY = -14


class _ColDesc(C.Structure):
    _fields_ = [("physical_type", C.c_int32), ("type_length", C.c_int32), ("max_def", C.c_int32), ("max_rep", C.c_int32)]


class _LeafDesc(C.Structure):
    _fields_ = [("physical_type", C.c_int32), ("type_length", C.c_int32), ("max_def", C.c_int32), ("max_rep", C.c_int32),
                ("schema_index", C.c_int32), ("path", C.c_char * 236)]


class _ChunkDesc(C.Structure):
    _fields_ = [("row_group", C.c_int32), ("leaf", C.c_int32), ("codec", C.c_int32), ("reserved", C.c_int32),
                ("offset", C.c_int64), ("total_compressed_size", C.c_int64), ("num_values", C.c_int64), ("num_rows", C.c_int64)]


class _ChunkResult(C.Structure):
    _fields_ = [("num_slots", C.c_int64), ("num_values", C.c_int64), ("num_nulls", C.c_int64), ("num_rows", C.c_int64),
                ("bytes_len", C.c_int64), ("elem_size", C.c_int32), ("status", C.c_int32), ("bad_page", C.c_int32),
                ("has_levels", C.c_int32)]


class _DeviceResult(C.Structure):
    _fields_ = [("values", C.c_void_p), ("def_levels", C.c_void_p), ("rep_levels", C.c_void_p),
                ("offsets", C.c_void_p), ("bytes", C.c_void_p)]


class _KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("launches", C.c_int32)]


class DecodeError(RuntimeError):
    def __init__(self, code, page=-1, msg=""):
        self.code, self.page = code, page
        super().__init__("pqdecode error %d (%s)%s%s" % (code, strerror(code), " at page %d" % page if page >= 0 else "",
                                                      ": " + msg if msg else ""))


def lib_path():
    """The in-tree build; PQD_LIB points tests and benches at another build of the same library."""
    return os.environ.get("PQD_LIB") or os.path.join(_HERE, "lib", "libpqdecode.so")


def lib():
    """Loads libpqdecode.so; raises if it has not been built (no fallback path exists)."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError("libpqdecode.so is missing at %s: build it with `make -C parquet-go_amd/csrc` "
                              "(or __graft_entry__.build())" % path)
        L = C.CDLL(path)
        L.pqd_version.restype = C.c_char_p
        L.pqd_last_error.restype = C.c_char_p
        L.pqd_strerror.restype = C.c_char_p
        L.pqd_strerror.argtypes = [C.c_int]
        L.pqd_handle_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.pqd_handle_destroy.argtypes = [C.c_void_p]
        L.pqd_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.pqd_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.pqd_direct_streams.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        for fn in (L.pqd_decode_chunks, L.pqd_decode_chunks_device):
            fn.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_ColDesc), C.c_int32]
        L.pqd_scan_pages.argtypes = [C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.pqd_decode_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.POINTER(_ColDesc)]
        L.pqd_decode_chunk_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.POINTER(_ColDesc)]
        L.pqd_decode_levels.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.POINTER(_ColDesc)]
        L.pqd_decode_levels_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.POINTER(_ColDesc)]
        L.pqd_decode_chunk_device_noalloc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.POINTER(_ColDesc),
                                                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(_ChunkResult)]
        L.pqd_footer_locate.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int64)]
        L.pqd_footer_parse.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.pqd_footer_free.argtypes = [C.c_void_p]
        L.pqd_footer_free.restype = None
        for fn in (L.pqd_footer_num_row_groups, L.pqd_footer_num_leaves, L.pqd_footer_num_chunks):
            fn.argtypes, fn.restype = [C.c_void_p], C.c_int32
        L.pqd_footer_num_rows.argtypes, L.pqd_footer_num_rows.restype = [C.c_void_p], C.c_int64
        L.pqd_footer_leaf.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_LeafDesc)]
        L.pqd_footer_chunk.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_ChunkDesc)]
        L.pqd_row_starts.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.pqd_wait.argtypes = [C.c_void_p, C.POINTER(_ChunkResult)]
        L.pqd_result_device.argtypes = [C.c_void_p, C.POINTER(_DeviceResult)]
        L.pqd_result_host.argtypes = [C.c_void_p, C.POINTER(_DeviceResult)]
        L.pqd_input_buffer.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.pqd_copy_out.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pqd_page_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.pqd_kernel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        _LIB = L
    return _LIB


def strerror(code):
    return lib().pqd_strerror(int(code)).decode()


def _check(rc):
    if rc != 0:
        raise DecodeError(rc, msg=lib().pqd_last_error().decode())


def scan_pages(chunk, total_values=-1, cap=None):
    """Host-side page-header walk -> structured array of page descriptors (PAGE_DESC_DTYPE)."""
    a = np.frombuffer(chunk, dtype=np.uint8) if not isinstance(chunk, np.ndarray) else np.ascontiguousarray(chunk, np.uint8)
    cap = cap or 64
    while True:
        out = np.zeros(cap, dtype=PAGE_DESC_DTYPE)
        n = C.c_int32(0)
        rc = lib().pqd_scan_pages(C.c_void_p(a.ctypes.data if a.size else None) if a.size else C.c_void_p(out.ctypes.data),
                                  a.size, total_values, C.c_void_p(out.ctypes.data), cap, C.byref(n))
        if rc == ERR_CAPACITY:      # more pages than descriptors: grow and walk again
            cap *= 4
            continue
        if rc != 0:
            raise DecodeError(rc, page=n.value)
        return out[:n.value].copy()


def read_footer(file_bytes):
    """Footer of a whole Parquet file held in memory -> dict(num_rows, num_row_groups, leaves=[...], chunks=[...]).
    Host code only (no GPU needed)."""
    a = np.frombuffer(file_bytes, dtype=np.uint8)
    if a.size < 12:
        raise DecodeError(-12)
    flen = C.c_int64(0)
    tail = a[-8:].copy()
    _check(lib().pqd_footer_locate(C.c_void_p(tail.ctypes.data), 8, C.byref(flen)))
    if flen.value < 0 or flen.value + 8 > a.size:
        raise DecodeError(-12)
    foot = a[a.size - 8 - flen.value:a.size - 8].copy()
    h = C.c_void_p()
    _check(lib().pqd_footer_parse(C.c_void_p(foot.ctypes.data), foot.size, C.byref(h)))
    try:
        out = {"num_rows": lib().pqd_footer_num_rows(h), "num_row_groups": lib().pqd_footer_num_row_groups(h), "leaves": [], "chunks": []}
        for i in range(lib().pqd_footer_num_leaves(h)):
            d = _LeafDesc()
            _check(lib().pqd_footer_leaf(h, i, C.byref(d)))
            out["leaves"].append({"path": d.path.decode(), "physical_type": d.physical_type, "type_length": d.type_length,
                                  "max_def": d.max_def, "max_rep": d.max_rep})
        for i in range(lib().pqd_footer_num_chunks(h)):
            d = _ChunkDesc()
            _check(lib().pqd_footer_chunk(h, i, C.byref(d)))
            out["chunks"].append({k: getattr(d, k) for k in ("row_group", "leaf", "codec", "offset", "total_compressed_size",
                                                             "num_values", "num_rows")})
    finally:
        lib().pqd_footer_free(h)
    return out


class Table:
    """Typed image of the reference's layout.Table for one column chunk (host copies)."""

    def __init__(self):
        self.status = 0
        self.bad_page = -1
        self.n_slots = self.n_rows = self.n_nulls = self.n_values = 0
        self.def_levels = self.rep_levels = None
        self.kind = None
        self.values = None    # uint8 [n_slots, elem_size]
        self.offsets = None
        self.bytes = None
        self.page_counts = None


def decode_chunk_device_noalloc(device, stream, dev_ptr, nbytes, pages, physical_type, type_length, max_def, max_rep,
                                values_ptr, values_bytes, def_ptr, rep_ptr, level_slots):
    """pqd_decode_chunk_device_noalloc: every buffer is the caller's device memory.  -> (return code, _ChunkResult)"""
    pg = np.ascontiguousarray(pages, dtype=PAGE_DESC_DTYPE)
    cd = _ColDesc(physical_type, type_length, max_def, max_rep)
    r = _ChunkResult()
    rc = lib().pqd_decode_chunk_device_noalloc(device, C.c_void_p(stream) if stream else None, C.c_void_p(dev_ptr), nbytes,
                                               C.c_void_p(pg.ctypes.data if pg.size else None), pg.size, C.byref(cd),
                                               C.c_void_p(values_ptr), values_bytes, C.c_void_p(def_ptr) if def_ptr else None,
                                               C.c_void_p(rep_ptr) if rep_ptr else None, level_slots, C.byref(r))
    return rc, r


class Decoder:
    """One pqd_handle: a HIP stream plus grow-only workspaces on one device."""

    def __init__(self, device=0, stream=None, profiling=False):
        self._h = C.c_void_p()
        _check(lib().pqd_handle_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)))
        if profiling:
            _check(lib().pqd_set_profiling(self._h, 1))
        self._pages = None
        self._keep = None

    def close(self):
        if self._h:
            lib().pqd_handle_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _col(physical_type, type_length, max_def, max_rep):
        return _ColDesc(physical_type, type_length, max_def, max_rep)

    def set_option(self, option, value):
        """Pins a kernel-variant choice of this handle (OPT_*; -1 hands it back to the library)."""
        _check(lib().pqd_set_option(self._h, option, value))

    def submit_chunks(self, chunks, pages_list, physical_type, type_length=0, max_def=0, max_rep=0, levels_only=False):
        """Several chunks of one leaf column in one call: chunks = list of host byte buffers, pages_list = their page
        descriptor arrays."""
        arrs = [np.frombuffer(c, dtype=np.uint8) if not isinstance(c, np.ndarray) else np.ascontiguousarray(c, np.uint8) for c in chunks]
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data if a.size else None for a in arrs])
        lens = (C.c_size_t * max(n, 1))(*[a.size for a in arrs])
        pg = np.concatenate([np.ascontiguousarray(p, dtype=PAGE_DESC_DTYPE) for p in pages_list]) if n else np.zeros(0, PAGE_DESC_DTYPE)
        per = (C.c_int32 * max(n, 1))(*[len(p) for p in pages_list])
        self._pages, self._keep = pg, arrs
        self._is_bytes = physical_type == BYTE_ARRAY
        cd = self._col(physical_type, type_length, max_def, max_rep)
        _check(lib().pqd_decode_chunks(self._h, n, ptrs, lens, C.c_void_p(pg.ctypes.data if pg.size else None), per, C.byref(cd),
                                       LEVELS_ONLY if levels_only else 0))

    def submit_chunks_device(self, dev_ptrs, nbytes, pages_list, physical_type, type_length=0, max_def=0, max_rep=0, levels_only=False):
        """The same with the chunk bytes already in device memory (16-byte aligned pointers)."""
        n = len(dev_ptrs)
        ptrs = (C.c_void_p * max(n, 1))(*dev_ptrs)
        lens = (C.c_size_t * max(n, 1))(*nbytes)
        pg = np.concatenate([np.ascontiguousarray(p, dtype=PAGE_DESC_DTYPE) for p in pages_list]) if n else np.zeros(0, PAGE_DESC_DTYPE)
        per = (C.c_int32 * max(n, 1))(*[len(p) for p in pages_list])
        self._pages = pg
        self._is_bytes = physical_type == BYTE_ARRAY
        cd = self._col(physical_type, type_length, max_def, max_rep)
        _check(lib().pqd_decode_chunks_device(self._h, n, ptrs, lens, C.c_void_p(pg.ctypes.data if pg.size else None), per, C.byref(cd),
                                              LEVELS_ONLY if levels_only else 0))

    def submit(self, chunk, pages, physical_type, type_length=0, max_def=0, max_rep=0):
        """Chunk bytes in host memory (bytes / numpy uint8)."""
        a = np.frombuffer(chunk, dtype=np.uint8) if not isinstance(chunk, np.ndarray) else np.ascontiguousarray(chunk, np.uint8)
        pg = np.ascontiguousarray(pages, dtype=PAGE_DESC_DTYPE)
        self._pages, self._keep = pg, a
        self._is_bytes = physical_type == BYTE_ARRAY
        cd = self._col(physical_type, type_length, max_def, max_rep)
        _check(lib().pqd_decode_chunk(self._h, C.c_void_p(a.ctypes.data if a.size else None), a.size,
                                      C.c_void_p(pg.ctypes.data if pg.size else None), pg.size, C.byref(cd)))

    def submit_device(self, dev_ptr, nbytes, pages, physical_type, type_length=0, max_def=0, max_rep=0):
        """Chunk bytes already in device memory (16-byte aligned pointer)."""
        pg = np.ascontiguousarray(pages, dtype=PAGE_DESC_DTYPE)
        self._pages = pg
        self._is_bytes = physical_type == BYTE_ARRAY
        cd = self._col(physical_type, type_length, max_def, max_rep)
        _check(lib().pqd_decode_chunk_device(self._h, C.c_void_p(dev_ptr), nbytes,
                                             C.c_void_p(pg.ctypes.data if pg.size else None), pg.size, C.byref(cd)))

    def wait(self, check=False):
        r = _ChunkResult()
        rc = lib().pqd_wait(self._h, C.byref(r))
        if rc in (-8, -9):
            _check(rc)
        if check and r.status != 0:
            raise DecodeError(r.status, r.bad_page)
        return r

    def device_result(self):
        d = _DeviceResult()
        _check(lib().pqd_result_device(self._h, C.byref(d)))
        return d

    def row_starts(self, r):
        """Slot index of every row start of the last decoded chunk + end marker (int64[num_rows + 1])."""
        out = np.zeros(r.num_rows + 1, np.int64)
        _check(lib().pqd_row_starts(self._h, C.c_void_p(out.ctypes.data), out.size, None))
        return out

    def direct_streams(self):
        """Data pages of the last call whose index stream was expanded straight from the page (a diagnostic)."""
        n = C.c_int32(0)
        _check(lib().pqd_direct_streams(self._h, C.byref(n)))
        return n.value

    def kernel_times(self):
        arr = (_KernelTime * 32)()
        n = C.c_int32(0)
        _check(lib().pqd_kernel_times(self._h, arr, 32, C.byref(n)))
        return {arr[i].name.decode(): (arr[i].ms, arr[i].launches) for i in range(n.value)}

    def fetch(self, r=None):
        """Wait and copy the typed table to host numpy arrays."""
        r = r or self.wait()
        t = Table()
        t.status, t.bad_page = r.status, r.bad_page
        if r.status != 0:
            return t
        n = r.num_slots
        t.n_slots, t.n_rows, t.n_nulls, t.n_values = n, r.num_rows, r.num_nulls, r.num_values
        t.def_levels = np.zeros(n, np.int32)
        t.rep_levels = np.zeros(n, np.int32)
        if r.elem_size > 0 or r.bytes_len == 0 and r.elem_size == 0 and False:
            pass
        if r.bytes_len >= 0 and r.elem_size == 0 and self._is_bytes:
            t.kind = "bytes"
            t.offsets = np.zeros(n + 1, np.int64)
            t.bytes = np.zeros(max(r.bytes_len, 1), np.uint8)
            _check(lib().pqd_copy_out(self._h, None, C.c_void_p(t.def_levels.ctypes.data), C.c_void_p(t.rep_levels.ctypes.data),
                                      C.c_void_p(t.offsets.ctypes.data), C.c_void_p(t.bytes.ctypes.data)))
            t.bytes = t.bytes[:r.bytes_len]
        else:
            t.kind = "fixed"
            vals = np.zeros(max(n * r.elem_size, 1), np.uint8)
            _check(lib().pqd_copy_out(self._h, C.c_void_p(vals.ctypes.data), C.c_void_p(t.def_levels.ctypes.data),
                                      C.c_void_p(t.rep_levels.ctypes.data), None, None))
            t.values = vals[:n * r.elem_size].reshape(n, r.elem_size)
        if self._pages is not None and self._pages.size:
            pc = np.zeros((self._pages.size, 3), np.int64)
            _check(lib().pqd_page_counts(self._h, C.c_void_p(pc.ctypes.data), self._pages.size))
            t.page_counts = pc
        return t

    def input_buffer(self, nbytes):
        """numpy view (uint8, nbytes) of the handle's page-locked input buffer: fill it and pass it to submit()."""
        p = C.c_void_p()
        _check(lib().pqd_input_buffer(self._h, nbytes, C.byref(p)))
        return np.frombuffer((C.c_uint8 * max(nbytes, 1)).from_address(p.value), dtype=np.uint8, count=nbytes)

    def fetch_pinned(self, r=None):
        """Wait and return the typed table as numpy views of the handle's page-locked result memory (no copy into
        pageable arrays).  The views are valid until this decoder's next decode call."""
        r = r or self.wait()
        t = Table()
        t.status, t.bad_page = r.status, r.bad_page
        if r.status != 0:
            return t
        n = r.num_slots
        t.n_slots, t.n_rows, t.n_nulls, t.n_values = n, r.num_rows, r.num_nulls, r.num_values
        d = _DeviceResult()
        _check(lib().pqd_result_host(self._h, C.byref(d)))

        def view(addr, count, dtype):
            if not addr or count == 0:
                return np.zeros(count, dtype)
            nbytes = count * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_uint8 * nbytes).from_address(addr), dtype=dtype, count=count)

        t.def_levels = view(d.def_levels, n, np.int32)
        t.rep_levels = view(d.rep_levels, n, np.int32)
        if self._is_bytes:
            t.kind = "bytes"
            t.offsets = view(d.offsets, n + 1, np.int64)
            t.bytes = view(d.bytes, r.bytes_len, np.uint8)
        else:
            t.kind = "fixed"
            t.values = view(d.values, n * r.elem_size, np.uint8).reshape(n, r.elem_size)
        return t

    _is_bytes = False

    def decode_levels(self, chunk, pages, physical_type, type_length=0, max_def=0, max_rep=0):
        """Skip path: levels and counts only -> Table with kind 'levels' (no values)."""
        a = np.frombuffer(chunk, dtype=np.uint8) if not isinstance(chunk, np.ndarray) else np.ascontiguousarray(chunk, np.uint8)
        pg = np.ascontiguousarray(pages, dtype=PAGE_DESC_DTYPE)
        self._pages, self._keep = pg, a
        cd = self._col(physical_type, type_length, max_def, max_rep)
        _check(lib().pqd_decode_levels(self._h, C.c_void_p(a.ctypes.data if a.size else None), a.size,
                                       C.c_void_p(pg.ctypes.data if pg.size else None), pg.size, C.byref(cd)))
        r = self.wait()
        t = Table()
        t.status, t.bad_page, t.kind = r.status, r.bad_page, "levels"
        if r.status != 0:
            return t
        n = r.num_slots
        t.n_slots, t.n_rows, t.n_nulls, t.n_values = n, r.num_rows, r.num_nulls, r.num_values
        t.def_levels = np.zeros(n, np.int32)
        t.rep_levels = np.zeros(n, np.int32)
        _check(lib().pqd_copy_out(self._h, None, C.c_void_p(t.def_levels.ctypes.data), C.c_void_p(t.rep_levels.ctypes.data), None, None))
        if pg.size:
            pc = np.zeros((pg.size, 3), np.int64)
            _check(lib().pqd_page_counts(self._h, C.c_void_p(pc.ctypes.data), pg.size))
            t.page_counts = pc
        return t

    def decode(self, chunk, pages, physical_type, type_length=0, max_def=0, max_rep=0):
        """submit + wait + fetch."""
        self._is_bytes = physical_type == BYTE_ARRAY
        self.submit(chunk, pages, physical_type, type_length, max_def, max_rep)
        return self.fetch()
