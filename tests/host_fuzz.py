"""Host parsers (page headers, footer) under ASan/UBSan on mutated pyarrow files.  CPU only."""
import ctypes as C, io, sys, os
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.environ["PQD_HOST_ASAN_LIB"])
from importlib import import_module
B = import_module("parquet_go_amd.binding")
L.pqd_footer_locate.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int64)]
L.pqd_footer_parse.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
L.pqd_footer_free.argtypes = [C.c_void_p]; L.pqd_footer_free.restype = None
for fn in (L.pqd_footer_num_row_groups, L.pqd_footer_num_leaves, L.pqd_footer_num_chunks):
    fn.argtypes, fn.restype = [C.c_void_p], C.c_int32
L.pqd_footer_leaf.argtypes = [C.c_void_p, C.c_int32, C.POINTER(B._LeafDesc)]
L.pqd_footer_chunk.argtypes = [C.c_void_p, C.c_int32, C.POINTER(B._ChunkDesc)]
L.pqd_scan_pages.argtypes = [C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]

def make_file(rng, k):
    n = int(rng.integers(1, 3000))
    cols = {"a": pa.array(rng.integers(0, 50, n), pa.int32()),
            "s": pa.array([None if i % 7 == 0 else "k%d" % (i % 13) for i in range(n)]),
            "l": pa.array([[int(x) for x in rng.integers(0, 9, int(rng.integers(0, 4)))] for _ in range(n)], pa.list_(pa.int64()))}
    t = pa.table(cols)
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="NONE", data_page_version="2.0" if k % 2 else "1.0", data_page_size=int(rng.integers(200, 4000)),
                   row_group_size=int(rng.integers(100, 2000)), use_dictionary=bool(k % 3))
    return bytearray(buf.getvalue())

def exercise(data):
    arr = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    off = C.c_int64(0)
    rc = L.pqd_footer_locate(arr, len(data), C.byref(off))
    if rc != 0: return
    f = C.c_void_p()
    rc = L.pqd_footer_parse(C.byref(arr, off.value), len(data) - 8 - off.value, C.byref(f))
    if rc != 0: return
    nl, nc = L.pqd_footer_num_leaves(f), L.pqd_footer_num_chunks(f)
    L.pqd_footer_num_row_groups(f)
    for i in range(-1, nl + 1):
        d = B._LeafDesc(); L.pqd_footer_leaf(f, i, C.byref(d))
    descs = np.zeros(4096, B.PAGE_DESC_DTYPE)
    for i in range(-1, nc + 1):
        d = B._ChunkDesc()
        if L.pqd_footer_chunk(f, i, C.byref(d)) == 0 and 0 <= d.offset < len(data):
            ln = min(max(d.total_compressed_size, 0), len(data) - d.offset)
            n = C.c_int32(0)
            L.pqd_scan_pages(C.byref(arr, d.offset), ln, d.num_values, C.c_void_p(descs.ctypes.data), descs.size, C.byref(n))
    L.pqd_footer_free(f)

n_cases = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for k in range(n_cases):
    rng = np.random.default_rng(seed + k)
    data = make_file(rng, k)
    exercise(data)                       # the intact file
    for _ in range(20):
        m = bytearray(data)
        foot = int.from_bytes(m[-8:-4], "little")
        for _ in range(int(rng.integers(1, 6))):
            if rng.integers(0, 2):      # inside the footer
                pos = len(m) - 8 - int(rng.integers(0, max(foot, 1)))
            else:                       # anywhere (page headers among it)
                pos = int(rng.integers(0, len(m)))
            m[max(pos, 0)] = int(rng.integers(0, 256))
        if rng.integers(0, 10) == 0: m = m[:int(rng.integers(0, len(m)))]
        if len(m) >= 12: exercise(m)
print("host fuzz ok:", n_cases, "files x 21")
