"""The shared library loads without a GPU and exports every function include/pqdecode.h declares (no compute calls)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pqdecode.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(pqd_[a-z_0-9]+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported():
    import parquet_go_amd as pq
    names = declared_functions()
    assert len(names) >= 15 and "pqd_decode_chunk_device" in names
    lib = ctypes.CDLL(pq.lib_path())
    for n in names:
        assert hasattr(lib, n), n


def test_struct_sizes_match_the_binding():
    import parquet_go_amd as pq
    assert pq.PAGE_DESC_DTYPE.itemsize == 40
    assert pq.PAGE_DESC_DTYPE.fields["offset"][1] == 24 and pq.PAGE_DESC_DTYPE.fields["length"][1] == 32


def test_host_only_calls_work_without_a_gpu():
    import parquet_go_amd as pq
    assert pq.lib().pqd_version()
    assert b"EOF" in pq.lib().pqd_strerror(-1).upper() or pq.strerror(-1)
    # the thrift page walk is host code: a hand-assembled compact-protocol PageHeader (parquet.thrift PageHeader:
    # 1 type, 2 uncompressed_page_size, 3 compressed_page_size, 5 data_page_header{1 num_values, 2 encoding,
    # 3 definition_level_encoding, 4 repetition_level_encoding}) followed by a 10-byte body
    header = bytes([0x15, 0x00, 0x15, 0x14, 0x15, 0x14, 0x2C, 0x15, 0x0A, 0x15, 0x00, 0x15, 0x06, 0x15, 0x06, 0x00, 0x00])
    chunk = header + bytes(range(10))
    got = pq.scan_pages(chunk, 5)
    assert len(got) == 1
    g = got[0]
    assert (g["page_type"], g["encoding"], g["num_values"], g["offset"], g["length"]) == (0, 0, 5, len(header), 10)
