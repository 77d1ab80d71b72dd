"""Cross-check on real Parquet files written by pyarrow (an independent implementation of the format).

The column chunks of UNCOMPRESSED files are cut out using the footer (read with pyarrow), the page headers are
walked by libpqdecode's own thrift reader (pqd_scan_pages), and the pages are decoded by the CPU oracle and -
in the gpu-marked test - by the HIP kernels.  Expected values come from pyarrow's reader.  This pins the
oracle on everything the reference's own tests never cover: bit-packed runs, minimal index widths, multi-block
deltas, v2 pages, nulls under every encoding, nested lists.
"""
import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
pq_ = pytest.importorskip("pyarrow.parquet")

PHYS = {"BOOLEAN": 0, "INT32": 1, "INT64": 2, "INT96": 3, "FLOAT": 4, "DOUBLE": 5, "BYTE_ARRAY": 6, "FIXED_LEN_BYTE_ARRAY": 7}
NP_OF = {"INT32": "<i4", "INT64": "<i8", "FLOAT": "<f4", "DOUBLE": "<f8"}


def make_table(n, rng, nulls):
    def mask(a):
        if not nulls:
            return pa.array(a)
        m = rng.random(len(a)) < 0.2
        return pa.array(a, mask=m)

    keys = sorted("key-%08d-%s" % (int(x), "x" * int(x % 7)) for x in rng.integers(0, 10**7, n))
    cols = {
        "i32_dict": mask(rng.integers(0, 500, n).astype(np.int32)),
        "i64_delta": mask((1_700_000_000_000_000 + np.cumsum(rng.integers(1000, 5000, n))).astype(np.int64)),
        "i32_delta": mask(np.cumsum(rng.integers(-50, 50, n)).astype(np.int32)),
        "f32_plain": mask(rng.random(n).astype(np.float32)),
        "f32_bss": mask(rng.standard_normal(n).astype(np.float32)),
        "f64_bss": mask(rng.standard_normal(n).astype(np.float64)),
        "f64_dict": mask(rng.integers(0, 50, n).astype(np.float64) / 7.0),
        "bool_plain": mask(rng.integers(0, 2, n).astype(bool)),
        "str_plain": mask(np.array(["s%d" % x for x in rng.integers(0, 10**6, n)], dtype=object)),
        "str_dict": mask(np.array(["word%d" % x for x in rng.integers(0, 300, n)], dtype=object)),
        "str_dlba": mask(np.array(["v" * int(x) for x in rng.integers(0, 30, n)], dtype=object)),
        "str_dba": mask(np.array(keys, dtype=object)),
        "flba": pa.array([bytes(rng.integers(0, 256, 6, dtype=np.uint8)) for _ in range(n)], type=pa.binary(6)),
        # FIXED_LEN_BYTE_ARRAY under DELTA_BYTE_ARRAY (layout/page.go:696-706): keys that share prefixes
        "flba_dba": pa.array([None if (nulls and rng.random() < 0.1) else b"key%07d" % (i // 3 * 7) for i in range(n)], type=pa.binary(10)),
    }
    lists = []
    for i in range(n):
        r = rng.random()
        if nulls and r < 0.1:
            lists.append(None)
        elif r < 0.2:
            lists.append([])
        else:
            lists.append([None if (nulls and rng.random() < 0.1) else int(v) for v in rng.integers(-10**9, 10**9, int(rng.integers(1, 6)))])
    cols["list_i64"] = pa.array(lists, type=pa.list_(pa.int64()))
    return pa.table(cols)


def write_file(path, table, version):
    pq_.write_table(
        table, path, compression="NONE", data_page_version=version, data_page_size=8 * 1024, write_statistics=False,
        use_dictionary=["i32_dict", "f64_dict", "str_dict", "list_i64"],
        column_encoding={"i64_delta": "DELTA_BINARY_PACKED", "i32_delta": "DELTA_BINARY_PACKED", "f32_plain": "PLAIN",
                         "f32_bss": "BYTE_STREAM_SPLIT", "f64_bss": "BYTE_STREAM_SPLIT",
                         "bool_plain": "PLAIN", "str_plain": "PLAIN", "str_dlba": "DELTA_LENGTH_BYTE_ARRAY",
                         "str_dba": "DELTA_BYTE_ARRAY", "flba": "PLAIN", "flba_dba": "DELTA_BYTE_ARRAY"},
        row_group_size=len(table) // 2 + 1)


def column_chunks(path):
    """-> list of (column name, physical type, type_length, max_def, max_rep, chunk bytes, num_values, row group, col idx)."""
    f = pq_.ParquetFile(path)
    raw = open(path, "rb").read()
    md = f.metadata
    out = []
    for rg in range(md.num_row_groups):
        for ci in range(md.num_columns):
            cm = md.row_group(rg).column(ci)
            sc = f.schema.column(ci)
            start = cm.dictionary_page_offset if cm.has_dictionary_page and cm.dictionary_page_offset else cm.data_page_offset
            chunk = raw[start:start + cm.total_compressed_size]
            out.append((cm.path_in_schema, cm.physical_type, sc.length or 0, sc.max_definition_level, sc.max_repetition_level,
                        chunk, cm.num_values, rg, ci))
    return f, out


def expected_leaf(f, rg, name):
    """Non-null leaf values of the column in row group rg, in order, plus the count of level slots is checked apart."""
    top = name.split(".")[0]
    col = f.read_row_group(rg, columns=[top]).column(0).combine_chunks()
    if pa.types.is_list(col.type):
        col = col.flatten()
    return col


def check_table(t, f, rg, name, ptype, max_def, num_values):
    assert t.status == 0, (name, t.status, t.bad_page)
    assert t.n_slots == num_values
    exp = expected_leaf(f, rg, name)
    valid = t.def_levels == max_def
    exp_valid = exp.drop_null()
    assert int(valid.sum()) == len(exp_valid), name
    if ptype in NP_OF:
        got = t.values.view(NP_OF[ptype]).reshape(-1)[valid]
        want = np.asarray(exp_valid.to_numpy(zero_copy_only=False)).astype(NP_OF[ptype])
        assert got.tobytes() == want.tobytes(), name
    elif ptype == "BOOLEAN":
        got = t.values.reshape(-1)[valid].astype(bool)
        assert np.array_equal(got, np.asarray(exp_valid.to_numpy(zero_copy_only=False)).astype(bool)), name
    elif ptype == "FIXED_LEN_BYTE_ARRAY":
        got = [bytes(r) for r in t.values[valid]]
        assert got == exp_valid.to_pylist(), name
    else:
        b = t.bytes.tobytes()
        got = [b[t.offsets[i]:t.offsets[i + 1]] for i in np.nonzero(valid)[0]]
        want = [x.encode() if isinstance(x, str) else x for x in exp_valid.to_pylist()]
        assert got == want, name
        # null slots are empty
        lens = np.diff(t.offsets)
        assert not lens[~valid].any()


@pytest.fixture(scope="module", params=[("1.0", False), ("1.0", True), ("2.0", True)], ids=["v1", "v1-nulls", "v2-nulls"])
def parquet_file(request, tmp_path_factory):
    version, nulls = request.param
    rng = np.random.default_rng(11 if nulls else 12)
    path = str(tmp_path_factory.mktemp("pq") / ("t_%s_%d.parquet" % (version, nulls)))
    write_file(path, make_table(6000, rng, nulls), version)
    return path


def test_oracle_matches_pyarrow(parquet_file, oracle):
    import parquet_go_amd as pq
    f, chunks = column_chunks(parquet_file)
    seen_enc = set()
    for name, ptype, tl, max_def, max_rep, chunk, num_values, rg, ci in chunks:
        pages = pq.scan_pages(chunk, num_values)
        assert len(pages) >= 1
        seen_enc.update(int(e) for e in pages["encoding"])
        t = oracle.read_chunk(chunk, pages, PHYS[ptype], tl, max_def, max_rep)
        check_table(t, f, rg, name, ptype, max_def, num_values)
        # rows: one per slot with repetition level 0
        assert t.n_rows == f.metadata.row_group(rg).num_rows
    assert {0, 5, 6, 7, 9}.issubset(seen_enc) and (2 in seen_enc or 8 in seen_enc)


def test_footer_walk_matches_pyarrow(parquet_file):
    """The C++ footer walk against pyarrow's view of the same footer: leaves, levels, chunk offsets and counts."""
    import parquet_go_amd as pq
    raw = open(parquet_file, "rb").read()
    foot = pq.read_footer(raw)
    f = pq_.ParquetFile(parquet_file)
    md = f.metadata
    assert foot["num_rows"] == md.num_rows and foot["num_row_groups"] == md.num_row_groups
    assert len(foot["leaves"]) == md.num_columns
    for ci, leaf in enumerate(foot["leaves"]):
        sc = f.schema.column(ci)
        assert leaf["path"] == sc.path
        assert leaf["physical_type"] == PHYS[sc.physical_type]
        assert (leaf["max_def"], leaf["max_rep"]) == (sc.max_definition_level, sc.max_repetition_level)
        assert leaf["type_length"] == (sc.length if sc.physical_type == "FIXED_LEN_BYTE_ARRAY" else leaf["type_length"])
    assert len(foot["chunks"]) == md.num_row_groups * md.num_columns
    k = 0
    for rg in range(md.num_row_groups):
        for ci in range(md.num_columns):
            cm = md.row_group(rg).column(ci)
            ch = foot["chunks"][k]
            k += 1
            start = cm.dictionary_page_offset if cm.has_dictionary_page and cm.dictionary_page_offset else cm.data_page_offset
            assert (ch["row_group"], ch["leaf"], ch["codec"]) == (rg, ci, 0)
            assert (ch["offset"], ch["total_compressed_size"], ch["num_values"]) == (start, cm.total_compressed_size, cm.num_values)
            assert ch["num_rows"] == md.row_group(rg).num_rows


def test_footer_walk_rejects_garbage():
    import parquet_go_amd as pq
    with pytest.raises(pq.DecodeError):
        pq.read_footer(b"PAR1" + b"\x00" * 20 + b"\x05\x00\x00\x00NOPE")
    with pytest.raises(pq.DecodeError):
        pq.read_footer(b"PAR1" + b"\xff" * 40 + b"\x28\x00\x00\x00PAR1")


@pytest.mark.gpu
def test_gpu_reads_whole_file_through_the_footer(parquet_file, oracle):
    """End to end on the library's own host code: footer walk -> chunk bytes -> page walk -> GPU decode."""
    import parquet_go_amd as pq
    raw = open(parquet_file, "rb").read()
    foot = pq.read_footer(raw)
    f = pq_.ParquetFile(parquet_file)
    dec = pq.Decoder(0)
    names = {v: k for k, v in PHYS.items()}
    for ch in foot["chunks"]:
        leaf = foot["leaves"][ch["leaf"]]
        chunk = raw[ch["offset"]:ch["offset"] + ch["total_compressed_size"]]
        pages = pq.scan_pages(chunk, ch["num_values"])
        g = dec.decode(chunk, pages, leaf["physical_type"], leaf["type_length"], leaf["max_def"], leaf["max_rep"])
        check_table(g, f, ch["row_group"], leaf["path"], names[leaf["physical_type"]], leaf["max_def"], ch["num_values"])
        assert g.n_rows == ch["num_rows"]
    dec.close()


@pytest.mark.gpu
def test_gpu_matches_pyarrow_and_oracle(parquet_file, oracle):
    import parquet_go_amd as pq
    dec = pq.Decoder(0)
    f, chunks = column_chunks(parquet_file)
    for name, ptype, tl, max_def, max_rep, chunk, num_values, rg, ci in chunks:
        pages = pq.scan_pages(chunk, num_values)
        g = dec.decode(chunk, pages, PHYS[ptype], tl, max_def, max_rep)
        o = oracle.read_chunk(chunk, pages, PHYS[ptype], tl, max_def, max_rep)
        assert g.status == o.status == 0, (name, g.status, g.bad_page)
        assert np.array_equal(g.def_levels, o.def_levels) and np.array_equal(g.rep_levels, o.rep_levels), name
        assert (g.n_slots, g.n_rows, g.n_nulls) == (o.n_slots, o.n_rows, o.n_nulls)
        if o.kind == "fixed":
            assert np.array_equal(g.values, o.values), name
        else:
            assert np.array_equal(g.offsets, o.offsets) and np.array_equal(g.bytes, o.bytes), name
        check_table(g, f, rg, name, ptype, max_def, num_values)
    dec.close()


def _v2_bool_chunks(tmp_path, nulls):
    """A BOOLEAN column written by pyarrow with v2 data pages and no encoding named: pyarrow then picks Encoding_RLE
    for the values (bit width 1 in its reading of the format)."""
    rng = np.random.default_rng(5 + nulls)
    n = 20000
    vals = np.repeat(rng.integers(0, 2, n // 8 + 1).astype(bool), rng.integers(1, 16, n // 8 + 1))[:n]
    arr = pa.array(vals, mask=(rng.random(len(vals)) < 0.15) if nulls else None)
    path = str(tmp_path / ("b%d.parquet" % nulls))
    pq_.write_table(pa.table({"b": arr}), path, compression="NONE", data_page_version="2.0", data_page_size=4096,
                    write_statistics=False, use_dictionary=False)
    return column_chunks(path)[1]


def _check_v2_bool(decode, oracle, tmp_path):
    import parquet_go_amd as pq
    seen = set()
    for nulls in (0, 1):
        for name, ptype, tl, max_def, max_rep, chunk, num_values, rg, ci in _v2_bool_chunks(tmp_path, nulls):
            pages = pq.scan_pages(chunk, num_values)
            seen.update(int(e) for e in pages["encoding"])
            o = oracle.read_chunk(chunk, pages, PHYS[ptype], tl, max_def, max_rep)
            # The reference reads Encoding_RLE values with the column's type length as the bit width (layout/page.go:657-667),
            # which is 0 for BOOLEAN: the oracle says what comes of that, and the GPU path has to say the same
            if decode is not None:
                g = decode(chunk, pages, PHYS[ptype], tl, max_def, max_rep)
                assert (g.status, g.bad_page) == (o.status, o.bad_page)
                if o.status == 0:
                    assert (g.n_slots, g.n_rows, g.n_nulls) == (o.n_slots, o.n_rows, o.n_nulls)
                    assert np.array_equal(g.def_levels, o.def_levels) and np.array_equal(g.values, o.values)
    assert 3 in seen, "pyarrow did not write Encoding_RLE values: %r" % (seen,)


def test_oracle_reads_pyarrow_v2_default_bool(oracle, tmp_path):
    _check_v2_bool(None, oracle, tmp_path)


@pytest.mark.gpu
def test_gpu_matches_oracle_on_pyarrow_v2_default_bool(oracle, tmp_path):
    import parquet_go_amd as pq
    dec = pq.Decoder(0)
    try:
        _check_v2_bool(dec.decode, oracle, tmp_path)
    finally:
        dec.close()


def test_scan_pages_reports_capacity_and_the_wrapper_grows(parquet_file):
    """pqd_scan_pages with room for fewer descriptors than the chunk has pages: PQD_ERR_CAPACITY and the number of
    pages it got to; the Python wrapper retries with more room."""
    import ctypes
    import parquet_go_amd as pq
    f, chunks = column_chunks(parquet_file)
    name, ptype, tl, max_def, max_rep, chunk, num_values, rg, ci = max(chunks, key=lambda c: len(c[5]))
    pages = pq.scan_pages(chunk, num_values)
    assert len(pages) > 2
    lib = pq.lib()
    small = np.zeros(2, pq.PAGE_DESC_DTYPE)
    n = ctypes.c_int32(0)
    buf = (ctypes.c_uint8 * len(chunk)).from_buffer_copy(chunk)
    rc = lib.pqd_scan_pages(buf, ctypes.c_size_t(len(chunk)), ctypes.c_int64(num_values),
                            small.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(2), ctypes.byref(n))
    assert rc == pq.binding.ERR_CAPACITY
    assert np.array_equal(small, pages[:2])
