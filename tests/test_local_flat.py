"""BASELINE configs[0]: the file example/local_flat.go writes (six flat columns, 100 rows, PLAIN + PLAIN_DICTIONARY,
UNCOMPRESSED, 8 KiB pages), read back the way the example reads it: through the footer, column by column, skipping
and reading ten rows in turn (local_flat.go:75-85).  The reader side is parquet_go_amd.ColumnBuffer - the
statement-by-statement counterpart of reader/columnbuffer.go over the GPU-backed layout calls - checked against the
values the example wrote and against the oracle's model of the column buffer."""
import numpy as np
import pytest

import pgo_writer as pw

DT = {pw.INT32: "<i4", pw.INT64: "<i8", pw.FLOAT: "<f4", pw.DOUBLE: "<f8", pw.BOOLEAN: "u1"}


def chunk_bytes(file_bytes, c):
    return np.frombuffer(file_bytes, np.uint8)[c["offset"]:c["offset"] + c["total_compressed_size"]]


def expected_rows(col, lo, hi):
    if col["type"] == pw.BYTE_ARRAY:
        return list(col["values"][lo:hi])
    return np.asarray(col["values"][lo:hi])


def rows_of(slice_, col):
    if col["type"] == pw.BYTE_ARRAY:
        b = slice_.bytes.tobytes()
        return [b[slice_.offsets[i]:slice_.offsets[i + 1]] for i in range(len(slice_.offsets) - 1)]
    return slice_.values.reshape(-1).view(DT[col["type"]]) if slice_.values.size else np.zeros(0, DT[col["type"]])


def same_rows(got, want):
    return got == want if isinstance(want, list) else np.array_equal(got, want)


def test_footer_pages_and_oracle_on_the_local_flat_file(oracle):
    import parquet_go_amd as pq
    cols = pw.local_flat_columns()
    data, info = pw.write_file(oracle, cols, 100)
    ft = pq.read_footer(data)
    assert ft["num_rows"] == 100 and ft["num_row_groups"] == 1
    assert [l["path"] for l in ft["leaves"]] == ["name", "age", "id", "weight", "sex", "day"]
    assert [l["physical_type"] for l in ft["leaves"]] == [pw.BYTE_ARRAY, pw.INT32, pw.INT64, pw.FLOAT, pw.BOOLEAN, pw.INT32]
    assert all(l["max_def"] == 0 and l["max_rep"] == 0 for l in ft["leaves"])
    for c in ft["chunks"]:
        col = cols[c["leaf"]]
        chunk = chunk_bytes(data, c)
        pages = pq.scan_pages(chunk, c["num_values"])
        # the dictionary column: dictionary page, then index pages of width 32 made of RLE runs only
        if col["name"] == "name":
            assert [int(p["page_type"]) for p in pages] == [2, 0] and int(pages[1]["encoding"]) == pw.ENC_PLAIN_DICTIONARY
            body = chunk[pages[1]["offset"]:pages[1]["offset"] + pages[1]["length"]].tobytes()
            assert body == bytes([32]) + b"\xC8\x01" + b"\0\0\0\0"      # one run of 100 x index 0 (encodingwrite.go:183-205)
        else:
            assert [int(p["page_type"]) for p in pages] == [0]
        t = oracle.read_chunk(chunk, pages, col["type"])
        assert t.status == 0 and t.n_slots == t.n_rows == 100 and t.n_nulls == 0
        if col["type"] == pw.BYTE_ARRAY:
            assert oracle.col_strings(("bytes", t.offsets, t.bytes)) == col["values"]
        else:
            assert np.array_equal(t.values.reshape(-1).view(DT[col["type"]]), np.asarray(col["values"]))


@pytest.mark.gpu
def test_local_flat_read_back_skipping_and_reading_ten_rows_in_turn(oracle):
    import parquet_go_amd as pq
    cols = pw.local_flat_columns()
    data, info = pw.write_file(oracle, cols, 100)
    ft = pq.read_footer(data)
    dec = pq.Decoder(0)
    try:
        for c in ft["chunks"]:
            col, leaf = cols[c["leaf"]], ft["leaves"][c["leaf"]]
            cb = pq.ColumnBuffer(dec, [(chunk_bytes(data, c), None, c["num_values"])], leaf["physical_type"], leaf["type_length"],
                                 leaf["max_def"], leaf["max_rep"], footer_num_rows=ft["num_rows"])
            num = ft["num_rows"]
            for i in range(num // 10):                       # example/local_flat.go:75-85
                if i % 2 == 0:
                    assert cb.skip_rows(10) == 10
                    continue
                got, n = cb.read_rows(10)
                assert n == 10
                assert same_rows(rows_of(got, col), expected_rows(col, 10 * i, 10 * i + 10)), (col["name"], i)
            # nothing is left behind the last round.  (The count is 1, not 0: NextRowGroup bumps DataTableNumRows each
            # time it is called past the last row group, reader/columnbuffer.go:57-59, and ReadRows reports that.)
            got, n = cb.read_rows(10)
            assert n == 1 and got.def_levels.size == 0
    finally:
        dec.close()


def big_columns(rng, n):
    valid = rng.random(n) < 0.85
    k = int(valid.sum())
    words = [b"w%03d" % x for x in range(200)]
    return [
        {"name": "key", "type": pw.BYTE_ARRAY, "values": [words[x] for x in rng.integers(0, 200, n)], "encoding": pw.ENC_PLAIN_DICTIONARY},
        {"name": "opt_id", "type": pw.INT64, "values": rng.integers(-2**40, 2**40, k).astype(np.int64), "valid": valid},
        {"name": "tag", "type": pw.INT32, "values": rng.integers(0, 50, n).astype(np.int32), "encoding": pw.ENC_PLAIN_DICTIONARY},
        {"name": "opt_name", "type": pw.BYTE_ARRAY, "values": [bytes(rng.integers(97, 123, int(l), dtype=np.uint8)) for l in rng.integers(0, 30, k)],
         "valid": valid},
        {"name": "flag", "type": pw.BOOLEAN, "values": (rng.random(n) < 0.5).astype(np.uint8)},
        {"name": "x", "type": pw.DOUBLE, "values": rng.standard_normal(k), "valid": valid},
    ]


@pytest.mark.gpu
def test_reader_loop_over_row_groups_and_many_pages(oracle):
    # several row groups, many 8 KiB pages per chunk, optional columns, dictionary columns (a dictionary page in front
    # of every chunk); skips and reads of any length, across page and row-group boundaries, past the end of the file
    import parquet_go_amd as pq
    rng = np.random.default_rng(2024)
    n = 30000
    cols = big_columns(rng, n)
    data, info = pw.write_file(oracle, cols, n, rows_per_group=11000, page_size=2048)
    ft = pq.read_footer(data)
    assert ft["num_row_groups"] == 3
    dec = pq.Decoder(0)
    try:
        for li, leaf in enumerate(ft["leaves"]):
            col = cols[li]
            chunks = [c for c in ft["chunks"] if c["leaf"] == li]
            tabs = []
            for c in chunks:
                ch = chunk_bytes(data, c)
                tabs.append(oracle.read_chunk(ch, pq.scan_pages(ch, c["num_values"]), leaf["physical_type"], leaf["type_length"],
                                              leaf["max_def"], leaf["max_rep"]))
                assert tabs[-1].status == 0
            assert sum(len(pq.scan_pages(chunk_bytes(data, c), c["num_values"])) for c in chunks) > 6
            model = oracle.ColumnBufferModel(tabs)
            cb = pq.ColumnBuffer(dec, [(chunk_bytes(data, c), None, c["num_values"]) for c in chunks], leaf["physical_type"],
                                 leaf["type_length"], leaf["max_def"], leaf["max_rep"], footer_num_rows=n)
            all_def = np.concatenate([t.def_levels for t in tabs])
            if leaf["physical_type"] == pw.BYTE_ARRAY:
                all_vals = [s for t in tabs for s in oracle.col_strings(("bytes", t.offsets, t.bytes))]
            else:
                all_vals = np.concatenate([t.values for t in tabs])
            slot = 0
            steps = 0
            while True:
                k = int(rng.choice([0, 1, 3, 10, 57, 500, 2047, 2048, 4100, 9000]))
                if rng.random() < 0.5:
                    want_slots, want_rows = model.skip_rows(k)
                    assert cb.skip_rows(k) == want_rows, (col["name"], steps)
                else:
                    want_slots, want_rows = model.read_rows(k)
                    got, rows = cb.read_rows(k)
                    assert rows == want_rows and got.def_levels.size == want_slots, (col["name"], steps)
                    assert np.array_equal(got.def_levels, all_def[slot:slot + want_slots])
                    if leaf["physical_type"] == pw.BYTE_ARRAY:
                        assert rows_of(got, col) == all_vals[slot:slot + want_slots]
                    else:
                        assert np.array_equal(got.values, all_vals[slot:slot + want_slots])
                slot += want_slots
                steps += 1
                if k > 0 and want_slots == 0:
                    break
            assert slot == n
    finally:
        dec.close()


@pytest.mark.gpu
def test_pages_handed_out_one_at_a_time_match_the_oracle_page_by_page(oracle):
    # the layout surface itself: ReadPage after ReadPage over a chunk, (numValues, numRows) and table of every page
    # against the oracle decoding that page alone (behind its dictionary page); then the skip-path calls
    import parquet_go_amd as pq
    from parquet_go_amd.layout_gpu import ChunkPager
    rng = np.random.default_rng(8)
    n = 12000
    cols = big_columns(rng, n)
    data, info = pw.write_file(oracle, cols, n)
    ft = pq.read_footer(data)
    dec = pq.Decoder(0)
    try:
        for c in ft["chunks"]:
            leaf = ft["leaves"][c["leaf"]]
            col = (leaf["physical_type"], leaf["type_length"], leaf["max_def"], leaf["max_rep"])
            ch = chunk_bytes(data, c)
            pages = pq.scan_pages(ch, c["num_values"])
            pager = ChunkPager(dec, ch, *col, total_values=c["num_values"])
            skipper = ChunkPager(dec, ch, *col, total_values=c["num_values"])
            dict_desc = None
            for k, d in enumerate(pages):
                page, nv, nr = pager.read_page()
                raw = skipper.read_page_raw_data()
                snv, snr = raw.get_rldl_from_raw_data()
                if d["page_type"] == pq.DICTIONARY_PAGE:
                    assert (nv, nr, snv, snr) == (0, 0, 0, 0)
                    raw.get_value_from_raw_data()
                    dict_desc = pages[k:k + 1]
                    continue
                descs = pages[k:k + 1] if dict_desc is None else np.concatenate([dict_desc, pages[k:k + 1]])
                o = oracle.read_chunk(ch, descs, *col)
                assert o.status == 0
                assert (nv, nr) == (snv, snr) == (o.n_slots, o.n_rows)
                t = page.data_table
                assert np.array_equal(t.def_levels, o.def_levels) and np.array_equal(t.rep_levels, o.rep_levels)
                assert np.array_equal(raw.data_table.def_levels, o.def_levels) and not raw.data_table.has_values
                if k % 3 == 0:
                    raw.get_value_from_raw_data()
                for tt in ([t, raw.data_table] if k % 3 == 0 else [t]):
                    if o.kind == "bytes":
                        assert np.array_equal(tt.offsets, o.offsets) and np.array_equal(tt.bytes, o.bytes)
                    else:
                        assert np.array_equal(tt.values, o.values)
            with pytest.raises(EOFError):
                pager.read_page()
    finally:
        dec.close()


@pytest.mark.gpu
def test_nested_column_skipped_and_read_across_chunks(oracle):
    # rows of a nested column span several level slots: SkipRows / ReadRows cut at repetition level 0
    import parquet_go_amd as pq
    from gen import pqgen
    rng = np.random.default_rng(3)
    parts = [pqgen.list_chunk(0xAB0 + k, 30000 + 1000 * k, 4096) for k in range(3)]
    tabs = [oracle.read_chunk(c, p, pq.INT64, 0, 3, 1) for c, p in parts]
    model = oracle.ColumnBufferModel(tabs)
    dec = pq.Decoder(0)
    try:
        cb = pq.ColumnBuffer(dec, parts, pq.INT64, 0, 3, 1)
        all_vals = np.concatenate([t.values for t in tabs])
        all_rep = np.concatenate([t.rep_levels for t in tabs])
        slot = 0
        while True:
            k = int(rng.choice([1, 2, 17, 300, 1500, 7000]))
            if rng.random() < 0.5:
                ws, wr = model.skip_rows(k)
                assert cb.skip_rows(k) == wr
            else:
                ws, wr = model.read_rows(k)
                got, rows = cb.read_rows(k)
                assert rows == wr and got.rep_levels.size == ws
                assert np.array_equal(got.rep_levels, all_rep[slot:slot + ws]) and np.array_equal(got.values, all_vals[slot:slot + ws])
            slot += ws
            if ws == 0:
                break
        assert slot == all_rep.size
    finally:
        dec.close()
