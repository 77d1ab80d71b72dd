"""Page-at-a-time hand-out behind the reference's `layout` surface, fed by chunk-at-a-time decodes on the GPU.

reader.ColumnBufferType asks `layout` for one page at a time (reader/columnbuffer.go:105-192):

    layout.ReadPage(thriftReader, sh, colMeta)      -> (page, numValues, numRows, err)      layout/page.go:721
    page.Decode(dictPage)                                                                    layout/page.go:139
    layout.ReadPageRawData(thriftReader, sh, colMeta) -> page                                layout/page.go:395
    page.GetRLDLFromRawData(sh)                     -> (numValues, numRows, err)             layout/page.go:433
    page.GetValueFromRawData(sh)                    -> err                                   layout/page.go:562

A GPU launch per 8 KiB page would be all overhead, so the first of these calls on a column chunk decodes every
remaining page of the chunk in one call of the C ABI and the pages are then handed out from that result, one per
call, with the (numValues, numRows) the reference would have returned for each.  The caller - columnbuffer.py here,
reader/columnbuffer.go in the Go binding (integration/go/layout_gpu) - keeps its own loop unchanged.

ChunkPager stands for the thrift reader positioned inside one column chunk (the reference creates one per row group,
reader/columnbuffer.go:99): the state of the hand-out is keyed by it.

Chunks the GPU path does not take go to the host's own page functions (`cpu_route`; in the Go binding the reference's
ReadPage / ReadPageRawData under private names): compressed chunks - the reference's writer compresses with SNAPPY unless
told otherwise, writer/writer.go:63-65, and reads such pages through compress.Uncompress, layout/page.go:753-758,
:780-783 - from their first page, and chunks holding a page the library answers with an unsupported-encoding, unsupported-
type, bit-width, no-dictionary or refused code from that page on.  Nothing in this package decodes pages on the CPU: the
route is the caller's.
"""
import numpy as np

from . import binding as B


class PageTable(object):
    """layout.Table of one page, typed: every level slot of the page; null slots hold zero / an empty string."""

    def __init__(self, kind, def_levels, rep_levels, values=None, offsets=None, data=None):
        self.kind = kind                    # 'fixed' | 'bytes'
        self.def_levels = def_levels
        self.rep_levels = rep_levels
        self.values = values                # fixed: uint8 [n, elem_size]; None while the values are not decoded
        self.offsets = offsets              # bytes: int64 [n + 1] from 0
        self.bytes = data

    def __len__(self):
        return len(self.def_levels)

    @property
    def has_values(self):
        return self.values is not None or self.offsets is not None


class Page(object):
    """layout.Page as far as the reader looks at it: the header's type and the page's table."""

    def __init__(self, pager, index, desc):
        self._pager, self._index = pager, index
        self.page_type = int(desc["page_type"])
        self.encoding = int(desc["encoding"])
        self.header_num_values = int(desc["num_values"])
        self.data_table = None
        self.raw = False                    # handed out by read_page_raw_data: values not decoded yet

    def decode(self, dict_page):
        """Page.Decode (layout/page.go:139-164): the dictionary gather already happened on the device."""
        return None

    def get_rldl_from_raw_data(self):
        """Page.GetRLDLFromRawData (layout/page.go:433-559) -> (numValues, numRows)."""
        return self._pager._levels_of(self)

    def get_value_from_raw_data(self):
        """Page.GetValueFromRawData (layout/page.go:562-622)."""
        return self._pager._values_of(self)


class _Decoded(object):
    """Result of one C-ABI call over pages [first, last) of the chunk (plus the dictionary page in front)."""

    def __init__(self, first, last, table, lead):
        self.first, self.last, self.table = first, last, table
        counts = table.page_counts[lead:lead + (last - first)] if last > first else np.zeros((0, 3), np.int64)
        self.rows = counts[:, 1]
        self.slot0 = np.concatenate([[0], np.cumsum(counts[:, 2])]).astype(np.int64)


UNCOMPRESSED = 0                            # parquet.CompressionCodec_UNCOMPRESSED
# codes on which the chunk goes over to the caller's CPU route from the failing page on: PQD_ERR_UNSUPPORTED_ENCODING,
# PQD_ERR_UNSUPPORTED_TYPE, PQD_ERR_BIT_WIDTH, PQD_ERR_NO_DICTIONARY, PQD_ERR_REFUSED (include/pqdecode.h)
DELEGABLE = (-4, -6, -7, -11, -13)


class ChunkPager(object):
    def __init__(self, decoder, chunk_bytes, physical_type, type_length=0, max_def=0, max_rep=0, total_values=-1, codec=UNCOMPRESSED,
                 cpu_route=None):
        """cpu_route(chunk_bytes, header_offset, dict_header_offset) -> an object with read_page() / read_page_raw_data()
        (the layout surface over the pages from header_offset on; dict_header_offset: where the header of the dictionary page
        in effect starts, or None) and, when dict_header_offset is given, read_dictionary() -> the table of that page."""
        self.dec = decoder
        self.chunk = np.frombuffer(chunk_bytes, dtype=np.uint8) if not isinstance(chunk_bytes, np.ndarray) else chunk_bytes
        self.col = (physical_type, type_length, max_def, max_rep)
        self.is_bytes = physical_type == B.BYTE_ARRAY
        self.cpu_route = cpu_route
        self.cpu = None                     # the CPU route's layout object once the chunk is its own
        self.dict_page_obj = None           # the dictionary page as the caller holds it
        if codec != UNCOMPRESSED:
            # compressed pages are the reference's own business (layout/page.go:753-758, :780-783): the GPU never sees the chunk
            if cpu_route is None:
                raise B.DecodeError(-4, 0, "compressed chunk and no CPU route to hand it to")
            self.cpu = cpu_route(self.chunk, 0, None)
            self.pages = np.zeros(0, B.PAGE_DESC_DTYPE)
            self.cursor, self.dict_index = 0, None
            self._full = self._levels = None
            self._error = {False: None, True: None}
            return
        # the page headers, walked on the host once (layout.ReadPageHeader, layout/page.go:625-630)
        self.pages = B.scan_pages(self.chunk, total_values) if self.chunk.size else np.zeros(0, B.PAGE_DESC_DTYPE)
        self.cursor = 0
        self.dict_index = None              # the dictionary page handed out last (ColumnBufferType.DictPage)
        self._full = None                   # _Decoded with values
        self._levels = None                 # _Decoded, levels only
        self._error = {False: None, True: None}   # per kind of call (full / levels only): (page index, DecodeError) it found

    # ---- the C-ABI calls ----
    def _call(self, first, last, levels_only):
        lead = 1 if (self.dict_index is not None and not levels_only and self.pages[first]["page_type"] != B.DICTIONARY_PAGE) else 0
        descs = np.concatenate([self.pages[self.dict_index:self.dict_index + 1], self.pages[first:last]]) if lead else self.pages[first:last]
        self.dec._is_bytes = self.is_bytes
        if levels_only:
            t = self.dec.decode_levels(self.chunk, descs, *self.col)
        else:
            t = self.dec.decode(self.chunk, descs, *self.col)
        return t, lead

    def _decode_from(self, k, levels_only):
        """Decodes the pages from k to the end of the chunk; pages in front of a failing one stay available."""
        last = len(self.pages)
        t, lead = self._call(k, last, levels_only)
        if t.status != 0:
            bad = k + t.bad_page - lead
            if lead and t.bad_page == 0:
                bad = k                     # the dictionary page handed out earlier failed: the next page meets it
            self._error[levels_only] = (bad, B.DecodeError(t.status, bad))
            last = bad
            if last > k:
                t, lead = self._call(k, last, levels_only)
                if t.status != 0:
                    raise B.DecodeError(t.status, k + t.bad_page - lead)
        d = _Decoded(k, last, t, lead) if last > k else _Decoded(k, k, B.Table(), 0)
        if levels_only:
            self._levels = d
        else:
            self._full = d
            if self._error[False] is None:
                self._levels = d            # a full decode without failures serves the level calls too

    def _covered(self, d, k):
        return d is not None and d.first <= k < d.last

    def _failed(self, k, levels_only):
        e = self._error[levels_only]
        return e is not None and k >= e[0]

    def _check_error(self, k, levels_only):
        if self._failed(k, levels_only):
            raise self._error[levels_only][1]

    def _header_offset(self, k):
        """Where the header of page k starts in the chunk: the end of the page before it."""
        return 0 if k == 0 else int(self.pages[k - 1]["offset"] + self.pages[k - 1]["length"])

    def _to_cpu(self, k):
        """Hands the chunk over to the CPU route from page k on.  A dictionary page handed out earlier went out without a
        table (its entries were used on the device); the pages the CPU route decodes look their values up in it
        (Page.Decode, layout/page.go:139-164), so the route decodes it now and the table is filled in."""
        dict_off = self._header_offset(self.dict_index) if self.dict_index is not None else None
        self.cpu = self.cpu_route(self.chunk, self._header_offset(k), dict_off)
        if dict_off is not None and self.dict_page_obj is not None:
            self.dict_page_obj.data_table = self.cpu.read_dictionary()

    def _slice(self, d, k, with_values):
        j = k - d.first
        a, b = int(d.slot0[j]), int(d.slot0[j + 1])
        t = d.table
        pt = PageTable("bytes" if self.is_bytes else "fixed", t.def_levels[a:b], t.rep_levels[a:b])
        if with_values:
            if self.is_bytes:
                o = t.offsets[a:b + 1]
                pt.offsets, pt.bytes = o - o[0], t.bytes[int(o[0]):int(o[-1])]
            else:
                pt.values = t.values[a:b]
        return pt, b - a, int(d.rows[j])

    # ---- the layout surface ----
    def read_page(self):
        """layout.ReadPage -> (page, numValues, numRows).  EOFError stands for io.EOF: no page header left."""
        if self.cpu is not None:
            return self.cpu.read_page()
        k = self.cursor
        if k >= len(self.pages):
            raise EOFError("no more pages in the chunk")
        page = Page(self, k, self.pages[k])
        if page.page_type not in (B.DATA_PAGE, B.DATA_PAGE_V2, B.DICTIONARY_PAGE):
            raise B.DecodeError(-5, k)      # layout/page.go:813-814, :933-935
        if not self._covered(self._full, k) and not self._failed(k, False):
            self._decode_from(k, False)
        if self._failed(k, False) and self._error[False][1].code in DELEGABLE and self.cpu_route is not None:
            # the reference decodes (or refuses) this page in its own way: from here on the chunk is the CPU route's
            self._to_cpu(k)
            return self.cpu.read_page()
        self._check_error(k, False)
        self.cursor += 1
        if page.page_type == B.DICTIONARY_PAGE:
            self.dict_index, self.dict_page_obj = k, page
            return page, 0, 0
        page.data_table, n, rows = self._slice(self._full, k, True)
        return page, n, rows

    def read_page_raw_data(self):
        """layout.ReadPageRawData: the page as it is in the file; nothing decoded yet."""
        if self.cpu is not None:
            return self.cpu.read_page_raw_data()
        k = self.cursor
        if k >= len(self.pages):
            raise EOFError("no more pages in the chunk")
        page = Page(self, k, self.pages[k])
        if page.page_type not in (B.DATA_PAGE, B.DATA_PAGE_V2, B.DICTIONARY_PAGE):
            raise B.DecodeError(-5, k)      # layout/page.go:410-412
        page.raw = True
        self.cursor += 1
        return page

    def _levels_of(self, page):
        k = page._index
        if page.page_type == B.DICTIONARY_PAGE:
            self.dict_index = k             # (ReadPageForSkip keeps it as DictPage, reader/columnbuffer.go:170-174)
            return 0, 0
        if not self._covered(self._levels, k) and not self._failed(k, True):
            self._decode_from(k, True)
        self._check_error(k, True)
        page.data_table, n, rows = self._slice(self._levels, k, False)
        return n, rows

    def _values_of(self, page):
        k = page._index
        if page.page_type == B.DICTIONARY_PAGE:
            return                          # its entries are decoded with the data pages that use them
        if self._covered(self._full, k):
            page.data_table, _, _ = self._slice(self._full, k, True)
        else:
            t, lead = self._call(k, k + 1, False)
            if t.status != 0:
                raise B.DecodeError(t.status, k)
            page.data_table, _, _ = self._slice(_Decoded(k, k + 1, t, lead), k, True)
        page.raw = False
