"""reader.ColumnBufferType, statement by statement, over the GPU-backed layout surface (layout_gpu.py).

This is the caller the boundary must keep working: reader/columnbuffer.go pulls pages through layout.ReadPage /
ReadPageRawData, merges them into DataTable and hands rows out with ReadRows / SkipRows.  The methods below follow
that file line by line (NextRowGroup :53-103, ReadPage :105-156, ReadPageForSkip :158-192, SkipRows :194-230,
ReadRows :232-262) with typed arrays in place of []interface{}; only the `layout` calls underneath are new.  The Go
binding leaves the Go file itself untouched; this mirror is what the tests drive, because no Go toolchain is here.
"""
import numpy as np

from .binding import BYTE_ARRAY, DICTIONARY_PAGE, DecodeError
from .layout_gpu import ChunkPager


class RowSlice(object):
    """What Table.Pop returns: the slots of the rows handed out (values typed instead of boxed)."""

    def __init__(self):
        self.def_levels = np.zeros(0, np.int32)
        self.rep_levels = np.zeros(0, np.int32)
        self.kind = "fixed"
        self.values = None      # fixed: uint8 [n_slots, elem_size]
        self.offsets = None     # bytes: int64 [n_slots + 1], rebased to 0
        self.bytes = None


class _Table(object):
    """layout.Table (layout/table.go:28-49) with typed columns.  Slots whose value the reference holds as nil - null
    slots, and every slot of a page read for skipping - hold zero / an empty string."""

    def __init__(self, is_bytes, elem_size=0):
        self.is_bytes = is_bytes
        self.def_levels = np.zeros(0, np.int32)
        self.rep_levels = np.zeros(0, np.int32)
        self.values = np.zeros((0, elem_size), np.uint8)
        self.offsets = np.zeros(1, np.int64)
        self.bytes = np.zeros(0, np.uint8)

    def __len__(self):
        return self.def_levels.size

    def merge(self, pt):
        """Table.Merge, layout/table.go:52-71, for one page table (or n nil values when pt is an int)."""
        if isinstance(pt, int):
            n = pt
            dl = rl = np.zeros(n, np.int32)
            vals, offs, data = None, None, None
        else:
            n, dl, rl = len(pt), pt.def_levels, pt.rep_levels
            vals, offs, data = pt.values, pt.offsets, pt.bytes
        self.def_levels = np.concatenate([self.def_levels, dl])
        self.rep_levels = np.concatenate([self.rep_levels, rl])
        if self.is_bytes:
            if offs is None:
                offs, data = np.zeros(n + 1, np.int64), np.zeros(0, np.uint8)
            self.offsets = np.concatenate([self.offsets, offs[1:] + self.offsets[-1]])
            self.bytes = np.concatenate([self.bytes, data])
        else:
            if vals is None:
                vals = np.zeros((n, self.values.shape[1]), np.uint8)
            if self.values.shape[1] != vals.shape[1] and len(self) == n:
                self.values = np.zeros((0, vals.shape[1]), np.uint8)
            self.values = np.concatenate([self.values, vals])

    def set_tail(self, pt):
        """The loop of SkipRows that copies the last page's values over the tail of the table
        (reader/columnbuffer.go:214-218)."""
        n = min(len(pt), len(self))
        if n == 0:
            return
        if self.is_bytes:
            keep = len(self) - n
            o = pt.offsets[len(pt) - n:]
            base = int(self.offsets[keep])
            self.bytes = np.concatenate([self.bytes[:base], pt.bytes[int(o[0]):int(o[-1])]])
            self.offsets = np.concatenate([self.offsets[:keep + 1], o[1:] - o[0] + base])
        else:
            if self.values.shape[1] != pt.values.shape[1]:
                self.values = np.zeros((len(self), pt.values.shape[1]), np.uint8)
            self.values[len(self) - n:] = pt.values[len(pt) - n:]

    def pop(self, num_rows):
        """Table.Pop, layout/table.go:73-103: the slots in front of row start number num_rows."""
        starts = np.flatnonzero(self.rep_levels == 0)
        end = int(starts[num_rows]) if num_rows < starts.size else len(self)
        out = RowSlice()
        out.def_levels, out.rep_levels = self.def_levels[:end], self.rep_levels[:end]
        self.def_levels, self.rep_levels = self.def_levels[end:], self.rep_levels[end:]
        if self.is_bytes:
            out.kind = "bytes"
            b0, b1 = int(self.offsets[0]), int(self.offsets[end])
            out.offsets = self.offsets[:end + 1] - b0
            out.bytes = self.bytes[b0:b1]
            self.offsets = self.offsets[end:]
        else:
            out.values = self.values[:end]
            self.values = self.values[end:]
        return out


class ColumnBuffer(object):
    """chunks: the column's chunk of every row group, in order, as (chunk_bytes, pages) or (chunk_bytes, pages,
    num_values): the bytes from the chunk's first page header on (reader/columnbuffer.go:89-99), its page descriptors
    (None: the headers are walked with pqd_scan_pages) and ColumnMetaData.NumValues."""

    def __init__(self, decoder, chunks, physical_type, type_length=0, max_def=0, max_rep=0, footer_num_rows=None):
        self.dec = decoder
        self.row_groups = list(chunks)
        self.col = (physical_type, type_length, max_def, max_rep)
        self.is_bytes = physical_type == BYTE_ARRAY
        self.footer_num_rows = footer_num_rows
        self.row_group_index = 0
        self.pager = None                      # ThriftReader positioned in the current chunk
        self.chunk_num_values = 0              # ChunkHeader.MetaData.NumValues
        self.chunk_read_values = 0
        self.dict_page = None
        self.data_table = None
        self.data_table_num_rows = -1          # :44
        self.last_error = None
        self.next_row_group()                  # :47-49 (io.EOF is not an error here)

    # ---- NextRowGroup, :53-103 ----
    def next_row_group(self):
        if self.row_group_index >= len(self.row_groups):
            self.data_table_num_rows += 1      # :58 "DataTableNumRows is one smaller than real rows number"
            return EOFError("no more row groups")
        self.row_group_index += 1
        rg = self.row_groups[self.row_group_index - 1]
        chunk, pages = rg[0], rg[1]
        self.pager = ChunkPager(self.dec, chunk, *self.col) if pages is None else _with_pages(self.dec, chunk, pages, self.col)
        if len(rg) > 2:
            self.chunk_num_values = int(rg[2])
        else:
            pg = self.pager.pages
            self.chunk_num_values = int(sum(max(int(p["num_values"]), 0) for p in pg if p["page_type"] != DICTIONARY_PAGE))
        self.chunk_read_values = 0
        self.dict_page = None                  # :101
        return None

    def _new_table(self):
        return _Table(self.is_bytes)

    # ---- ReadPage, :105-156 ----
    def read_page(self):
        if self.pager is not None and self.chunk_read_values < self.chunk_num_values:
            try:
                page, num_values, num_rows = self.pager.read_page()
            except EOFError as err:
                # data is nil and rl/dl=0, no pages in file (:110-127)
                if self.data_table is None:
                    self.data_table = self._new_table()
                self.data_table_num_rows = self.chunk_num_values
                self.data_table.merge(self.chunk_num_values - self.chunk_read_values)
                self.chunk_read_values = self.chunk_num_values
                return err
            except DecodeError as err:
                return err
            if page.page_type == DICTIONARY_PAGE:
                self.dict_page = page
                return None
            page.decode(self.dict_page)
            if self.data_table is None:
                self.data_table = self._new_table()
            self.data_table.merge(page.data_table)
            self.chunk_read_values += num_values
            self.data_table_num_rows += num_rows
        else:
            err = self.next_row_group()
            if err is not None:
                return err
            return self.read_page()
        return None

    # ---- ReadPageForSkip, :158-192 ----
    def read_page_for_skip(self):
        if self.pager is not None and self.chunk_read_values < self.chunk_num_values:
            try:
                page = self.pager.read_page_raw_data()
                num_values, num_rows = page.get_rldl_from_raw_data()
            except (EOFError, DecodeError) as err:
                return None, err
            if page.page_type == DICTIONARY_PAGE:
                page.get_value_from_raw_data()
                self.dict_page = page
                return page, None
            if self.data_table is None:
                self.data_table = self._new_table()
            self.data_table.merge(page.data_table)
            self.chunk_read_values += num_values
            self.data_table_num_rows += num_rows
            return page, None
        err = self.next_row_group()
        if err is not None:
            return None, err
        return self.read_page_for_skip()

    # ---- SkipRows, :194-230 ----
    def skip_rows(self, num):
        err, page = None, None
        while self.data_table_num_rows < num and err is None:
            page, err = self.read_page_for_skip()
        self.last_error = err
        if num > self.data_table_num_rows:
            num = self.data_table_num_rows
        if page is not None:
            try:
                page.get_value_from_raw_data()
            except DecodeError as e:
                self.last_error = e
                return 0
            page.decode(self.dict_page)
            if page.page_type != DICTIONARY_PAGE:
                self.data_table.set_tail(page.data_table)
        if self.data_table is None:            # (the reference would dereference nil here: only with no row groups at all)
            self.data_table = self._new_table()
        if num < 0:
            num = 0
        self.data_table.pop(num)
        self.data_table_num_rows -= num
        return num

    # ---- ReadRows, :232-262 ----
    def read_rows(self, num):
        if self.footer_num_rows == 0:
            return RowSlice(), 0
        err = None
        while self.data_table_num_rows < num and err is None:
            err = self.read_page()
        self.last_error = err
        if self.data_table_num_rows < 0:
            self.data_table_num_rows = 0
            self.data_table = self._new_table()
        if self.data_table is None:
            self.data_table = self._new_table()
        if num > self.data_table_num_rows:
            num = self.data_table_num_rows
        res = self.data_table.pop(num)
        self.data_table_num_rows -= num
        return res, num


def _with_pages(dec, chunk, pages, col):
    """A pager over a chunk whose page descriptors the caller already has (no thrift headers to walk)."""
    p = ChunkPager.__new__(ChunkPager)
    ChunkPager.__init__(p, dec, b"", *col)
    p.chunk = np.frombuffer(chunk, dtype=np.uint8) if not isinstance(chunk, np.ndarray) else chunk
    p.pages = np.ascontiguousarray(pages)
    return p
