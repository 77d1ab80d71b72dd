"""Row-wise hand-out of decoded column chunks: the typed counterpart of reader.ColumnBufferType.

The reference appends every decoded page to ColumnBufferType.DataTable and hands rows out with ReadRows(num)
(reader/columnbuffer.go:232-262), which cuts the table with Table.Pop (layout/table.go:73-103): the first n rows are
the slots in front of the (n+1)-th slot whose repetition level is 0.  Here the unit appended is a whole column chunk
decoded on the GPU, and the cut positions come from pqd_row_starts instead of a walk over the levels.  The rows
handed out are the same: DataTableNumRows is kept one below the row starts seen until the last chunk has been read
(columnbuffer.go:44, :58), because the last row of a chunk may continue in the next one.
"""
import numpy as np

from .binding import BYTE_ARRAY, DecodeError, Table


class RowSlice(object):
    """What Pop returns: the slots of the rows handed out (values typed instead of boxed)."""
    def __init__(self):
        self.def_levels = np.zeros(0, np.int32)
        self.rep_levels = np.zeros(0, np.int32)
        self.kind = "fixed"
        self.values = None      # fixed: uint8 [n_slots, elem_size]
        self.offsets = None     # bytes: int64 [n_slots + 1], rebased to 0
        self.bytes = None


class ColumnBuffer(object):
    """chunks: iterable of (chunk_bytes, pages) in row-group order for one leaf column."""

    def __init__(self, decoder, chunks, physical_type, type_length=0, max_def=0, max_rep=0):
        self.dec = decoder
        self.chunks = iter(chunks)
        self.col = (physical_type, type_length, max_def, max_rep)
        self.is_bytes = physical_type == BYTE_ARRAY
        self.data_table_num_rows = -1          # columnbuffer.go:44
        self.eof = False
        self._def = np.zeros(0, np.int32)
        self._rep = np.zeros(0, np.int32)
        self._values = None
        self._offsets = np.zeros(1, np.int64)
        self._bytes = np.zeros(0, np.uint8)
        self._row_starts = np.zeros(0, np.int64)   # slot of every row start still in the table

    def _read_chunk(self):
        """ReadPage for all pages of the next chunk (columnbuffer.go:105-156)."""
        try:
            chunk, pages = next(self.chunks)
        except StopIteration:
            if not self.eof:
                self.eof = True
                self.data_table_num_rows += 1      # NextRowGroup at the end of the file, columnbuffer.go:56-60
            return False
        self.dec._is_bytes = self.is_bytes
        self.dec.submit(chunk, pages, *self.col)
        r = self.dec.wait()
        if r.status != 0:
            raise DecodeError(r.status, r.bad_page)
        t = self.dec.fetch(r)
        starts = self.dec.row_starts(r)[:-1]
        base = self._def.size
        self._row_starts = np.concatenate([self._row_starts, starts + base])
        self._def = np.concatenate([self._def, t.def_levels])
        self._rep = np.concatenate([self._rep, t.rep_levels])
        if self.is_bytes:
            self._offsets = np.concatenate([self._offsets, t.offsets[1:] + self._bytes.size])
            self._bytes = np.concatenate([self._bytes, t.bytes])
        else:
            self._values = t.values if self._values is None else np.concatenate([self._values, t.values])
        self.data_table_num_rows += r.num_rows
        return True

    def read_rows(self, num):
        """-> (RowSlice, rows handed out), reader/columnbuffer.go:232-262."""
        while self.data_table_num_rows < num and self._read_chunk():
            pass
        if self.data_table_num_rows < 0:
            self.data_table_num_rows = 0
        num = min(num, self.data_table_num_rows)
        # Table.Pop: the slots in front of row start number `num` (all of them when the table holds no more starts)
        end = int(self._row_starts[num]) if num < self._row_starts.size else self._def.size
        out = RowSlice()
        out.def_levels, out.rep_levels = self._def[:end], self._rep[:end]
        self._def, self._rep = self._def[end:], self._rep[end:]
        if self.is_bytes:
            out.kind = "bytes"
            b0, b1 = int(self._offsets[0]), int(self._offsets[end])
            out.offsets = self._offsets[:end + 1] - b0
            out.bytes = self._bytes[b0:b1]
            self._offsets = self._offsets[end:]
        else:
            esz = 0 if self._values is None else self._values.shape[1]
            out.values = np.zeros((0, esz), np.uint8) if self._values is None else self._values[:end]
            if self._values is not None:
                self._values = self._values[end:]
        self._row_starts = self._row_starts[num:] - end
        self.data_table_num_rows -= num
        return out, num
