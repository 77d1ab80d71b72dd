"""parquet-go page-decode path on MI355X: thin ctypes binding over libpqdecode.so (HIP kernels + C ABI).

The compute lives in csrc/ (hand-written gfx950 kernels behind the C ABI of include/pqdecode.h); this
package only loads the library for tests and benchmarks.  There is no CPU fallback: a missing library is
an ImportError at first use.
"""
from .binding import (  # noqa: F401
    PAGE_DESC_DTYPE, Decoder, DecodeError, Table, lib, lib_path, read_footer, scan_pages, strerror,
    BOOLEAN, INT32, INT64, INT96, FLOAT, DOUBLE, BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY,
    ENC_PLAIN, ENC_PLAIN_DICTIONARY, ENC_RLE, ENC_BIT_PACKED, ENC_DELTA_BINARY_PACKED,
    ENC_DELTA_LENGTH_BYTE_ARRAY, ENC_DELTA_BYTE_ARRAY, ENC_RLE_DICTIONARY, ENC_BYTE_STREAM_SPLIT,
    DATA_PAGE, INDEX_PAGE, DICTIONARY_PAGE, DATA_PAGE_V2,
)
from .columnbuffer import ColumnBuffer, RowSlice  # noqa: F401,E402
