/*
 * pqdecode — MI355X (gfx950) implementation of parquet-go's column-chunk page-decode path.
 *
 * C ABI of libpqdecode.so.  Plain pointers and sizes only; this is the surface a cgo shim
 * binds in place of the reference's per-page Go calls.  The batch unit is one COLUMN CHUNK
 * (all pages of one leaf column in one row group): the caller walks the page headers
 * (pqd_scan_pages, or its own thrift reader), hands over the chunk bytes plus one
 * pqd_page_desc per page, and gets back the typed image of the layout.Table the reference
 * would have built by calling layout.ReadPage + Page.Decode + Table.Merge for every page
 * (reader/columnbuffer.go:105-156).
 *
 * Error convention: every function returns 0 or a negative PQD_ERR_* code; nothing aborts or
 * throws across the ABI.  Codes -1..-3 correspond to the reference's io.EOF,
 * io.ErrUnexpectedEOF and the inputs on which it panics.
 *
 * Threading: a pqd_handle is owned by one thread at a time (one per ColumnBufferType /
 * goroutine in the reference's model, reader/reader.go:306-327); distinct handles are
 * independent and may be used concurrently.
 */
#ifndef PQDECODE_H
#define PQDECODE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQD_OK 0
#define PQD_ERR_EOF (-1)                   /* io.EOF */
#define PQD_ERR_UNEXPECTED_EOF (-2)        /* io.ErrUnexpectedEOF */
#define PQD_ERR_MALFORMED (-3)             /* input on which the reference panics (short stream, index out of range, ...) */
#define PQD_ERR_UNSUPPORTED_ENCODING (-4)  /* "Unsupported Encoding method ..." / "Unknown Encoding method", layout/page.go:671,716 */
#define PQD_ERR_UNSUPPORTED_PAGE_TYPE (-5) /* "Unsupported page type: INDEX_PAGE", layout/page.go:813-814,934 */
#define PQD_ERR_UNSUPPORTED_TYPE (-6)      /* physical type / encoding combination without a typed result: Encoding_RLE
                                              values of a column that is not BOOLEAN, INT32 or INT64 (the reference hands
                                              out int64 numbers in place of such values, layout/page.go:657-667) */
#define PQD_ERR_BIT_WIDTH (-7)             /* bit width the kernels do not implement (> 32 hybrid, > 64 delta) */
#define PQD_ERR_INVALID_ARG (-8)
#define PQD_ERR_HIP (-9)                   /* a HIP runtime call failed; see pqd_last_error() */
#define PQD_ERR_NOMEM (-10)
#define PQD_ERR_NO_DICTIONARY (-11)        /* dictionary-encoded data page without a dictionary page before it */
#define PQD_ERR_THRIFT (-12)               /* page header could not be parsed */
#define PQD_ERR_REFUSED (-13)              /* malformed input that the reference decodes into padded or reinterpreted values and
                                              this build refuses: a BYTE_ARRAY value running past the end of its page (the
                                              reference zero pads it, encodingread.go:98-125), a value of 16 MiB or more,
                                              v2 level sections that contradict the column's maximum levels (the reference
                                              reads the next section in their place, layout/page.go:736-773), length or
                                              prefix streams of byte-array pages announcing more values than the page holds */
#define PQD_ERR_CAPACITY (-14)             /* an output array of the caller is too small (pqd_scan_pages: more pages than `cap`
                                              descriptors); nothing is wrong with the input: grow the array and call again */

/* parquet.Type (parquet/parquet.go:28-37) */
enum { PQD_BOOLEAN = 0, PQD_INT32 = 1, PQD_INT64 = 2, PQD_INT96 = 3, PQD_FLOAT = 4, PQD_DOUBLE = 5,
       PQD_BYTE_ARRAY = 6, PQD_FIXED_LEN_BYTE_ARRAY = 7 };
/* parquet.Encoding (parquet/parquet.go:345-355) */
enum { PQD_ENC_PLAIN = 0, PQD_ENC_PLAIN_DICTIONARY = 2, PQD_ENC_RLE = 3, PQD_ENC_BIT_PACKED = 4,
       PQD_ENC_DELTA_BINARY_PACKED = 5, PQD_ENC_DELTA_LENGTH_BYTE_ARRAY = 6, PQD_ENC_DELTA_BYTE_ARRAY = 7,
       PQD_ENC_RLE_DICTIONARY = 8, PQD_ENC_BYTE_STREAM_SPLIT = 9 };
/* parquet.PageType (parquet/parquet.go:533-538) */
enum { PQD_DATA_PAGE = 0, PQD_INDEX_PAGE = 1, PQD_DICTIONARY_PAGE = 2, PQD_DATA_PAGE_V2 = 3 };

/* One page of a column chunk, header already parsed (replaces what layout.ReadPage reads from
 * parquet.PageHeader, layout/page.go:726-833). */
typedef struct {
  int32_t page_type;  /* PQD_DATA_PAGE | PQD_DICTIONARY_PAGE | PQD_DATA_PAGE_V2 */
  int32_t encoding;   /* DataPageHeader.Encoding / DataPageHeaderV2.Encoding / DictionaryPageHeader.Encoding */
  int32_t num_values; /* header num_values (level slots for data pages, entries for dictionary pages) */
  int32_t rep_len;    /* v2: repetition_levels_byte_length, else 0 */
  int32_t def_len;    /* v2: definition_levels_byte_length, else 0 */
  int32_t reserved;
  int64_t offset;     /* byte offset of the page body (first byte after the thrift header) in the chunk buffer */
  int64_t length;     /* compressed_page_size; pages must be UNCOMPRESSED (compress/ is out of scope) */
} pqd_page_desc;

/* What the reference takes from schema.SchemaHandler and parquet.ColumnMetaData for one leaf
 * (layout/page.go:797-805, :821-822, :897-900). */
typedef struct {
  int32_t physical_type; /* PQD_* type */
  int32_t type_length;   /* FIXED_LEN_BYTE_ARRAY length, else 0 */
  int32_t max_def;       /* SchemaHandler.MaxDefinitionLevel(path) */
  int32_t max_rep;       /* SchemaHandler.MaxRepetitionLevel(path) */
} pqd_column_desc;

/* Totals of one decoded chunk (what ReadPage returns per page, summed: numValues, numRows). */
typedef struct {
  int64_t num_slots;   /* level slots = sum of data-page num_values = len(Table.Values) */
  int64_t num_values;  /* non-null values */
  int64_t num_nulls;   /* slots with def != max_def (layout/page.go:883-888) */
  int64_t num_rows;    /* slots with rep == 0 (layout/page.go:925-927) */
  int64_t bytes_len;   /* BYTE_ARRAY: total payload bytes, else 0 */
  int32_t elem_size;   /* bytes per element of `values` (0 for BYTE_ARRAY) */
  int32_t status;      /* PQD_OK or the first failing page's error */
  int32_t bad_page;    /* index of that page, else -1 */
  int32_t has_levels;  /* bit 0: def array present, bit 1: rep array present (absent = all zero) */
} pqd_chunk_result;

/* Device-resident result buffers; owned by the handle, valid until its next decode call.
 * values: num_slots elements of elem_size bytes, null slots zero (BOOLEAN one byte 0/1, INT96 12 bytes).
 * BYTE_ARRAY: offsets[num_slots+1] (int64) into bytes; null slots have zero length.
 * def_levels / rep_levels: int32[num_slots], NULL when the column's max level is 0. */
typedef struct {
  void *values;
  int32_t *def_levels;
  int32_t *rep_levels;
  int64_t *offsets;
  uint8_t *bytes;
} pqd_device_result;

/* Per-kernel device time of the last decode call, when profiling is on. */
typedef struct {
  char name[48];
  float ms;
  int32_t launches;
} pqd_kernel_time;

typedef struct pqd_handle pqd_handle;

const char *pqd_version(void);
const char *pqd_last_error(void);      /* thread-local text of the last PQD_ERR_HIP / INVALID_ARG */
const char *pqd_strerror(int code);
int pqd_device_count(int *n_out);

/* One handle = one HIP stream + grow-only device workspaces on `device`.
 * stream == NULL: the handle creates its own stream; otherwise it enqueues on the given hipStream_t. */
int pqd_handle_create(int device, void *stream, pqd_handle **out);
int pqd_handle_destroy(pqd_handle *h);
int pqd_set_profiling(pqd_handle *h, int on);

/* Kernel-variant choices of a handle.  Results never depend on them; tests pin each variant with them.  value -1
 * returns a choice to the library.  New handles start from the environment (read once per process):
 * PQD_PRESCAN=wide|wave, PQD_DIRECT_RLE=0|1, PQD_JUMP=0|1, PQD_NO_PLAIN_SCATTER, PQD_SPLIT=0|1, PQD_POISON=1, PQD_DELTA_BLOCKS=0|1, PQD_FUSED=0|1, PQD_BYTE_OFFSETS=0|1, PQD_DEVICE_TABLES=0|1. */
enum {
  PQD_OPT_PRESCAN = 0,        /* run-header walk: 0 a wavefront per stream, 1 a workgroup per stream */
  PQD_OPT_DIRECT_RLE = 1,     /* 0 / 1: expansion of RLE-only pages straight from the page */
  PQD_OPT_JUMP = 2,           /* 0 / 1: spans of short runs resolved by all lanes at once */
  PQD_OPT_PLAIN_SCATTER = 3,  /* 0 / 1 / 2: nullable PLAIN columns scattered straight from the pages (0: through a dense copy); 1: by the validity
                                 bits the level expansion leaves where it can, 2: always by reading the definition levels again */
  PQD_OPT_SPLIT = 4,          /* 0 / 1: head of a mid-size dictionary kept in LDS */
  PQD_OPT_POISON = 5,         /* 1: every work and output buffer is filled with 0xFF before each call (debugging aid) */
  PQD_OPT_DELTA_BLOCKS = 6,   /* 0 / 1 / 2: DELTA_BINARY_PACKED streams decoded block-parallel (0: a wavefront per page); 2: the block headers of
                                 every stream are walked first, 1: streams whose blocks all look one length long skip that walk */
  PQD_OPT_FUSED = 7,          /* 0 / 1: index streams of large pages walked and expanded by one workgroup per page, in one pass */
  PQD_OPT_BYTE_OFFSETS = 8,   /* 0 / 1: BYTE_ARRAY output offsets in one pass over the lengths (tiles look back for the sum in front of them), and for
                                 chunks of DELTA_LENGTH_BYTE_ARRAY pages only straight from the decoded lengths; 0: sums, scan and offsets as three kernels */
  PQD_OPT_DEVICE_TABLES = 9,  /* 0 / 1: per-stream tables of a call (run-table ranges, tile prefixes, tile -> stream maps) worked out by a kernel from
                                 the uploaded page table instead of by the host; the library's choice: chunks of 4096 data pages and more */
  PQD_OPT_COUNT = 10
};
int pqd_set_option(pqd_handle *h, int option, int value);

/* Host-side page walk over the bytes of one column chunk: parses each thrift compact PageHeader and
 * fills one descriptor per page until total_values level slots are covered (the loop condition of
 * ColumnBufferType.ReadPage, reader/columnbuffer.go:106) or the buffer ends.
 * Returns PQD_ERR_CAPACITY (with *n_out = cap) when the chunk holds more pages than `cap`: call again with a larger
 * array.  Replaces layout.ReadPageHeader, layout/page.go:625-630. */
int pqd_scan_pages(const uint8_t *chunk, size_t len, int64_t total_values, pqd_page_desc *out, int32_t cap,
                   int32_t *n_out);

/* ---- footer walk (host) ----
 * pqd_footer_locate: the last 8 bytes of a file -> length of the thrift FileMetaData in front of them (checks the
 * PAR1 magic; ParquetReader.GetFooterSize, reader/reader.go:148-160).
 * pqd_footer_parse: the FileMetaData bytes -> leaf columns (physical type, type length, maximum definition and
 * repetition level as schema.SchemaHandler derives them, schema/schemahandler.go:101-135, dotted path without the
 * root) and one entry per column chunk (row group, leaf, codec, file offset of its first page as
 * ColumnBufferType.NextRowGroup picks it, reader/columnbuffer.go:89-99, byte length, values, rows).
 * Replaces ParquetReader.ReadFooter (reader/reader.go:162-176) for what the decode path needs. */
typedef struct {
  int32_t physical_type, type_length, max_def, max_rep;
  int32_t schema_index;   /* position in the flattened schema list */
  char path[236];         /* "a.b.c", NUL terminated, truncated if longer */
} pqd_leaf_desc;
typedef struct {
  int32_t row_group, leaf;
  int32_t codec;          /* parquet.CompressionCodec; only UNCOMPRESSED (0) chunks can be decoded */
  int32_t reserved;
  int64_t offset;         /* file offset of the chunk's first page header */
  int64_t total_compressed_size;
  int64_t num_values;     /* level slots in the chunk */
  int64_t num_rows;       /* rows of the row group */
} pqd_chunk_desc;
typedef struct pqd_footer pqd_footer;
int pqd_footer_locate(const uint8_t *tail8, size_t len, int64_t *footer_len);
int pqd_footer_parse(const uint8_t *footer, size_t len, pqd_footer **out);
void pqd_footer_free(pqd_footer *f);
int64_t pqd_footer_num_rows(const pqd_footer *f);
int32_t pqd_footer_num_row_groups(const pqd_footer *f);
int32_t pqd_footer_num_leaves(const pqd_footer *f);
int32_t pqd_footer_num_chunks(const pqd_footer *f);
int pqd_footer_leaf(const pqd_footer *f, int32_t i, pqd_leaf_desc *out);
int pqd_footer_chunk(const pqd_footer *f, int32_t i, pqd_chunk_desc *out);

/* Decode one column chunk whose bytes are in HOST memory: copies them through pinned staging to the
 * device, then runs the same kernels as pqd_decode_chunk_device.  Asynchronous on the handle's stream
 * apart from the staging copy; pqd_wait() completes it.
 * Replaces the per-page loop layout.ReadPage (layout/page.go:721) + Page.Decode (:139) + Table.Merge. */
int pqd_decode_chunk(pqd_handle *h, const uint8_t *chunk, size_t len, const pqd_page_desc *pages,
                     int32_t n_pages, const pqd_column_desc *col);

/* A handle runs one call at a time: a decode call on a handle whose previous call has not been waited for first
 * waits for it (its result is dropped).  Use several handles to keep several calls in flight.
 *
 * Page-locked input buffer of the handle, at least len bytes: a caller that reads the chunk from the file straight
 * into it (the io.ReadFull of reader/columnbuffer.go:105-156's callers) and passes the same pointer to
 * pqd_decode_chunk / pqd_decode_levels saves the copy into staging.  A handle has two such buffers that take turns:
 * the pointer is valid until the second next pqd_input_buffer call, and may be filled while the decode of the
 * buffer handed out before it is still in flight. */
int pqd_input_buffer(pqd_handle *h, size_t len, uint8_t **buf);

/* Same, chunk bytes already in device memory.  d_chunk must be 16-byte aligned and readable up to
 * len rounded up to a multiple of 16 (any hipMalloc / torch allocation is). */
int pqd_decode_chunk_device(pqd_handle *h, const void *d_chunk, size_t len, const pqd_page_desc *pages,
                            int32_t n_pages, const pqd_column_desc *col);

/* Levels only: repetition / definition levels, null and row counts of every page, values left undecoded.
 * What ColumnBufferType.SkipRows needs to find a row boundary (reader/columnbuffer.go:158-230,
 * layout.ReadPageRawData + Page.GetRLDLFromRawData, layout/page.go:433-559).  Results as for pqd_decode_chunk
 * with no values / offsets / bytes (pqd_copy_out rejects those pointers, pqd_result_device returns NULL). */
int pqd_decode_levels(pqd_handle *h, const uint8_t *chunk, size_t len, const pqd_page_desc *pages,
                      int32_t n_pages, const pqd_column_desc *col);
int pqd_decode_levels_device(pqd_handle *h, const void *d_chunk, size_t len, const pqd_page_desc *pages,
                             int32_t n_pages, const pqd_column_desc *col);

/* Several column chunks of one leaf column in one call (consecutive row groups: what ColumnBufferType.ReadRows pulls
 * through NextRowGroup + ReadPage until it has its rows, reader/columnbuffer.go:53-103, :232-262).  chunks[k] / lens[k]
 * are the bytes of chunk k, `pages` holds the descriptors of all chunks back to back (offsets relative to their own
 * chunk), pages_per_chunk[k] says how many belong to chunk k.  One meta block, one kernel sequence; the result is the
 * table of all chunks merged in order, pqd_page_counts reports every page.  A dictionary page serves the data pages
 * after it up to the next dictionary page or the end of its chunk.  flags: PQD_LEVELS_ONLY = levels and counts only.
 * The _device variant takes 16-byte aligned device pointers, each readable up to lens[k] rounded up to 16. */
#define PQD_LEVELS_ONLY 1
int pqd_decode_chunks(pqd_handle *h, int32_t n_chunks, const uint8_t *const *chunks, const size_t *lens,
                      const pqd_page_desc *pages, const int32_t *pages_per_chunk, const pqd_column_desc *col, int32_t flags);
int pqd_decode_chunks_device(pqd_handle *h, int32_t n_chunks, const void *const *d_chunks, const size_t *lens,
                             const pqd_page_desc *pages, const int32_t *pages_per_chunk, const pqd_column_desc *col,
                             int32_t flags);

/* pqd_decode_chunk_device for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc,
 * code that must not disturb a graph capture going on elsewhere in the process by allocating).  The call itself is
 * synchronous - it waits for `stream` and copies the totals back - so it cannot be made while `stream` is being
 * captured.  It has no counterpart in the reference.  No handle: the work space is a fixed
 * array of 32 MiB that every GPU the library runs on holds from load to process end, the outputs are the caller's
 * device buffers, and the call makes no hipMalloc / hipFree / hipHostMalloc (host heap only, for the page table).
 * Fixed-width physical types only (BYTE_ARRAY: PQD_ERR_INVALID_ARG).  d_values: num_slots * elem_size bytes
 * (values_bytes says how many there are), d_def_levels / d_rep_levels: level_slots int32 each, NULL for a column
 * without that level.  A chunk whose run tables, tile tables and (nullable columns) dense values do not fit the array
 * returns PQD_ERR_NOMEM, outputs that are too small PQD_ERR_INVALID_ARG; nothing has been written then.  Synchronous:
 * enqueues on `stream` (NULL: the default stream), waits for it and fills *res; returns the chunk's status.  Calls
 * on one GPU share that GPU's array: they are serialised inside the library, and the array keeps no state between
 * calls. */
int pqd_decode_chunk_device_noalloc(int device, void *stream, const void *d_chunk, size_t len, const pqd_page_desc *pages,
                                    int32_t n_pages, const pqd_column_desc *col, void *d_values, size_t values_bytes,
                                    int32_t *d_def_levels, int32_t *d_rep_levels, size_t level_slots,
                                    pqd_chunk_result *res);

/* Wait for the handle's stream and collect status and totals.  Decode calls only enqueue work (BYTE_ARRAY columns
 * included: their byte output is sized from the page bytes, and a call whose strings need more - dictionary or
 * prefix expansion - has its copies redone here into a larger buffer). */
int pqd_wait(pqd_handle *h, pqd_chunk_result *res);

/* Device pointers of the last decoded chunk (after pqd_wait). */
int pqd_result_device(pqd_handle *h, pqd_device_result *out);

/* Copy the last result to host buffers (any pointer may be NULL to skip it).  def/rep are written as
 * zeros when the column has no such levels, as the reference's Table holds them. */
int pqd_copy_out(pqd_handle *h, void *values, int32_t *def_levels, int32_t *rep_levels, int64_t *offsets,
                 uint8_t *bytes);

/* The last result in page-locked host memory owned by the handle (valid until its next decode call): the same
 * arrays as pqd_device_result, moved by DMA at link speed instead of through a bounce buffer into pageable
 * memory.  The caller reads them in place (the Go shim boxes layout.Table.Values straight from them,
 * layout/table.go:28-49) or copies what it keeps.  Level pointers are NULL when the column has no such level.
 * Fields of pqd_device_result carry host addresses here. */
int pqd_result_host(pqd_handle *h, pqd_device_result *out);

/* Row index of the last decoded chunk: entry r = slot at which row r begins (a slot with repetition level 0; every
 * slot when max_rep is 0), entry num_rows = num_slots.  With it the caller cuts the typed arrays at row boundaries
 * the way Table.Pop does (layout/table.go:73-103: the first n rows are the slots before the (n+1)-th slot with
 * repetition level 0) and ColumnBufferType.ReadRows hands rows out (reader/columnbuffer.go:232-262).
 * host_out: num_rows + 1 entries (cap says how many fit) or NULL; device_out: receives the device array or NULL.
 * The device array is owned by the handle and valid until its next decode call. */
int pqd_row_starts(pqd_handle *h, int64_t *host_out, int64_t cap, int64_t **device_out);

/* Per-page (numNulls, numRows, numValues) of the last decoded chunk, 3 x int64 per page in page order
 * (dictionary pages report zeros). */
int pqd_page_counts(pqd_handle *h, int64_t *out, int32_t cap_pages);

/* How many data pages of the last decode call (after pqd_wait) had their dictionary-index stream expanded straight from
 * the page, by the RLE-only kernel or the one-pass kernel, instead of through a run table.  A diagnostic: results do
 * not depend on it.  No counterpart in the reference (its decoders have one path, encoding/encodingread.go:230-268). */
int pqd_direct_streams(pqd_handle *h, int32_t *n_out);

/* Per-kernel timings of the last decode call (pqd_set_profiling(h, 1) first). */
int pqd_kernel_times(pqd_handle *h, pqd_kernel_time *out, int32_t cap, int32_t *n_out);

#ifdef __cplusplus
}
#endif
#endif
