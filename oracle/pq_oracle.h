/*
 * CPU oracle for the parquet-go page-decode path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference decoders
 *   encoding/encodingread.go, encoding/binaryread.go,
 *   layout/page.go:139-164 (Decode), :633-718 (ReadDataPageValues), :721-937 (ReadPage)
 *   reader/columnbuffer.go:105-156 (per-chunk page loop)
 * and of the encoders in encoding/encodingwrite.go that the reference's own
 * round-trip tests pair with them.  Nothing in the product library links or
 * calls this file; it is the checker used by tests/, by smoke() and by the
 * cpu_baseline leg of bench.py.
 *
 * Parity pin: checked against the literal byte vectors of
 * encoding/encodingwrite_test.go and the round-trip inputs of
 * encoding/encodingread_test.go (tests/golden/reference_kat.json).
 */
#ifndef PQ_ORACLE_H
#define PQ_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Go error / panic classes of the reference path. */
enum {
  ORC_OK = 0,
  ORC_EOF = -1,            /* io.EOF */
  ORC_UNEXPECTED_EOF = -2, /* io.ErrUnexpectedEOF */
  ORC_PANIC = -3,          /* slice/index out of range, division by zero, negative make(), endless loop */
  ORC_UNSUPPORTED = -4,    /* fmt.Errorf("Unsupported ...") / "Unknown ..." returns */
  ORC_NOMEM = -5,
  ORC_UNSUPPORTED_TYPE = -6, /* a value encoding without a typed result for the column's type (stand-in of the checker) */
  ORC_FLBA_LENGTH = -7      /* a delta-encoded string of a FIXED_LEN_BYTE_ARRAY column whose length is not type_length: the
                               reference hands it on (layout/page.go:689-693, :701-705); the fixed-width image cannot hold it */
};

/* parquet enums (parquet/parquet.go:28-37, :345-355, :533-538) */
enum { ORC_BOOLEAN = 0, ORC_INT32 = 1, ORC_INT64 = 2, ORC_INT96 = 3, ORC_FLOAT = 4, ORC_DOUBLE = 5,
       ORC_BYTE_ARRAY = 6, ORC_FIXED_LEN_BYTE_ARRAY = 7 };
enum { ORC_ENC_PLAIN = 0, ORC_ENC_PLAIN_DICTIONARY = 2, ORC_ENC_RLE = 3, ORC_ENC_BIT_PACKED = 4,
       ORC_ENC_DELTA_BINARY_PACKED = 5, ORC_ENC_DELTA_LENGTH_BYTE_ARRAY = 6, ORC_ENC_DELTA_BYTE_ARRAY = 7,
       ORC_ENC_RLE_DICTIONARY = 8, ORC_ENC_BYTE_STREAM_SPLIT = 9 };
enum { ORC_DATA_PAGE = 0, ORC_INDEX_PAGE = 1, ORC_DICTIONARY_PAGE = 2, ORC_DATA_PAGE_V2 = 3 };

/* bytes.Reader */
typedef struct { const uint8_t *s; int64_t n; int64_t i; } orc_rd;

/* growable int64 vector ([]interface{} of int64 / int32 values) */
typedef struct { int64_t *v; int64_t n; int64_t cap; } orc_i64v;

/* growable byte buffer */
typedef struct { uint8_t *b; int64_t n; int64_t cap; } orc_buf;

/* A decoded value vector.  kind 0: fixed-width elements (elem_size bytes each,
 * BOOLEAN as one 0/1 byte, INT96 as 12 bytes); kind 1: byte arrays as
 * off[0..n] + bytes. */
typedef struct {
  int32_t kind;
  int32_t elem_size;
  int64_t n;
  orc_buf fixed;
  orc_i64v off;
  orc_buf bytes;
} orc_col;

/* Same field order and sizes as pqd_page_desc / pqd_column_desc of the GPU library. */
typedef struct {
  int32_t page_type;   /* ORC_DATA_PAGE, ORC_DICTIONARY_PAGE, ORC_DATA_PAGE_V2 */
  int32_t encoding;
  int32_t num_values;  /* header num_values */
  int32_t rep_len;     /* v2: repetition_levels_byte_length */
  int32_t def_len;     /* v2: definition_levels_byte_length */
  int32_t reserved;
  int64_t offset;      /* byte offset of the page body inside the chunk buffer */
  int64_t length;      /* compressed_page_size (== uncompressed: UNCOMPRESSED codec) */
} orc_page_desc;

typedef struct {
  int32_t physical_type;
  int32_t type_length;
  int32_t max_def;
  int32_t max_rep;
} orc_column_desc;

/* Result of decoding one column chunk: the typed image of layout.Table after
 * all pages were merged (reader/columnbuffer.go:137-146): one entry per level
 * slot; null slots hold zero bytes / zero-length strings. */
typedef struct {
  int64_t n_slots;
  int64_t n_rows;
  int64_t n_nulls;
  int32_t *def;
  int32_t *rep;
  orc_col values;
  int32_t status;
  int32_t bad_page;
} orc_table;

void orc_i64v_free(orc_i64v *v);
void orc_buf_free(orc_buf *b);
void orc_col_free(orc_col *c);
void orc_table_free(orc_table *t);

/* --- decoders (encoding/encodingread.go) --- */
int orc_read_unsigned_varint(orc_rd *r, uint64_t *out);
int orc_read_rle(orc_rd *r, uint64_t header, uint64_t bit_width, orc_i64v *out);
int orc_read_bit_packed(orc_rd *r, uint64_t header, uint64_t bit_width, orc_i64v *out);
int orc_read_rle_bit_packed_hybrid(orc_rd *r, uint64_t bit_width, uint64_t length, orc_i64v *out);
int orc_read_delta_binary_packed_int32(orc_rd *r, orc_i64v *out);
int orc_read_delta_binary_packed_int64(orc_rd *r, orc_i64v *out);
int orc_read_delta_length_byte_array(orc_rd *r, orc_col *out);
int orc_read_delta_byte_array(orc_rd *r, orc_col *out);
int orc_read_plain(orc_rd *r, int32_t data_type, uint64_t cnt, uint64_t bit_width, orc_col *out);
int orc_read_byte_stream_split(orc_rd *r, int32_t data_type, uint64_t cnt, orc_col *out);

/* layout/page.go:633-718.  Index / level results (dictionary and RLE
 * encodings) come back in idx, everything else in vals. */
int orc_read_data_page_values(orc_rd *r, int32_t encoding, int32_t data_type, uint64_t cnt,
                              uint64_t bit_width, orc_i64v *idx, orc_col *vals, int *is_idx);

/* reader/columnbuffer.go:105-156 over the pages of one chunk (page headers
 * already parsed into descs). */
int orc_read_chunk(const uint8_t *chunk, int64_t chunk_len, const orc_page_desc *pages, int32_t n_pages,
                   const orc_column_desc *col, orc_table *out);

/* --- encoders (encoding/encodingwrite.go), used for round trips and fixtures --- */
void orc_write_unsigned_varint(uint64_t num, orc_buf *out);
void orc_write_rle_int64(const int64_t *vals, int64_t n, int32_t bit_width, orc_buf *out);   /* WriteRLE, INT64 */
void orc_write_rle_int32(const int32_t *vals, int64_t n, int32_t bit_width, orc_buf *out);   /* WriteRLEInt32 */
void orc_write_rle_bit_packed_hybrid_int64(const int64_t *vals, int64_t n, int32_t bit_width, orc_buf *out);
void orc_write_rle_bit_packed_hybrid_int32(const int32_t *vals, int64_t n, int32_t bit_width, orc_buf *out);
void orc_write_bit_packed(const int64_t *vals, int64_t n, int64_t bit_width, int if_header, orc_buf *out);
void orc_write_delta_int32(const int32_t *nums, int64_t n, orc_buf *out);
void orc_write_delta_int64(const int64_t *nums, int64_t n, orc_buf *out);
void orc_write_delta_length_byte_array(const int64_t *off, const uint8_t *bytes, int64_t n, orc_buf *out);
void orc_write_delta_byte_array(const int64_t *off, const uint8_t *bytes, int64_t n, orc_buf *out);
void orc_write_plain_boolean(const uint8_t *vals, int64_t n, orc_buf *out);
void orc_write_plain_byte_array(const int64_t *off, const uint8_t *bytes, int64_t n, orc_buf *out);

#ifdef __cplusplus
}
#endif
#endif
