/*
 * CPU oracle for the parquet-go page-decode path.  TEST INFRASTRUCTURE ONLY:
 * never linked into or called from the product library.  See pq_oracle.h.
 *
 * Every function names the reference lines it restates.  The loops keep the
 * reference's structure (byte-at-a-time bit unpack, per-run appends, per-value
 * string building) on purpose: this file is also the "port" CPU baseline.
 */
#include "pq_oracle.h"
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_ALLOC ((int64_t)1 << 33) /* stands in for Go's out-of-memory / makeslice panics */

/* ------------------------------------------------------------------ */
/* containers                                                          */
/* ------------------------------------------------------------------ */
static int i64v_reserve(orc_i64v *v, int64_t need) {
  if (need <= v->cap) return ORC_OK;
  if (need > ORC_MAX_ALLOC / 8) return ORC_NOMEM;
  int64_t nc = v->cap ? v->cap : 64;
  while (nc < need) nc *= 2;
  int64_t *p = (int64_t *)realloc(v->v, (size_t)nc * sizeof(int64_t));
  if (!p) return ORC_NOMEM;
  v->v = p; v->cap = nc;
  return ORC_OK;
}
static int i64v_push(orc_i64v *v, int64_t x) {
  if (v->n == v->cap) { int rc = i64v_reserve(v, v->n + 1); if (rc) return rc; }
  v->v[v->n++] = x;
  return ORC_OK;
}
static int buf_reserve(orc_buf *b, int64_t need) {
  if (need <= b->cap) return ORC_OK;
  /* (strings copied value by value: a dictionary entry of a gigabyte of zero padding, looked up a few thousand times, is
   *  a few terabytes here and a few kilobytes of string headers in the reference - the checker gives up where the
   *  reference's machine would) */
  if (need > ORC_MAX_ALLOC / 4) return ORC_NOMEM;
  int64_t nc = b->cap ? b->cap : 256;
  while (nc < need) nc *= 2;
  uint8_t *p = (uint8_t *)realloc(b->b, (size_t)nc);
  if (!p) return ORC_NOMEM;
  b->b = p; b->cap = nc;
  return ORC_OK;
}
static int buf_append(orc_buf *b, const void *src, int64_t n) {
  int rc = buf_reserve(b, b->n + n); if (rc) return rc;
  if (n) memcpy(b->b + b->n, src, (size_t)n);
  b->n += n;
  return ORC_OK;
}
static int buf_push(orc_buf *b, uint8_t x) { return buf_append(b, &x, 1); }

void orc_i64v_free(orc_i64v *v) { free(v->v); v->v = NULL; v->n = v->cap = 0; }
void orc_buf_free(orc_buf *b) { free(b->b); b->b = NULL; b->n = b->cap = 0; }
void orc_col_free(orc_col *c) { orc_buf_free(&c->fixed); orc_i64v_free(&c->off); orc_buf_free(&c->bytes); c->n = 0; }
void orc_table_free(orc_table *t) { free(t->def); free(t->rep); t->def = t->rep = NULL; orc_col_free(&t->values); }

static void col_init_fixed(orc_col *c, int32_t elem) { memset(c, 0, sizeof *c); c->kind = 0; c->elem_size = elem; }
static int col_init_bytes(orc_col *c) { memset(c, 0, sizeof *c); c->kind = 1; return i64v_push(&c->off, 0); }
static int col_push_fixed(orc_col *c, const void *p) { c->n++; return buf_append(&c->fixed, p, c->elem_size); }
static int col_push_str(orc_col *c, const uint8_t *p, int64_t n) {
  int rc = buf_append(&c->bytes, p, n); if (rc) return rc;
  c->n++;
  return i64v_push(&c->off, c->bytes.n);
}

/* ------------------------------------------------------------------ */
/* bytes.Reader / io helpers                                           */
/* ------------------------------------------------------------------ */
/* bytes.Reader.Read: (0, EOF) when the cursor is at the end, even for an empty
 * destination; otherwise copies what is there and reports no error.  The
 * destination is expected to be zero-initialised (Go make()). */
static int64_t rd_read(orc_rd *r, uint8_t *dst, int64_t len) {
  if (r->i >= r->n) return -1;
  int64_t k = r->n - r->i; if (k > len) k = len;
  if (k > 0) memcpy(dst, r->s + r->i, (size_t)k);
  r->i += k;
  return k;
}
static int rd_read_byte(orc_rd *r, uint8_t *b) {
  if (r->i >= r->n) return ORC_EOF;
  *b = r->s[r->i++];
  return ORC_OK;
}
/* io.ReadFull */
static int rd_read_full(orc_rd *r, uint8_t *dst, int64_t len) {
  if (len == 0) return ORC_OK;
  int64_t got = 0;
  while (got < len) {
    int64_t k = rd_read(r, dst + got, len - got);
    if (k < 0) return got == 0 ? ORC_EOF : ORC_UNEXPECTED_EOF;
    got += k;
  }
  return ORC_OK;
}
static int64_t rd_len(const orc_rd *r) { return r->i >= r->n ? 0 : r->n - r->i; }

/* Go shift semantics: a count >= 64 yields 0. */
static inline uint64_t shl64(uint64_t x, uint64_t s) { return s >= 64 ? 0 : x << s; }
static inline uint64_t shr64(uint64_t x, uint64_t s) { return s >= 64 ? 0 : x >> s; }

/* ------------------------------------------------------------------ */
/* decoders                                                            */
/* ------------------------------------------------------------------ */

/* encoding/encodingread.go:127-143.  The inner err shadows the outer one, so
 * the function never reports an error; EOF ends the loop with a partial value. */
int orc_read_unsigned_varint(orc_rd *r, uint64_t *out) {
  uint64_t res = 0, shift = 0;
  for (;;) {
    uint8_t b;
    if (rd_read_byte(r, &b) != ORC_OK) break;
    res |= shl64((uint64_t)b & 0x7F, shift);
    if ((b & 0x80) == 0) break;
    shift += 7;
  }
  *out = res;
  return ORC_OK;
}

/* encoding/encodingread.go:146-167 */
int orc_read_rle(orc_rd *r, uint64_t header, uint64_t bit_width, orc_i64v *out) {
  uint64_t cnt = header >> 1;
  uint64_t width = (bit_width + 7) / 8;
  uint8_t data[40];
  if (width > 32) return ORC_PANIC; /* bit widths above 255 cannot occur: widths come from one byte */
  memset(data, 0, sizeof data);
  if (width > 0) {
    if (rd_read(r, data, (int64_t)width) < 0) return ORC_EOF;
  }
  /* zero padded to 4 bytes, then LittleEndian.Uint32 of the first four */
  int64_t val = (int64_t)((uint32_t)data[0] | (uint32_t)data[1] << 8 | (uint32_t)data[2] << 16 | (uint32_t)data[3] << 24);
  if (cnt > (uint64_t)(ORC_MAX_ALLOC / 8) || out->n + (int64_t)cnt > ORC_MAX_ALLOC / 8) return ORC_PANIC;
  int rc = i64v_reserve(out, out->n + (int64_t)cnt); if (rc) return rc;
  for (uint64_t i = 0; i < cnt; i++) out->v[out->n++] = val;
  return ORC_OK;
}

/* encoding/encodingread.go:170-228 — the byte-at-a-time state machine. */
int orc_read_bit_packed(orc_rd *r, uint64_t header, uint64_t bit_width, orc_i64v *out) {
  uint64_t num_group = header >> 1;
  uint64_t cnt = num_group * 8;
  uint64_t byte_cnt = cnt * bit_width / 8;
  if (cnt == 0) return ORC_OK;
  if (num_group > (uint64_t)(ORC_MAX_ALLOC / 64) || out->n + (int64_t)cnt > ORC_MAX_ALLOC / 8 || (int64_t)byte_cnt < 0 ||
      (int64_t)byte_cnt > ORC_MAX_ALLOC) return ORC_PANIC;
  int rc = i64v_reserve(out, out->n + (int64_t)cnt); if (rc) return rc;
  if (bit_width == 0) {
    for (uint64_t i = 0; i < cnt; i++) out->v[out->n++] = 0;
    return ORC_OK;
  }
  uint8_t *bytes_buf = (uint8_t *)calloc(byte_cnt ? byte_cnt : 1, 1);
  if (!bytes_buf) return ORC_NOMEM;
  if (rd_read(r, bytes_buf, (int64_t)byte_cnt) < 0) { free(bytes_buf); return ORC_EOF; }

  uint64_t i = 0, len = byte_cnt;
  uint64_t res_cur = 0, need = bit_width, used = 0, left = 8;
  uint8_t b = bytes_buf[0];
  while (i < len) {
    if (left >= need) {
      res_cur |= shl64(shr64((uint64_t)b, used) & (shl64(1, need) - 1), bit_width - need);
      out->v[out->n++] = (int64_t)res_cur;
      left -= need;
      used += need;
      need = bit_width;
      res_cur = 0;
      if (left <= 0 && i + 1 < len) {
        i += 1;
        b = bytes_buf[i];
        left = 8;
        used = 0;
      }
    } else {
      res_cur |= shl64(shr64((uint64_t)b, used), bit_width - need);
      i += 1;
      if (i < len) b = bytes_buf[i];
      need -= left;
      left = 8;
      used = 0;
    }
  }
  free(bytes_buf);
  return ORC_OK;
}

/* encoding/encodingread.go:57-62 + binaryread.go:10-27 for one value */
static int read_plain_int32_one(orc_rd *r, int32_t *out) {
  uint8_t b[4] = {0, 0, 0, 0};
  int rc = rd_read_full(r, b, 4); if (rc) return rc;
  *out = (int32_t)((uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24);
  return ORC_OK;
}

/* encoding/encodingread.go:231-268 */
int orc_read_rle_bit_packed_hybrid(orc_rd *r, uint64_t bit_width, uint64_t length, orc_i64v *out) {
  if (length <= 0) {
    int32_t lb;
    int rc = read_plain_int32_one(r, &lb); if (rc) return rc;
    length = (uint64_t)(int64_t)lb;
  }
  if ((int64_t)length < 0 || (int64_t)length > ORC_MAX_ALLOC) return ORC_PANIC; /* make([]byte, huge) */
  uint8_t *buf = (uint8_t *)calloc(length ? length : 1, 1);
  if (!buf) return ORC_NOMEM;
  if (rd_read(r, buf, (int64_t)length) < 0) { free(buf); return ORC_EOF; }

  orc_rd nr = { buf, (int64_t)length, 0 };
  int rc = ORC_OK;
  while (rd_len(&nr) > 0) {
    uint64_t header;
    orc_read_unsigned_varint(&nr, &header);
    if ((header & 1) == 0) rc = orc_read_rle(&nr, header, bit_width, out);
    else rc = orc_read_bit_packed(&nr, header, bit_width, out);
    if (rc) break;
  }
  free(buf);
  return rc;
}

/* encoding/encodingread.go:270-326 */
int orc_read_delta_binary_packed_int32(orc_rd *r, orc_i64v *out) {
  uint64_t block_size, num_mb, num_values, first_zz;
  orc_read_unsigned_varint(r, &block_size);
  orc_read_unsigned_varint(r, &num_mb);
  orc_read_unsigned_varint(r, &num_values);
  orc_read_unsigned_varint(r, &first_zz);
  int32_t fv32 = (int32_t)first_zz;
  int32_t first = (int32_t)((uint32_t)fv32 >> 1) ^ -(fv32 & 1);
  if (num_mb == 0) return ORC_PANIC; /* integer divide by zero */
  uint64_t vals_in_mb = block_size / num_mb;
  if ((int64_t)num_values < 0 || (int64_t)num_values > ORC_MAX_ALLOC / 8) return ORC_PANIC;

  int64_t base = out->n;
  int rc = i64v_push(out, first); if (rc) return rc;
  orc_i64v cur = {0, 0, 0};
  uint8_t *widths = (uint8_t *)malloc(num_mb > 0 && num_mb < 65536 ? num_mb : 65536);
  if (!widths || num_mb >= 65536) { free(widths); return ORC_PANIC; }
  while ((uint64_t)(out->n - base) < num_values) {
    int64_t before = out->n;
    uint64_t md_zz;
    orc_read_unsigned_varint(r, &md_zz);
    int32_t md32 = (int32_t)md_zz;
    int32_t min_delta = (int32_t)((uint32_t)md32 >> 1) ^ -(md32 & 1);
    for (uint64_t i = 0; i < num_mb; i++) {
      if ((rc = rd_read_byte(r, &widths[i])) != ORC_OK) goto done;
    }
    for (uint64_t i = 0; i < num_mb && (uint64_t)(out->n - base) < num_values; i++) {
      cur.n = 0;
      if ((rc = orc_read_bit_packed(r, (vals_in_mb / 8) << 1, widths[i], &cur)) != ORC_OK) goto done;
      for (int64_t j = 0; j < cur.n && (uint64_t)(out->n - base) < num_values; j++) {
        int32_t prev = (int32_t)out->v[out->n - 1];
        int32_t nv = (int32_t)((uint32_t)prev + (uint32_t)(int32_t)cur.v[j] + (uint32_t)min_delta);
        if ((rc = i64v_push(out, nv)) != ORC_OK) goto done;
      }
    }
    (void)before; /* a block that adds nothing: the Go loop reads on, block header after block header, until ReadByte returns io.EOF */
  }
  out->n = base + (int64_t)num_values; /* res[:numValues] */
done:
  free(widths);
  orc_i64v_free(&cur);
  return rc;
}

/* encoding/encodingread.go:329-383 */
int orc_read_delta_binary_packed_int64(orc_rd *r, orc_i64v *out) {
  uint64_t block_size, num_mb, num_values, first_zz;
  orc_read_unsigned_varint(r, &block_size);
  orc_read_unsigned_varint(r, &num_mb);
  orc_read_unsigned_varint(r, &num_values);
  orc_read_unsigned_varint(r, &first_zz);
  int64_t first = (int64_t)(first_zz >> 1) ^ -((int64_t)first_zz & 1);
  if (num_mb == 0) return ORC_PANIC;
  uint64_t vals_in_mb = block_size / num_mb;
  if ((int64_t)num_values < 0 || (int64_t)num_values > ORC_MAX_ALLOC / 8) return ORC_PANIC;

  int64_t base = out->n;
  int rc = i64v_push(out, first); if (rc) return rc;
  orc_i64v cur = {0, 0, 0};
  uint8_t *widths = (uint8_t *)malloc(num_mb > 0 && num_mb < 65536 ? num_mb : 65536);
  if (!widths || num_mb >= 65536) { free(widths); return ORC_PANIC; }
  while ((uint64_t)(out->n - base) < num_values) {
    int64_t before = out->n;
    uint64_t md_zz;
    orc_read_unsigned_varint(r, &md_zz);
    int64_t min_delta = (int64_t)(md_zz >> 1) ^ -((int64_t)md_zz & 1);
    for (uint64_t i = 0; i < num_mb; i++) {
      if ((rc = rd_read_byte(r, &widths[i])) != ORC_OK) goto done;
    }
    for (uint64_t i = 0; i < num_mb && (uint64_t)(out->n - base) < num_values; i++) {
      cur.n = 0;
      if ((rc = orc_read_bit_packed(r, (vals_in_mb / 8) << 1, widths[i], &cur)) != ORC_OK) goto done;
      for (int64_t j = 0; j < cur.n; j++) {
        uint64_t nv = (uint64_t)out->v[out->n - 1] + (uint64_t)cur.v[j] + (uint64_t)min_delta;
        if ((rc = i64v_push(out, (int64_t)nv)) != ORC_OK) goto done;
      }
    }
    (void)before;
  }
  out->n = base + (int64_t)num_values;
done:
  free(widths);
  orc_i64v_free(&cur);
  return rc;
}

/* encoding/encodingread.go:114-125 for one value: make + lenient Read */
static int read_flba_one(orc_rd *r, uint64_t fixed_length, orc_col *out) {
  if ((int64_t)fixed_length < 0 || (int64_t)fixed_length > ORC_MAX_ALLOC) return ORC_PANIC;
  int rc = buf_reserve(&out->bytes, out->bytes.n + (int64_t)fixed_length); if (rc) return rc;
  uint8_t *dst = out->bytes.b + out->bytes.n;
  memset(dst, 0, (size_t)fixed_length);
  if (rd_read(r, dst, (int64_t)fixed_length) < 0) return ORC_EOF;
  out->bytes.n += (int64_t)fixed_length;
  out->n++;
  return i64v_push(&out->off, out->bytes.n);
}

/* encoding/encodingread.go:385-409 */
int orc_read_delta_length_byte_array(orc_rd *r, orc_col *out) {
  int rc = col_init_bytes(out); if (rc) return rc;
  orc_i64v lengths = {0, 0, 0};
  rc = orc_read_delta_binary_packed_int64(r, &lengths);
  if (rc) { orc_i64v_free(&lengths); return rc; }
  for (int64_t i = 0; i < lengths.n; i++) {
    uint64_t length = (uint64_t)lengths.v[i];
    if (length > 0) {
      if ((rc = read_flba_one(r, length, out)) != ORC_OK) break;
    } else {
      if ((rc = col_push_str(out, NULL, 0)) != ORC_OK) break;
    }
  }
  orc_i64v_free(&lengths);
  return rc;
}

/* encoding/encodingread.go:411-435 */
int orc_read_delta_byte_array(orc_rd *r, orc_col *out) {
  int rc = col_init_bytes(out); if (rc) return rc;
  orc_i64v prefix = {0, 0, 0};
  orc_col suffixes; memset(&suffixes, 0, sizeof suffixes);
  rc = orc_read_delta_binary_packed_int64(r, &prefix);
  if (rc) goto done;
  rc = orc_read_delta_length_byte_array(r, &suffixes);
  if (rc) goto done;
  if (prefix.n == 0 || suffixes.n == 0) { rc = ORC_PANIC; goto done; } /* res[0] = suffixes[0] */
  rc = col_push_str(out, suffixes.bytes.b + suffixes.off.v[0], suffixes.off.v[1] - suffixes.off.v[0]);
  if (rc) goto done;
  for (int64_t i = 1; i < prefix.n; i++) {
    int64_t pl = prefix.v[i];
    int64_t prev_off = out->off.v[i - 1], prev_len = out->off.v[i] - prev_off;
    if (pl < 0 || pl > prev_len) { rc = ORC_PANIC; goto done; }
    if (i >= suffixes.n) { rc = ORC_PANIC; goto done; }
    int64_t sl = suffixes.off.v[i + 1] - suffixes.off.v[i];
    if ((rc = buf_reserve(&out->bytes, out->bytes.n + pl + sl)) != ORC_OK) goto done;
    memcpy(out->bytes.b + out->bytes.n, out->bytes.b + prev_off, (size_t)pl);
    memcpy(out->bytes.b + out->bytes.n + pl, suffixes.bytes.b + suffixes.off.v[i], (size_t)sl);
    out->bytes.n += pl + sl;
    out->n++;
    if ((rc = i64v_push(&out->off, out->bytes.n)) != ORC_OK) goto done;
  }
done:
  orc_i64v_free(&prefix);
  orc_col_free(&suffixes);
  return rc;
}

/* encoding/encodingread.go:13-125 + binaryread.go */
int orc_read_plain(orc_rd *r, int32_t data_type, uint64_t cnt, uint64_t bit_width, orc_col *out) {
  int rc = ORC_OK;
  if ((int64_t)cnt < 0 || (int64_t)cnt > ORC_MAX_ALLOC / 16) return ORC_PANIC;
  switch (data_type) {
    case ORC_BOOLEAN: { /* :35-55 — ReadBitPacked with header cnt<<1, width 1 */
      col_init_fixed(out, 1);
      orc_i64v bits = {0, 0, 0};
      rc = orc_read_bit_packed(r, cnt << 1, 1, &bits);
      if (rc == ORC_OK) {
        for (uint64_t i = 0; i < cnt; i++) { uint8_t v = bits.v[i] > 0; if ((rc = col_push_fixed(out, &v))) break; }
      }
      orc_i64v_free(&bits);
      return rc;
    }
    case ORC_INT32: case ORC_FLOAT: case ORC_INT64: case ORC_DOUBLE: { /* :57-69, :84-96 */
      int32_t sz = (data_type == ORC_INT32 || data_type == ORC_FLOAT) ? 4 : 8;
      col_init_fixed(out, sz);
      if ((rc = buf_reserve(&out->fixed, (int64_t)cnt * sz + 1))) return rc;
      memset(out->fixed.b, 0, (size_t)cnt * sz);
      rc = rd_read_full(r, out->fixed.b, (int64_t)cnt * sz);
      if (rc) return rc;
      out->fixed.n = (int64_t)cnt * sz; out->n = (int64_t)cnt; /* little-endian host: the LE assemble is a copy */
      return ORC_OK;
    }
    case ORC_INT96: { /* :71-82 — lenient Read of 12 bytes, `cur` reused so a short tail keeps old bytes */
      col_init_fixed(out, 12);
      uint8_t cur[12]; memset(cur, 0, 12);
      for (uint64_t i = 0; i < cnt; i++) {
        if (rd_read(r, cur, 12) < 0) return ORC_EOF;
        if ((rc = col_push_fixed(out, cur))) return rc;
      }
      return ORC_OK;
    }
    case ORC_BYTE_ARRAY: { /* :98-112 */
      if ((rc = col_init_bytes(out))) return rc;
      for (uint64_t i = 0; i < cnt; i++) {
        uint8_t b[4] = {0, 0, 0, 0};
        if (rd_read(r, b, 4) < 0) return ORC_EOF;
        uint32_t ln = (uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24;
        if ((rc = buf_reserve(&out->bytes, out->bytes.n + (int64_t)ln))) return rc;
        memset(out->bytes.b + out->bytes.n, 0, ln);
        rd_read(r, out->bytes.b + out->bytes.n, ln); /* error ignored */
        out->bytes.n += ln; out->n++;
        if ((rc = i64v_push(&out->off, out->bytes.n))) return rc;
      }
      return ORC_OK;
    }
    case ORC_FIXED_LEN_BYTE_ARRAY: { /* :114-125 */
      if (bit_width > (uint64_t)1 << 30) return ORC_PANIC;
      col_init_fixed(out, (int32_t)bit_width);
      if ((rc = buf_reserve(&out->fixed, (int64_t)(cnt * bit_width) + 1))) return rc;
      for (uint64_t i = 0; i < cnt; i++) {
        uint8_t *dst = out->fixed.b + out->fixed.n;
        memset(dst, 0, (size_t)bit_width);
        if (rd_read(r, dst, (int64_t)bit_width) < 0) return ORC_EOF;
        out->fixed.n += (int64_t)bit_width; out->n++;
      }
      return ORC_OK;
    }
    default:
      return ORC_UNSUPPORTED; /* "Unknown parquet type" */
  }
}

/* encoding/encodingread.go:437-485 */
int orc_read_byte_stream_split(orc_rd *r, int32_t data_type, uint64_t cnt, orc_col *out) {
  int32_t sz = data_type == ORC_FLOAT ? 4 : 8;
  if ((int64_t)cnt < 0 || (int64_t)cnt > ORC_MAX_ALLOC / 16) return ORC_PANIC;
  col_init_fixed(out, sz);
  uint8_t *buf = (uint8_t *)calloc(cnt * sz + 1, 1);
  if (!buf) return ORC_NOMEM;
  int rc = rd_read_full(r, buf, (int64_t)cnt * sz);
  if (rc == ORC_OK && (rc = buf_reserve(&out->fixed, (int64_t)cnt * sz + 1)) == ORC_OK) {
    for (uint64_t i = 0; i < cnt; i++)
      for (int k = 0; k < sz; k++) out->fixed.b[i * sz + k] = buf[cnt * k + i];
    out->fixed.n = (int64_t)cnt * sz; out->n = (int64_t)cnt;
  }
  free(buf);
  return rc;
}

/* layout/page.go:633-718 */
int orc_read_data_page_values(orc_rd *r, int32_t encoding, int32_t data_type, uint64_t cnt,
                              uint64_t bit_width, orc_i64v *idx, orc_col *vals, int *is_idx) {
  *is_idx = 0;
  if (cnt <= 0) { /* :637-639 */
    if (data_type == ORC_BYTE_ARRAY || encoding == ORC_ENC_DELTA_LENGTH_BYTE_ARRAY || encoding == ORC_ENC_DELTA_BYTE_ARRAY)
      return col_init_bytes(vals);
    col_init_fixed(vals, 0);
    return ORC_OK;
  }
  int rc;
  if (encoding == ORC_ENC_PLAIN) {
    return orc_read_plain(r, data_type, cnt, bit_width, vals);
  } else if (encoding == ORC_ENC_PLAIN_DICTIONARY || encoding == ORC_ENC_RLE_DICTIONARY) { /* :644-655 */
    uint8_t b;
    if ((rc = rd_read_byte(r, &b))) return rc;
    *is_idx = 1;
    rc = orc_read_rle_bit_packed_hybrid(r, b, (uint64_t)rd_len(r), idx);
    if (rc) return rc;
    if ((uint64_t)idx->n < cnt) return ORC_PANIC; /* buf[:cnt] beyond what was decoded */
    idx->n = (int64_t)cnt;
    return ORC_OK;
  } else if (encoding == ORC_ENC_RLE) { /* :657-667 */
    *is_idx = 1;
    rc = orc_read_rle_bit_packed_hybrid(r, bit_width, 0, idx);
    if (rc) return rc;
    if (data_type == ORC_INT32) for (int64_t i = 0; i < idx->n; i++) idx->v[i] = (int32_t)idx->v[i];
    if ((uint64_t)idx->n < cnt) return ORC_PANIC;
    idx->n = (int64_t)cnt;
    return ORC_OK;
  } else if (encoding == ORC_ENC_BIT_PACKED) {
    return ORC_UNSUPPORTED;
  } else if (encoding == ORC_ENC_DELTA_BINARY_PACKED) { /* :673-682: result length is the stream's own */
    orc_i64v tmp = {0, 0, 0};
    if (data_type == ORC_INT32) {
      rc = orc_read_delta_binary_packed_int32(r, &tmp);
      if (rc == ORC_OK) {
        col_init_fixed(vals, 4);
        for (int64_t i = 0; i < tmp.n && rc == ORC_OK; i++) { int32_t v = (int32_t)tmp.v[i]; rc = col_push_fixed(vals, &v); }
      }
    } else if (data_type == ORC_INT64) {
      rc = orc_read_delta_binary_packed_int64(r, &tmp);
      if (rc == ORC_OK) {
        col_init_fixed(vals, 8);
        for (int64_t i = 0; i < tmp.n && rc == ORC_OK; i++) rc = col_push_fixed(vals, &tmp.v[i]);
      }
    } else {
      rc = ORC_UNSUPPORTED;
    }
    orc_i64v_free(&tmp);
    return rc;
  } else if (encoding == ORC_ENC_DELTA_LENGTH_BYTE_ARRAY || encoding == ORC_ENC_DELTA_BYTE_ARRAY) { /* :684-706 */
    rc = encoding == ORC_ENC_DELTA_LENGTH_BYTE_ARRAY ? orc_read_delta_length_byte_array(r, vals)
                                                     : orc_read_delta_byte_array(r, vals);
    if (rc) return rc;
    if ((uint64_t)vals->n < cnt) return ORC_PANIC; /* values[:cnt] */
    vals->n = (int64_t)cnt; vals->off.n = (int64_t)cnt + 1; vals->bytes.n = vals->off.v[cnt];
    return ORC_OK;
  } else if (encoding == ORC_ENC_BYTE_STREAM_SPLIT) {
    if (data_type == ORC_FLOAT || data_type == ORC_DOUBLE) return orc_read_byte_stream_split(r, data_type, cnt, vals);
    return ORC_UNSUPPORTED;
  }
  return ORC_UNSUPPORTED; /* "Unknown Encoding method" */
}

static int32_t bits_len32(uint32_t x) { int32_t n = 0; while (x) { n++; x >>= 1; } return n; }

/* element size of the typed output for a physical type */
static int32_t elem_size_of(const orc_column_desc *col) {
  switch (col->physical_type) {
    case ORC_BOOLEAN: return 1;
    case ORC_INT32: case ORC_FLOAT: return 4;
    case ORC_INT64: case ORC_DOUBLE: return 8;
    case ORC_INT96: return 12;
    case ORC_FIXED_LEN_BYTE_ARRAY: return col->type_length;
    default: return 0;
  }
}

/* One data page: layout/page.go:736-784 (v2 re-framing), :816-931 (levels,
 * null count, values, scatter, row count).  Appends to the table. */
static int read_data_page(const uint8_t *body, int64_t body_len, const orc_page_desc *pg, const orc_column_desc *col,
                          const orc_col *dict, orc_table *t, int64_t *cap_slots) {
  int rc = ORC_OK;
  orc_buf framed = {0, 0, 0};
  orc_rd r;
  if (pg->page_type == ORC_DATA_PAGE_V2) { /* :736-773 */
    int64_t rll = pg->rep_len, dll = pg->def_len;
    if (rll < 0 || dll < 0 || body_len - rll - dll < 0) return ORC_PANIC;
    if (rll > 0) { int32_t v = (int32_t)rll; buf_append(&framed, &v, 4); buf_append(&framed, body, rll); }
    if (dll > 0) { int32_t v = (int32_t)dll; buf_append(&framed, &v, 4); buf_append(&framed, body + rll, dll); }
    buf_append(&framed, body + rll + dll, body_len - rll - dll);
    r.s = framed.b; r.n = framed.n; r.i = 0;
  } else {
    r.s = body; r.n = body_len; r.i = 0;
  }
  uint64_t num_values = (uint64_t)(int64_t)pg->num_values;
  if ((int64_t)num_values < 0) { orc_buf_free(&framed); return ORC_PANIC; }

  orc_i64v rep = {0, 0, 0}, def = {0, 0, 0}, idx = {0, 0, 0};
  orc_col vals; memset(&vals, 0, sizeof vals);
  int is_idx = 0;
  orc_col dummy; memset(&dummy, 0, sizeof dummy);

  if (col->max_rep > 0) { /* :836-857 */
    rc = orc_read_data_page_values(&r, ORC_ENC_RLE, ORC_INT64, num_values, (uint64_t)bits_len32((uint32_t)col->max_rep), &rep, &dummy, &is_idx);
    if (rc) goto done;
  } else {
    if ((rc = i64v_reserve(&rep, (int64_t)num_values + 1))) goto done;
    memset(rep.v, 0, (size_t)num_values * 8); rep.n = (int64_t)num_values;
  }
  if (col->max_def > 0) { /* :859-881 */
    rc = orc_read_data_page_values(&r, ORC_ENC_RLE, ORC_INT64, num_values, (uint64_t)bits_len32((uint32_t)col->max_def), &def, &dummy, &is_idx);
    if (rc) goto done;
  } else {
    if ((rc = i64v_reserve(&def, (int64_t)num_values + 1))) goto done;
    memset(def.v, 0, (size_t)num_values * 8); def.n = (int64_t)num_values;
  }
  /* with num_values == 0 ReadDataPageValues returns early and both vectors are empty */
  uint64_t num_nulls = 0; /* :883-888 */
  for (int64_t i = 0; i < def.n; i++) if ((int32_t)def.v[i] != col->max_def) num_nulls++;

  rc = orc_read_data_page_values(&r, pg->encoding, col->physical_type, (uint64_t)def.n - num_nulls,
                                 (uint64_t)(int64_t)col->type_length, &idx, &vals, &is_idx); /* :895-903 */
  if (rc) goto done;

  { /* :905-931 scatter + Decode (:139-164) + Merge (layout/table.go:52-71) */
    int is_dict_enc = pg->encoding == ORC_ENC_PLAIN_DICTIONARY || pg->encoding == ORC_ENC_RLE_DICTIONARY;
    int64_t n = def.n, base = t->n_slots;
    if (base + n > *cap_slots) {
      int64_t nc = *cap_slots ? *cap_slots : 1024;
      while (nc < base + n) nc *= 2;
      t->def = (int32_t *)realloc(t->def, (size_t)nc * 4);
      t->rep = (int32_t *)realloc(t->rep, (size_t)nc * 4);
      if (!t->def || !t->rep) { rc = ORC_NOMEM; goto done; }
      *cap_slots = nc;
    }
    int32_t esz = t->values.elem_size;
    uint8_t zero[16]; memset(zero, 0, sizeof zero);
    int64_t j = 0;
    int64_t avail = is_idx ? idx.n : vals.n;
    for (int64_t i = 0; i < n; i++) {
      t->rep[base + i] = (int32_t)rep.v[i];
      t->def[base + i] = (int32_t)def.v[i];
      if (t->rep[base + i] == 0) t->n_rows++;
      if (t->def[base + i] == col->max_def) {
        if (j >= avail) { rc = ORC_PANIC; goto done; } /* values[j] out of range */
        if (is_idx) {
          if (!is_dict_enc) {
            /* Encoding_RLE values (:657-667): the numbers are the values, int32(value) for INT32.  The typed image
             * exists for BOOLEAN (low byte), INT32 and INT64; for any other type the reference holds int64 numbers
             * where values of that type belong, which has no typed counterpart. */
            int64_t v = idx.v[j];
            if (col->physical_type == ORC_BOOLEAN) { uint8_t b = (uint8_t)v; rc = col_push_fixed(&t->values, &b); }
            else if (col->physical_type == ORC_INT32) { int32_t w = (int32_t)v; rc = col_push_fixed(&t->values, &w); }
            else if (col->physical_type == ORC_INT64) rc = col_push_fixed(&t->values, &v);
            else rc = ORC_UNSUPPORTED_TYPE;
            if (rc) goto done;
            j++;
            continue;
          }
          if (!dict) { rc = ORC_UNSUPPORTED; goto done; }        /* Decode() leaves raw indices: treated as an error */
          int64_t k = idx.v[j];
          if (k < 0 || k >= dict->n) { rc = ORC_PANIC; goto done; } /* dictPage.DataTable.Values[index] */
          if (t->values.kind == 0) rc = col_push_fixed(&t->values, dict->fixed.b + k * esz);
          else rc = col_push_str(&t->values, dict->bytes.b + dict->off.v[k], dict->off.v[k + 1] - dict->off.v[k]);
        } else {
          if (t->values.kind == 0 && vals.kind == 1 && col->physical_type == ORC_FIXED_LEN_BYTE_ARRAY) {
            /* DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY pages of a FIXED_LEN_BYTE_ARRAY column (:684-706): the
             * reference hands the decoded strings on as they are.  The typed image of the column has type_length
             * bytes per value, so a string of any other length has no place in it. */
            int64_t ln = vals.off.v[j + 1] - vals.off.v[j];
            if (ln != esz) { rc = ORC_FLBA_LENGTH; goto done; }
            rc = col_push_fixed(&t->values, vals.bytes.b + vals.off.v[j]);
          } else if (t->values.kind == 0) {
            if (vals.kind != 0 || vals.elem_size != esz) { rc = ORC_UNSUPPORTED; goto done; }
            rc = col_push_fixed(&t->values, vals.fixed.b + j * esz);
          } else {
            if (vals.kind != 1) { rc = ORC_UNSUPPORTED; goto done; }
            rc = col_push_str(&t->values, vals.bytes.b + vals.off.v[j], vals.off.v[j + 1] - vals.off.v[j]);
          }
        }
        j++;
      } else {
        t->n_nulls++;
        if (t->values.kind == 0) {
          t->values.n++;
          rc = buf_reserve(&t->values.fixed, t->values.fixed.n + esz);
          if (rc == ORC_OK) { memset(t->values.fixed.b + t->values.fixed.n, 0, esz); t->values.fixed.n += esz; }
        } else {
          rc = col_push_str(&t->values, NULL, 0);
        }
      }
      if (rc) goto done;
    }
    t->n_slots += n;
  }
done:
  orc_i64v_free(&rep); orc_i64v_free(&def); orc_i64v_free(&idx);
  orc_col_free(&vals);
  orc_buf_free(&framed);
  return rc;
}

/* reader/columnbuffer.go:105-156 over the pages of one column chunk. */
int orc_read_chunk(const uint8_t *chunk, int64_t chunk_len, const orc_page_desc *pages, int32_t n_pages,
                   const orc_column_desc *col, orc_table *out) {
  memset(out, 0, sizeof *out);
  out->bad_page = -1;
  int rc = ORC_OK;
  if (col->physical_type == ORC_BYTE_ARRAY) rc = col_init_bytes(&out->values);
  else col_init_fixed(&out->values, elem_size_of(col));
  if (rc) { out->status = rc; return rc; }
  orc_col dict; memset(&dict, 0, sizeof dict);
  int have_dict = 0;
  int64_t cap_slots = 0;
  for (int32_t p = 0; p < n_pages; p++) {
    const orc_page_desc *pg = &pages[p];
    if (pg->offset < 0 || pg->length < 0 || pg->offset + pg->length > chunk_len) { rc = ORC_UNEXPECTED_EOF; out->bad_page = p; break; }
    const uint8_t *body = chunk + pg->offset;
    if (pg->page_type == ORC_DICTIONARY_PAGE) { /* layout/page.go:792-811 */
      if (have_dict) orc_col_free(&dict);
      orc_rd r = { body, pg->length, 0 };
      uint64_t bw = col->physical_type == ORC_FIXED_LEN_BYTE_ARRAY ? (uint64_t)(int64_t)col->type_length : 0;
      rc = orc_read_plain(&r, col->physical_type, (uint64_t)(int64_t)pg->num_values, bw, &dict);
      have_dict = 1;
    } else if (pg->page_type == ORC_DATA_PAGE || pg->page_type == ORC_DATA_PAGE_V2) {
      rc = read_data_page(body, pg->length, pg, col, have_dict ? &dict : NULL, out, &cap_slots);
    } else {
      rc = ORC_UNSUPPORTED; /* INDEX_PAGE and unknown page types, layout/page.go:813-814, :933-935 */
    }
    if (rc) { out->bad_page = p; break; }
  }
  if (have_dict) orc_col_free(&dict);
  out->status = rc;
  return rc;
}

/* ------------------------------------------------------------------ */
/* encoders (encoding/encodingwrite.go)                                */
/* ------------------------------------------------------------------ */
static int bits_len64(uint64_t x) { int n = 0; while (x) { n++; x >>= 1; } return n; }

/* encoding/encodingwrite.go:132-147 */
void orc_write_unsigned_varint(uint64_t num, orc_buf *out) {
  int byte_num = (bits_len64(num) + 6) / 7;
  if (byte_num == 0) { buf_push(out, 0); return; }
  uint64_t t = num;
  for (int i = 0; i < byte_num; i++) {
    uint8_t b = (uint8_t)(t & 0x7F) | 0x80;
    if (i == byte_num - 1) b &= 0x7F;
    buf_push(out, b);
    t >>= 7;
  }
}

/* encoding/encodingwrite.go:149-172 with pt = INT64 */
void orc_write_rle_int64(const int64_t *vals, int64_t n, int32_t bit_width, orc_buf *out) {
  int64_t i = 0;
  while (i < n) {
    int64_t j = i + 1;
    while (j < n && vals[j] == vals[i]) j++;
    orc_write_unsigned_varint((uint64_t)((j - i) << 1), out);
    int32_t byte_num = (bit_width + 7) / 8;
    uint8_t vb[8]; uint64_t u = (uint64_t)vals[i];
    for (int k = 0; k < 8; k++) vb[k] = (uint8_t)(u >> (8 * k));
    buf_append(out, vb, byte_num);
    i = j;
  }
}
/* encoding/encodingwrite.go:183-205 */
void orc_write_rle_int32(const int32_t *vals, int64_t n, int32_t bit_width, orc_buf *out) {
  int64_t i = 0;
  while (i < n) {
    int64_t j = i + 1;
    while (j < n && vals[j] == vals[i]) j++;
    orc_write_unsigned_varint((uint64_t)((j - i) << 1), out);
    int32_t byte_num = (bit_width + 7) / 8;
    uint8_t vb[4]; uint32_t u = (uint32_t)vals[i];
    for (int k = 0; k < 4; k++) vb[k] = (uint8_t)(u >> (8 * k));
    buf_append(out, vb, byte_num);
    i = j;
  }
}
/* encoding/encodingwrite.go:174-181, :207-214 */
void orc_write_rle_bit_packed_hybrid_int64(const int64_t *vals, int64_t n, int32_t bit_width, orc_buf *out) {
  orc_buf rle = {0, 0, 0};
  orc_write_rle_int64(vals, n, bit_width, &rle);
  int32_t ln = (int32_t)rle.n;
  buf_append(out, &ln, 4);
  buf_append(out, rle.b, rle.n);
  orc_buf_free(&rle);
}
void orc_write_rle_bit_packed_hybrid_int32(const int32_t *vals, int64_t n, int32_t bit_width, orc_buf *out) {
  orc_buf rle = {0, 0, 0};
  orc_write_rle_int32(vals, n, bit_width, &rle);
  int32_t ln = (int32_t)rle.n;
  buf_append(out, &ln, 4);
  buf_append(out, rle.b, rle.n);
  orc_buf_free(&rle);
}

/* encoding/encodingwrite.go:216-270 */
void orc_write_bit_packed(const int64_t *vals, int64_t ln, int64_t bit_width, int if_header, orc_buf *out) {
  if (ln <= 0) return;
  if (if_header) orc_write_unsigned_varint((uint64_t)(((ln / 8) << 1) | 1), out);
  int64_t i = 0, res_cur = 0, need = 8, used = 0, left = bit_width - used;
  int64_t val = vals[i];
  while (i < ln) {
    if (left >= need) {
      res_cur |= (int64_t)shl64((uint64_t)((val >> (used >= 64 ? 63 : used)) & (int64_t)(shl64(1, (uint64_t)need) - 1)), (uint64_t)(8 - need));
      buf_push(out, (uint8_t)res_cur);
      left -= need;
      used += need;
      need = 8;
      res_cur = 0;
      if (left <= 0 && (i + 1) < ln) {
        i += 1;
        val = vals[i];
        left = bit_width;
        used = 0;
      }
    } else {
      res_cur |= (int64_t)shl64((uint64_t)(val >> (used >= 64 ? 63 : used)), (uint64_t)(8 - need));
      i += 1;
      if (i < ln) val = vals[i];
      need -= left;
      left = bit_width;
      used = 0;
    }
  }
}

/* encoding/encodingwrite.go:287-343 */
void orc_write_delta_int32(const int32_t *nums, int64_t n, orc_buf *out) {
  const uint64_t block_size = 128, num_mb = 4, vals_in_mb = 32;
  int32_t num = nums[0];
  uint64_t first = (uint64_t)(int64_t)((num >> 31) ^ (int32_t)((uint32_t)num << 1));
  orc_write_unsigned_varint(block_size, out);
  orc_write_unsigned_varint(num_mb, out);
  orc_write_unsigned_varint((uint64_t)n, out);
  orc_write_unsigned_varint(first, out);
  int64_t i = 1;
  while (i < n) {
    int32_t block[128]; int64_t bl = 0;
    int32_t min_delta = 0x7FFFFFFF;
    while (i < n && (uint64_t)bl < block_size) {
      int32_t delta = (int32_t)((uint32_t)nums[i] - (uint32_t)nums[i - 1]);
      block[bl++] = delta;
      if (delta < min_delta) min_delta = delta;
      i++;
    }
    while ((uint64_t)bl < block_size) block[bl++] = min_delta;
    uint8_t widths[4];
    for (uint64_t j = 0; j < num_mb; j++) {
      int32_t max_value = 0;
      for (uint64_t k = j * vals_in_mb; k < (j + 1) * vals_in_mb; k++) {
        block[k] = (int32_t)((uint32_t)block[k] - (uint32_t)min_delta);
        if (block[k] > max_value) max_value = block[k];
      }
      widths[j] = (uint8_t)bits_len32((uint32_t)max_value);
    }
    uint64_t md_zz = (uint64_t)(int64_t)((min_delta >> 31) ^ (int32_t)((uint32_t)min_delta << 1));
    orc_write_unsigned_varint(md_zz, out);
    buf_append(out, widths, 4);
    for (uint64_t j = 0; j < num_mb; j++) {
      int64_t tmp[32];
      for (int k = 0; k < 32; k++) tmp[k] = block[j * vals_in_mb + k]; /* ToInt64 of int32 values */
      orc_write_bit_packed(tmp, 32, widths[j], 0, out);
    }
  }
}

/* encoding/encodingwrite.go:345-401 */
void orc_write_delta_int64(const int64_t *nums, int64_t n, orc_buf *out) {
  const uint64_t block_size = 128, num_mb = 4, vals_in_mb = 32;
  int64_t num = nums[0];
  uint64_t first = (uint64_t)((num >> 63) ^ (int64_t)((uint64_t)num << 1));
  orc_write_unsigned_varint(block_size, out);
  orc_write_unsigned_varint(num_mb, out);
  orc_write_unsigned_varint((uint64_t)n, out);
  orc_write_unsigned_varint(first, out);
  int64_t i = 1;
  while (i < n) {
    int64_t block[128]; int64_t bl = 0;
    int64_t min_delta = 0x7FFFFFFFFFFFFFFFLL;
    while (i < n && (uint64_t)bl < block_size) {
      int64_t delta = (int64_t)((uint64_t)nums[i] - (uint64_t)nums[i - 1]);
      block[bl++] = delta;
      if (delta < min_delta) min_delta = delta;
      i++;
    }
    while ((uint64_t)bl < block_size) block[bl++] = min_delta;
    uint8_t widths[4];
    for (uint64_t j = 0; j < num_mb; j++) {
      int64_t max_value = 0;
      for (uint64_t k = j * vals_in_mb; k < (j + 1) * vals_in_mb; k++) {
        block[k] = (int64_t)((uint64_t)block[k] - (uint64_t)min_delta);
        if (block[k] > max_value) max_value = block[k];
      }
      widths[j] = (uint8_t)bits_len64((uint64_t)max_value);
    }
    uint64_t md_zz = (uint64_t)((min_delta >> 63) ^ (int64_t)((uint64_t)min_delta << 1));
    orc_write_unsigned_varint(md_zz, out);
    buf_append(out, widths, 4);
    for (uint64_t j = 0; j < num_mb; j++) orc_write_bit_packed(block + j * vals_in_mb, 32, widths[j], 0, out);
  }
}

/* encoding/encodingwrite.go:403-418 */
void orc_write_delta_length_byte_array(const int64_t *off, const uint8_t *bytes, int64_t n, orc_buf *out) {
  int32_t *lens = (int32_t *)malloc((size_t)(n ? n : 1) * 4);
  for (int64_t i = 0; i < n; i++) lens[i] = (int32_t)(off[i + 1] - off[i]);
  if (n > 0) orc_write_delta_int32(lens, n, out);
  free(lens);
  buf_append(out, bytes + off[0], off[n] - off[0]);
}

/* encoding/encodingwrite.go:468-502 */
void orc_write_delta_byte_array(const int64_t *off, const uint8_t *bytes, int64_t n, orc_buf *out) {
  if (n <= 0) return;
  int32_t *prefix = (int32_t *)malloc((size_t)n * 4);
  int64_t *soff = (int64_t *)malloc((size_t)(n + 1) * 8);
  orc_buf sb = {0, 0, 0};
  prefix[0] = 0; soff[0] = 0;
  buf_append(&sb, bytes + off[0], off[1] - off[0]);
  soff[1] = sb.n;
  for (int64_t i = 1; i < n; i++) {
    const uint8_t *s1 = bytes + off[i - 1], *s2 = bytes + off[i];
    int64_t l1 = off[i] - off[i - 1], l2 = off[i + 1] - off[i], j = 0;
    while (j < l1 && j < l2) { if (s1[j] != s2[j]) break; j++; }
    prefix[i] = (int32_t)j;
    buf_append(&sb, s2 + j, l2 - j);
    soff[i + 1] = sb.n;
  }
  orc_write_delta_int32(prefix, n, out);
  orc_write_delta_length_byte_array(soff, sb.b ? sb.b : (const uint8_t *)"", n, out);
  free(prefix); free(soff); orc_buf_free(&sb);
}

/* encoding/encodingwrite.go:61-71 */
void orc_write_plain_boolean(const uint8_t *vals, int64_t n, orc_buf *out) {
  int64_t byte_num = (n + 7) / 8, base = out->n;
  buf_reserve(out, out->n + byte_num);
  memset(out->b + base, 0, (size_t)byte_num);
  out->n += byte_num;
  for (int64_t i = 0; i < n; i++) if (vals[i]) out->b[base + i / 8] |= (uint8_t)(1u << (i % 8));
}

/* encoding/encodingwrite.go:105-121 */
void orc_write_plain_byte_array(const int64_t *off, const uint8_t *bytes, int64_t n, orc_buf *out) {
  for (int64_t i = 0; i < n; i++) {
    uint32_t ln = (uint32_t)(off[i + 1] - off[i]);
    buf_append(out, &ln, 4);
    buf_append(out, bytes + off[i], ln);
  }
}
