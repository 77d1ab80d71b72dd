// Host-side walk over the page headers of one column chunk.
//
// Replaces, for the UNCOMPRESSED files this build reads, what the reference obtains per page from
// layout.ReadPageHeader (layout/page.go:625-630, thrift compact protocol over parquet.PageHeader,
// parquet/parquet.go:6329-6338) and the loop condition of ColumnBufferType.ReadPage
// (reader/columnbuffer.go:106).  The thrift runtime is a third-party module of the reference
// (github.com/apache/thrift v0.16.0, go.mod:7); this is an independent reader of the published compact
// protocol wire format, limited to the PageHeader struct (parquet/parquet.thrift: PageHeader,
// DataPageHeader, DictionaryPageHeader, DataPageHeaderV2).
#include <cstdint>
#include <cstring>
#include "../../include/pqdecode.h"

namespace {

struct Cur {
  const uint8_t *p;
  size_t n;
  size_t i;
  bool bad;
};

uint64_t varint(Cur &c) {
  uint64_t v = 0;
  int shift = 0;
  for (;;) {
    if (c.i >= c.n) { c.bad = true; return v; }
    uint8_t b = c.p[c.i++];
    if (shift < 64) v |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) break;
    shift += 7;
    if (shift > 70) { c.bad = true; break; }
  }
  return v;
}
int64_t zigzag(Cur &c) {
  uint64_t u = varint(c);
  return (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
}
void skip_bytes(Cur &c, uint64_t k) {
  if (k > c.n - c.i) { c.bad = true; c.i = c.n; return; }
  c.i += (size_t)k;
}

void skip_value(Cur &c, int type, int depth);

void skip_struct(Cur &c, int depth) {
  if (depth > 32) { c.bad = true; return; }
  int16_t last = 0;
  while (!c.bad) {
    if (c.i >= c.n) { c.bad = true; return; }
    uint8_t h = c.p[c.i++];
    if (h == 0) return;
    int type = h & 0x0F;
    int delta = h >> 4;
    if (delta == 0) last = (int16_t)zigzag(c); else last = (int16_t)(last + delta);
    skip_value(c, type, depth + 1);
  }
}

void skip_value(Cur &c, int type, int depth) {
  switch (type) {
    case 1: case 2: return;            // bool carried in the field header
    case 3: skip_bytes(c, 1); return;  // byte
    case 4: case 5: case 6: (void)varint(c); return;
    case 7: skip_bytes(c, 8); return;
    case 8: { uint64_t n = varint(c); skip_bytes(c, n); return; }
    case 9: case 10: {
      if (c.i >= c.n) { c.bad = true; return; }
      uint8_t h = c.p[c.i++];
      uint64_t n = h >> 4;
      int et = h & 0x0F;
      if (n == 15) n = varint(c);
      for (uint64_t k = 0; k < n && !c.bad; k++) {
        if (et == 1 || et == 2) skip_bytes(c, 1);  // bools inside containers take one byte
        else skip_value(c, et, depth + 1);
      }
      return;
    }
    case 11: {
      uint64_t n = varint(c);
      if (n == 0) return;
      if (c.i >= c.n) { c.bad = true; return; }
      uint8_t kv = c.p[c.i++];
      for (uint64_t k = 0; k < n && !c.bad; k++) {
        int kt = kv >> 4, vt = kv & 0x0F;
        if (kt == 1 || kt == 2) skip_bytes(c, 1); else skip_value(c, kt, depth + 1);
        if (vt == 1 || vt == 2) skip_bytes(c, 1); else skip_value(c, vt, depth + 1);
      }
      return;
    }
    case 12: skip_struct(c, depth + 1); return;
    default: c.bad = true; return;
  }
}

// Reads a struct whose interesting fields are all i32 (ids 1..8); out[id] receives them.
void read_i32_struct(Cur &c, int32_t out[9]) {
  int16_t last = 0;
  while (!c.bad) {
    if (c.i >= c.n) { c.bad = true; return; }
    uint8_t h = c.p[c.i++];
    if (h == 0) return;
    int type = h & 0x0F;
    int delta = h >> 4;
    if (delta == 0) last = (int16_t)zigzag(c); else last = (int16_t)(last + delta);
    if (type == 5 && last >= 1 && last <= 8) out[last] = (int32_t)zigzag(c);
    else skip_value(c, type, 1);
  }
}

}  // namespace

extern "C" int pqd_scan_pages(const uint8_t *chunk, size_t len, int64_t total_values, pqd_page_desc *out, int32_t cap,
                              int32_t *n_out) {
  if (!chunk || !n_out || (cap > 0 && !out)) return PQD_ERR_INVALID_ARG;
  Cur c{chunk, len, 0, false};
  int32_t n = 0;
  int64_t covered = 0;
  while (c.i < c.n && (total_values < 0 || covered < total_values)) {
    int32_t top[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
    int32_t dph[9] = {0}, dict[9] = {0}, v2[9] = {0};
    bool has_dph = false, has_dict = false, has_v2 = false;
    int16_t last = 0;
    for (;;) {
      if (c.i >= c.n) { c.bad = true; break; }
      uint8_t h = c.p[c.i++];
      if (h == 0) break;
      int type = h & 0x0F;
      int delta = h >> 4;
      if (delta == 0) last = (int16_t)zigzag(c); else last = (int16_t)(last + delta);
      if (type == 5 && last >= 1 && last <= 4) top[last] = (int32_t)zigzag(c);
      else if (type == 12 && last == 5) { read_i32_struct(c, dph); has_dph = true; }
      else if (type == 12 && last == 7) { read_i32_struct(c, dict); has_dict = true; }
      else if (type == 12 && last == 8) { read_i32_struct(c, v2); has_v2 = true; }
      else skip_value(c, type, 1);
      if (c.bad) break;
    }
    if (c.bad || top[1] < 0 || top[3] < 0) { *n_out = n; return PQD_ERR_THRIFT; }
    pqd_page_desc d;
    std::memset(&d, 0, sizeof d);
    d.page_type = top[1];
    d.offset = (int64_t)c.i;
    d.length = top[3];
    if (d.page_type == PQD_DATA_PAGE && has_dph) {
      d.num_values = dph[1]; d.encoding = dph[2];
    } else if (d.page_type == PQD_DATA_PAGE_V2 && has_v2) {
      d.num_values = v2[1]; d.encoding = v2[4]; d.def_len = v2[5]; d.rep_len = v2[6];
    } else if (d.page_type == PQD_DICTIONARY_PAGE && has_dict) {
      d.num_values = dict[1]; d.encoding = dict[2];
    } else if (d.page_type != PQD_INDEX_PAGE) {
      *n_out = n;
      return PQD_ERR_THRIFT;
    }
    if ((uint64_t)d.length > c.n - c.i) { *n_out = n; return PQD_ERR_UNEXPECTED_EOF; }  // io.ReadFull of the body
    if (n >= cap) { *n_out = n; return PQD_ERR_NOMEM; }
    out[n++] = d;
    if (d.page_type == PQD_DATA_PAGE || d.page_type == PQD_DATA_PAGE_V2) covered += d.num_values;
    c.i += (size_t)d.length;
  }
  *n_out = n;
  return PQD_OK;
}
