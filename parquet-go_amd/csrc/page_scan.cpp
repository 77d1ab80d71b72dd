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
#include "thrift_compact.h"

namespace {

using namespace pqd_thrift;

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
    if (n >= cap) { *n_out = n; return PQD_ERR_CAPACITY; }
    out[n++] = d;
    if (d.page_type == PQD_DATA_PAGE || d.page_type == PQD_DATA_PAGE_V2) covered += d.num_values;
    c.i += (size_t)d.length;
  }
  *n_out = n;
  return PQD_OK;
}
