// Host-side walk over a Parquet footer (thrift compact FileMetaData): the leaf columns with their maximum
// definition / repetition levels, and where every column chunk of every row group starts.
//
// Replaces, for what the decode path needs, ParquetReader.ReadFooter (reader/reader.go:148-176), the level
// bookkeeping of schema.SchemaHandler (schema/schemahandler.go:101-135: one definition level per non-REQUIRED
// ancestor-or-self, one repetition level per REPEATED one) and the chunk start of ColumnBufferType.NextRowGroup
// (reader/columnbuffer.go:89-99: the dictionary page offset when there is one, else the data page offset).
// Struct layouts: parquet/parquet.thrift FileMetaData (:999), SchemaElement (:353), RowGroup (:812),
// ColumnChunk (:778), ColumnMetaData (:714).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/pqdecode.h"
#include "thrift_compact.h"

using namespace pqd_thrift;

struct pqd_footer {
  int64_t num_rows = 0;
  int32_t version = 0;
  int32_t n_row_groups = 0;
  std::vector<pqd_leaf_desc> leaves;
  std::vector<pqd_chunk_desc> chunks;
};

namespace {

struct Elem {
  int32_t type = -1, type_length = 0, repetition = 0, num_children = 0;
  std::string name;
};

std::string read_string(Cur &c) {
  uint64_t n = varint(c);
  if (c.bad || n > c.n - c.i) { c.bad = true; return std::string(); }
  std::string s(reinterpret_cast<const char *>(c.p + c.i), (size_t)n);
  c.i += (size_t)n;
  return s;
}

void read_schema_element(Cur &c, Elem &e) {
  int16_t last = 0;
  int type;
  while (next_field(c, last, type)) {
    if (type == 5 && last == 1) e.type = (int32_t)zigzag(c);
    else if (type == 5 && last == 2) e.type_length = (int32_t)zigzag(c);
    else if (type == 5 && last == 3) e.repetition = (int32_t)zigzag(c);
    else if (type == 8 && last == 4) e.name = read_string(c);
    else if (type == 5 && last == 5) e.num_children = (int32_t)zigzag(c);
    else skip_value(c, type, 1);
  }
}

struct ColMeta {
  int32_t type = -1, codec = 0;
  int64_t num_values = 0, total_uncompressed = 0, total_compressed = 0, data_page_offset = 0, dict_page_offset = -1;
  std::vector<std::string> path;
  bool present = false;
};

void read_column_meta(Cur &c, ColMeta &m) {
  m.present = true;
  int16_t last = 0;
  int type;
  while (next_field(c, last, type)) {
    if (type == 5 && last == 1) m.type = (int32_t)zigzag(c);
    else if (type == 9 && last == 3) {
      uint64_t n; int et;
      if (!list_header(c, n, et)) return;
      for (uint64_t k = 0; k < n && !c.bad; k++) {
        if (et == 8) m.path.push_back(read_string(c)); else skip_value(c, et, 2);
      }
    }
    else if (type == 5 && last == 4) m.codec = (int32_t)zigzag(c);
    else if (type == 6 && last == 5) m.num_values = zigzag(c);
    else if (type == 6 && last == 6) m.total_uncompressed = zigzag(c);
    else if (type == 6 && last == 7) m.total_compressed = zigzag(c);
    else if (type == 6 && last == 9) m.data_page_offset = zigzag(c);
    else if (type == 6 && last == 11) m.dict_page_offset = zigzag(c);
    else skip_value(c, type, 1);
  }
}

// Depth-first walk of the flattened schema: elems[idx] with its num_children following it.
void walk_schema(const std::vector<Elem> &elems, size_t &idx, int def, int rep, const std::string &prefix, int depth,
                 std::vector<pqd_leaf_desc> &leaves, bool &bad) {
  if (idx >= elems.size() || depth > 64) { bad = true; return; }
  const size_t me = idx++;
  const Elem &e = elems[me];
  if (depth > 0) {  // the root carries no levels (the reference starts at path[:2])
    if (e.repetition != 0) def++;        // FieldRepetitionType: REQUIRED = 0, OPTIONAL = 1, REPEATED = 2
    if (e.repetition == 2) rep++;
  }
  const std::string path = depth == 0 ? std::string() : (prefix.empty() ? e.name : prefix + "." + e.name);
  if (e.num_children <= 0) {
    if (depth == 0) return;  // a schema with only a root
    pqd_leaf_desc l;
    std::memset(&l, 0, sizeof l);
    l.physical_type = e.type; l.type_length = e.type_length; l.max_def = def; l.max_rep = rep;
    l.schema_index = (int32_t)me;
    std::strncpy(l.path, path.c_str(), sizeof(l.path) - 1);
    leaves.push_back(l);
    return;
  }
  for (int k = 0; k < e.num_children && !bad; k++) walk_schema(elems, idx, def, rep, path, depth + 1, leaves, bad);
}

}  // namespace

extern "C" int pqd_footer_locate(const uint8_t *tail8, size_t len, int64_t *footer_len) {
  if (!tail8 || !footer_len || len < 8) return PQD_ERR_INVALID_ARG;
  if (std::memcmp(tail8 + 4, "PAR1", 4) != 0) return PQD_ERR_THRIFT;
  *footer_len = (int64_t)((uint32_t)tail8[0] | (uint32_t)tail8[1] << 8 | (uint32_t)tail8[2] << 16 | (uint32_t)tail8[3] << 24);
  return PQD_OK;
}

extern "C" int pqd_footer_parse(const uint8_t *footer, size_t len, pqd_footer **out) {
  if (!footer || !out) return PQD_ERR_INVALID_ARG;
  *out = nullptr;
  Cur c{footer, len, 0, false};
  pqd_footer *f = new pqd_footer();
  std::vector<Elem> elems;
  struct Rg { std::vector<ColMeta> cols; int64_t num_rows = 0; };
  std::vector<Rg> rgs;
  int16_t last = 0;
  int type;
  while (next_field(c, last, type)) {
    if (type == 5 && last == 1) f->version = (int32_t)zigzag(c);
    else if (type == 9 && last == 2) {
      uint64_t n; int et;
      if (!list_header(c, n, et)) break;
      for (uint64_t k = 0; k < n && !c.bad; k++) {
        if (et != 12) { skip_value(c, et, 1); continue; }
        Elem e;
        read_schema_element(c, e);
        elems.push_back(e);
      }
    }
    else if (type == 6 && last == 3) f->num_rows = zigzag(c);
    else if (type == 9 && last == 4) {
      uint64_t n; int et;
      if (!list_header(c, n, et)) break;
      for (uint64_t k = 0; k < n && !c.bad; k++) {
        if (et != 12) { skip_value(c, et, 1); continue; }
        Rg rg;
        int16_t l2 = 0;
        int t2;
        while (next_field(c, l2, t2)) {
          if (t2 == 9 && l2 == 1) {
            uint64_t nc; int ect;
            if (!list_header(c, nc, ect)) break;
            for (uint64_t q = 0; q < nc && !c.bad; q++) {
              if (ect != 12) { skip_value(c, ect, 2); continue; }
              ColMeta m;
              int16_t l3 = 0;
              int t3;
              while (next_field(c, l3, t3)) {
                if (t3 == 12 && l3 == 3) read_column_meta(c, m);
                else skip_value(c, t3, 3);
              }
              rg.cols.push_back(m);
            }
          }
          else if (t2 == 6 && l2 == 3) rg.num_rows = zigzag(c);
          else skip_value(c, t2, 2);
        }
        rgs.push_back(rg);
      }
    }
    else skip_value(c, type, 1);
  }
  if (c.bad || elems.empty()) { delete f; return PQD_ERR_THRIFT; }
  size_t idx = 0;
  bool bad = false;
  walk_schema(elems, idx, 0, 0, std::string(), 0, f->leaves, bad);
  if (bad) { delete f; return PQD_ERR_THRIFT; }
  f->n_row_groups = (int32_t)rgs.size();
  for (size_t r = 0; r < rgs.size(); r++) {
    for (size_t q = 0; q < rgs[r].cols.size(); q++) {
      const ColMeta &m = rgs[r].cols[q];
      if (!m.present) { delete f; return PQD_ERR_THRIFT; }  // the reference dereferences MetaData without a check
      // the leaf this chunk belongs to: by path, as NextRowGroup does (columnbuffer.go:68-82)
      std::string path;
      for (size_t k = 0; k < m.path.size(); k++) path += (k ? "." : "") + m.path[k];
      int leaf = -1;
      if (q < f->leaves.size() && path == f->leaves[q].path) leaf = (int)q;
      else for (size_t k = 0; k < f->leaves.size(); k++) if (path == f->leaves[k].path) { leaf = (int)k; break; }
      if (leaf < 0) { delete f; return PQD_ERR_THRIFT; }
      pqd_chunk_desc d;
      std::memset(&d, 0, sizeof d);
      d.row_group = (int32_t)r; d.leaf = leaf; d.codec = m.codec;
      d.offset = m.dict_page_offset >= 0 ? m.dict_page_offset : m.data_page_offset;  // columnbuffer.go:89-93
      if (m.dict_page_offset == 0) d.offset = m.data_page_offset;  // writers that store 0 for "none"
      d.total_compressed_size = m.total_compressed;
      d.num_values = m.num_values;
      d.num_rows = rgs[r].num_rows;
      f->chunks.push_back(d);
    }
  }
  *out = f;
  return PQD_OK;
}

extern "C" void pqd_footer_free(pqd_footer *f) { delete f; }
extern "C" int64_t pqd_footer_num_rows(const pqd_footer *f) { return f ? f->num_rows : 0; }
extern "C" int32_t pqd_footer_num_row_groups(const pqd_footer *f) { return f ? f->n_row_groups : 0; }
extern "C" int32_t pqd_footer_num_leaves(const pqd_footer *f) { return f ? (int32_t)f->leaves.size() : 0; }
extern "C" int32_t pqd_footer_num_chunks(const pqd_footer *f) { return f ? (int32_t)f->chunks.size() : 0; }
extern "C" int pqd_footer_leaf(const pqd_footer *f, int32_t i, pqd_leaf_desc *out) {
  if (!f || !out || i < 0 || (size_t)i >= f->leaves.size()) return PQD_ERR_INVALID_ARG;
  *out = f->leaves[(size_t)i];
  return PQD_OK;
}
extern "C" int pqd_footer_chunk(const pqd_footer *f, int32_t i, pqd_chunk_desc *out) {
  if (!f || !out || i < 0 || (size_t)i >= f->chunks.size()) return PQD_ERR_INVALID_ARG;
  *out = f->chunks[(size_t)i];
  return PQD_OK;
}
