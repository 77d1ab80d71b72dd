// Minimal reader of the thrift compact protocol wire format (field headers, varints, zigzag, skipping of any
// value), shared by the page-header walk and the footer walk.  An independent reader of the published format;
// the reference uses github.com/apache/thrift v0.16.0 (go.mod:7) for the same bytes.
#pragma once
#include <cstddef>
#include <cstdint>

namespace pqd_thrift {

struct Cur {
  const uint8_t *p;
  size_t n;
  size_t i;
  bool bad;
};

inline uint64_t varint(Cur &c) {
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
inline int64_t zigzag(Cur &c) {
  uint64_t u = varint(c);
  return (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
}
inline void skip_bytes(Cur &c, uint64_t k) {
  if (k > c.n - c.i) { c.bad = true; c.i = c.n; return; }
  c.i += (size_t)k;
}

inline void skip_value(Cur &c, int type, int depth);

inline void skip_struct(Cur &c, int depth) {
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

inline void skip_value(Cur &c, int type, int depth) {
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


// Next field header of a struct: false at the stop byte or on error.  type: compact type code, id: field id.
inline bool next_field(Cur &c, int16_t &last, int &type) {
  if (c.bad) return false;
  if (c.i >= c.n) { c.bad = true; return false; }
  uint8_t h = c.p[c.i++];
  if (h == 0) return false;
  type = h & 0x0F;
  int delta = h >> 4;
  if (delta == 0) last = (int16_t)zigzag(c); else last = (int16_t)(last + delta);
  return true;
}

// Header of a list / set: element count and element type.
inline bool list_header(Cur &c, uint64_t &n, int &et) {
  if (c.i >= c.n) { c.bad = true; return false; }
  uint8_t h = c.p[c.i++];
  n = h >> 4;
  et = h & 0x0F;
  if (n == 15) n = varint(c);
  return !c.bad;
}

}  // namespace pqd_thrift
