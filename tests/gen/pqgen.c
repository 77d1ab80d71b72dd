/*
 * Synthetic column-chunk generator for the benchmark and the large-size tests (test infrastructure).
 * Writes spec-conformant UNCOMPRESSED page bodies straight into a caller-provided buffer, one slot per
 * page, pages generated in parallel (OpenMP), every page seeded by (seed, page index) so the output does not
 * depend on the thread count.
 *
 *  - dictionary chunks: a PLAIN dictionary page + v1 data pages whose value section is
 *    [bit width][RLE / bit-packed hybrid runs]; run mix "arrow-like": ~90 % of the values in bit-packed runs of
 *    8*U[1,63] values, ~10 % in RLE runs of U[8,512] values; or "parquet-go-like": width 32, RLE runs only
 *    (what layout/dictpage.go:155-158 + encoding/encodingwrite.go:183-205 of the reference emit).
 *  - delta chunks: DELTA_BINARY_PACKED pages, block 128 / 4 miniblocks, every block complete (the layout
 *    encoding/encodingwrite.go:345-401 writes), t_i = t_{i-1} + base_step + U[0, 2^w).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int32_t page_type, encoding, num_values, rep_len, def_len, reserved;
  int64_t offset, length;
} page_desc;

static inline uint64_t splitmix(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t below(uint64_t *s, uint64_t n) { return splitmix(s) % n; }

static inline uint8_t *put_varint(uint8_t *p, uint64_t v) {
  while (v >= 0x80) { *p++ = (uint8_t)(v | 0x80); v >>= 7; }
  *p++ = (uint8_t)v;
  return p;
}

/* appends `n` values of width w (LSB first); n*w must be a multiple of 8 */
typedef struct { uint8_t *p; unsigned __int128 acc; int bits; } packer;
static inline void pk_put(packer *k, uint64_t v, int w) {
  if (w == 0) return;
  k->acc |= (unsigned __int128)v << k->bits;
  k->bits += w;
  while (k->bits >= 8) { *k->p++ = (uint8_t)k->acc; k->acc >>= 8; k->bits -= 8; }
}

int64_t pqgen_dict_slot_bytes(int64_t page_values, int w) {
  int64_t b = page_values * (int64_t)(w ? w : 1) / 8 + page_values / 2 + 256;
  if (w == 32) b = page_values * 5 + 256;  /* parquet-go-like worst case: one 5-byte run per value */
  return (b + 255) / 256 * 256;
}

/* Returns the number of pages (dictionary page first) or -1 if the buffer is too small.
 * style 0 = arrow-like mix, 1 = parquet-go-like (RLE runs only; repeats: each index is repeated U[1,rep_max] times) */
int32_t pqgen_dict_chunk(uint64_t seed, int64_t n_values, int w, int32_t dict_n, int64_t page_values, int style,
                         int rep_max, uint8_t *buf, int64_t cap, page_desc *pages, int32_t page_cap) {
  const int64_t n_pages = (n_values + page_values - 1) / page_values;
  const int64_t slot = pqgen_dict_slot_bytes(page_values, w);
  const int64_t dict_bytes = ((int64_t)dict_n * 4 + 255) / 256 * 256;
  if (n_pages + 1 > page_cap || dict_bytes + n_pages * slot > cap) return -1;
  /* dictionary: random int32 */
  {
    uint64_t s = seed ^ 0xD1C7ull;
    uint32_t *d = (uint32_t *)buf;
    for (int32_t i = 0; i < dict_n; i++) d[i] = (uint32_t)splitmix(&s);
    pages[0] = (page_desc){2, 0, dict_n, 0, 0, 0, 0, (int64_t)dict_n * 4};
  }
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t pgi = 0; pgi < n_pages; pgi++) {
    uint64_t s = seed * 0x100000001B3ull + (uint64_t)pgi * 0x9E3779B97F4A7C15ull + 12345;
    const int64_t nv = (pgi == n_pages - 1) ? n_values - pgi * page_values : page_values;
    uint8_t *base = buf + dict_bytes + pgi * slot, *p = base;
    *p++ = (uint8_t)w;
    const uint64_t dn = (uint64_t)dict_n;
    const int nb = (w + 7) / 8;
    int64_t done = 0;
    while (done < nv) {
      if (style == 1) {
        uint64_t v = below(&s, dn);
        int64_t c = 1 + (int64_t)below(&s, (uint64_t)rep_max);
        if (c > nv - done) c = nv - done;
        p = put_varint(p, (uint64_t)c << 1);
        for (int k = 0; k < nb; k++) *p++ = (uint8_t)(v >> (8 * k));
        done += c;
      } else if (below(&s, 10) == 0) {
        int64_t c = 8 + (int64_t)below(&s, 505);
        if (c > nv - done) c = nv - done;
        uint64_t v = below(&s, dn);
        p = put_varint(p, (uint64_t)c << 1);
        for (int k = 0; k < nb; k++) *p++ = (uint8_t)(v >> (8 * k));
        done += c;
      } else {
        int64_t g = 1 + (int64_t)below(&s, 63);
        if (g * 8 > nv - done) g = (nv - done + 7) / 8;
        p = put_varint(p, ((uint64_t)g << 1) | 1);
        packer k = {p, 0, 0};
        for (int64_t i = 0; i < g * 8; i++) pk_put(&k, below(&s, dn), w);
        p = k.p;
        done += g * 8;
      }
    }
    pages[pgi + 1] = (page_desc){0, 8, (int32_t)nv, 0, 0, 0, (int64_t)(base - buf), (int64_t)(p - base)};
  }
  return (int32_t)(n_pages + 1);
}

int64_t pqgen_delta_slot_bytes(int64_t page_values, int w) {
  int64_t blocks = (page_values + 127) / 128;
  int64_t b = 64 + blocks * (16 + 16 * (int64_t)(w + 1));
  return (b + 255) / 256 * 256;
}

/* elem_size 8: INT64, 4: INT32.  Values: v_0 = start + page-dependent offset, v_i = v_{i-1} + base_step + U[0, 2^w). */
int32_t pqgen_delta_chunk(uint64_t seed, int64_t n_values, int w, int64_t page_values, int elem_size, int64_t start,
                          int64_t base_step, uint8_t *buf, int64_t cap, page_desc *pages, int32_t page_cap) {
  const int64_t n_pages = (n_values + page_values - 1) / page_values;
  const int64_t slot = pqgen_delta_slot_bytes(page_values, w);
  if (n_pages > page_cap || n_pages * slot > cap) return -1;
  const uint64_t range = w >= 64 ? 0 : ((uint64_t)1 << w);
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t pgi = 0; pgi < n_pages; pgi++) {
    uint64_t s = seed * 0x100000001B3ull + (uint64_t)pgi * 0x9E3779B97F4A7C15ull + 777;
    const int64_t nv = (pgi == n_pages - 1) ? n_values - pgi * page_values : page_values;
    uint8_t *base = buf + pgi * slot, *p = base;
    uint64_t first = (uint64_t)start + (uint64_t)pgi * (uint64_t)page_values * (uint64_t)(base_step + (int64_t)(range / 2));
    if (elem_size == 4) first &= 0x7FFFFFFFull;
    p = put_varint(p, 128);
    p = put_varint(p, 4);
    p = put_varint(p, (uint64_t)nv);
    if (elem_size == 8) p = put_varint(p, (first << 1) ^ (uint64_t)((int64_t)first >> 63));
    else { uint32_t f = (uint32_t)first; p = put_varint(p, (uint32_t)((f << 1) ^ (uint32_t)((int32_t)f >> 31))); }
    int64_t left = nv - 1;
    while (left > 0) {
      uint64_t d[128];
      int64_t k = left < 128 ? left : 128;
      uint64_t mn = ~0ull;
      for (int64_t i = 0; i < k; i++) {
        d[i] = (uint64_t)base_step + (range ? splitmix(&s) % range : splitmix(&s));
        if (d[i] < mn) mn = d[i];
      }
      for (int64_t i = k; i < 128; i++) d[i] = mn;
      /* deltas are non-negative here, so the zigzag of the minimum is mn << 1 */
      p = put_varint(p, mn << 1);
      int ws[4];
      for (int m = 0; m < 4; m++) {
        uint64_t mx = 0;
        for (int i = 0; i < 32; i++) { d[m * 32 + i] -= mn; if (d[m * 32 + i] > mx) mx = d[m * 32 + i]; }
        ws[m] = mx ? 64 - __builtin_clzll(mx) : 0;
        *p++ = (uint8_t)ws[m];
      }
      for (int m = 0; m < 4; m++) {
        packer pk = {p, 0, 0};
        for (int i = 0; i < 32; i++) pk_put(&pk, d[m * 32 + i], ws[m]);
        p = pk.p;
      }
      left -= k;
    }
    pages[pgi] = (page_desc){0, 5, (int32_t)nv, 0, 0, 0, (int64_t)(base - buf), (int64_t)(p - base)};
  }
  return (int32_t)n_pages;
}
