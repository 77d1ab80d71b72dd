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

int64_t pqgen_dict_slot_bytes2(int64_t page_values, int w, int style) {
  int64_t b = page_values * (int64_t)(w ? w : 1) / 8 + page_values / 2 + 256;
  if (w == 32) b = page_values * 5 + 256;  /* parquet-go-like worst case: one 5-byte run per value */
  if (style == 1 && b < page_values * (1 + (w + 7) / 8) + 256) b = page_values * (1 + (w + 7) / 8) + 256;
  return (b + 255) / 256 * 256;
}
int64_t pqgen_dict_slot_bytes(int64_t page_values, int w) { return pqgen_dict_slot_bytes2(page_values, w, 0); }

/* Returns the number of pages (dictionary page first) or -1 if the buffer is too small.
 * style 0 = arrow-like mix, 1 = parquet-go-like (RLE runs only; repeats: each index is repeated U[1,rep_max] times) */
int32_t pqgen_dict_chunk(uint64_t seed, int64_t n_values, int w, int32_t dict_n, int64_t page_values, int style,
                         int rep_max, uint8_t *buf, int64_t cap, page_desc *pages, int32_t page_cap) {
  const int64_t n_pages = (n_values + page_values - 1) / page_values;
  const int64_t slot = pqgen_dict_slot_bytes2(page_values, w, style);
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

/* ---------------------------------------------------------------------------------------------------------------
 * DELTA_BINARY_PACKED of an arbitrary int64 sequence: block 128, 4 miniblocks, zigzag first value and minimum
 * deltas, the last block padded with zero deltas above the minimum (every block complete).
 */
static uint8_t *delta_bp_encode(const int64_t *v, int64_t n, uint8_t *p) {
  p = put_varint(p, 128);
  p = put_varint(p, 4);
  p = put_varint(p, (uint64_t)n);
  const int64_t first = n > 0 ? v[0] : 0;
  p = put_varint(p, ((uint64_t)first << 1) ^ (uint64_t)(first >> 63));
  int64_t i = 1;
  while (i < n) {
    int64_t d[128];
    int64_t k = n - i < 128 ? n - i : 128;
    int64_t mn = INT64_MAX;
    for (int64_t j = 0; j < k; j++) { d[j] = (int64_t)((uint64_t)v[i + j] - (uint64_t)v[i + j - 1]); if (d[j] < mn) mn = d[j]; }
    for (int64_t j = k; j < 128; j++) d[j] = mn;
    p = put_varint(p, ((uint64_t)mn << 1) ^ (uint64_t)(mn >> 63));
    int ws[4];
    uint64_t u[128];
    for (int m = 0; m < 4; m++) {
      uint64_t mx = 0;
      for (int j = 0; j < 32; j++) { u[m * 32 + j] = (uint64_t)d[m * 32 + j] - (uint64_t)mn; if (u[m * 32 + j] > mx) mx = u[m * 32 + j]; }
      ws[m] = mx ? 64 - __builtin_clzll(mx) : 0;
      *p++ = (uint8_t)ws[m];
    }
    for (int m = 0; m < 4; m++) {
      packer pk = {p, 0, 0};
      for (int j = 0; j < 32; j++) pk_put(&pk, u[m * 32 + j], ws[m]);
      p = pk.p;
    }
    i += k;
  }
  return p;
}

/* RLE / bit-packed hybrid of small integers the way Arrow-style writers lay it out: a run of >= 8 equal values
 * that starts on a group boundary becomes an RLE run, everything else goes into bit-packed groups of 8
 * (at most 63 groups per run header; the last group zero padded). */
static uint8_t *hybrid_encode(const uint8_t *lv, int64_t n, int w, uint8_t *p) {
  const int nb = (w + 7) / 8;
  int64_t i = 0;
  while (i < n) {
    int64_t r = 1;
    while (i + r < n && lv[i + r] == lv[i]) r++;
    if (r >= 8) {
      p = put_varint(p, (uint64_t)r << 1);
      for (int k = 0; k < nb; k++) *p++ = (uint8_t)(lv[i] >> (8 * k));
      i += r;
      continue;
    }
    /* bit-packed groups until a long run starts on a group boundary, or 63 groups */
    int64_t g = 0, j = i;
    while (g < 63 && j < n) {
      int64_t rr = 1;
      while (j + rr < n && lv[j + rr] == lv[j] && rr < 8) rr++;
      if (rr >= 8 && g > 0) break;
      j += 8;
      g++;
    }
    p = put_varint(p, ((uint64_t)g << 1) | 1);
    packer pk = {p, 0, 0};
    for (int64_t q = 0; q < g * 8; q++) pk_put(&pk, i + q < n ? lv[i + q] : 0, w);
    p = pk.p;
    i += g * 8;
  }
  return p;
}

/* ---------------------------------------------------------------------------------------------------------------
 * list<int64> chunk: optional group (LIST) { repeated group list { optional int64 element } }: max_def 3, max_rep 1.
 * Per row: 10 % null list (def 0), 10 % empty list (def 1), else 1 + Geom(mean 3) elements, each null (def 2) with
 * probability 10 % else present (def 3); rep 0 at the row start, 1 inside.  v1 pages of page_slots level slots:
 * [i32 len][rep runs][i32 len][def runs][PLAIN INT64 values].
 */
int64_t pqgen_list_slot_bytes(int64_t page_slots) { return (page_slots * 8 + page_slots + 1024 + 255) / 256 * 256; }

int32_t pqgen_list_chunk(uint64_t seed, int64_t n_slots, int64_t page_slots, uint8_t *buf, int64_t cap, page_desc *pages,
                         int32_t page_cap) {
  const int64_t n_pages = (n_slots + page_slots - 1) / page_slots;
  const int64_t slot = pqgen_list_slot_bytes(page_slots);
  if (n_pages > page_cap || n_pages * slot > cap) return -1;
  int failed = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t pgi = 0; pgi < n_pages; pgi++) {
    uint64_t s = seed * 0x100000001B3ull + (uint64_t)pgi * 0x9E3779B97F4A7C15ull + 4242;
    const int64_t nv = (pgi == n_pages - 1) ? n_slots - pgi * page_slots : page_slots;
    uint8_t *dl = (uint8_t *)malloc((size_t)nv), *rl = (uint8_t *)malloc((size_t)nv);
    if (!dl || !rl) { failed = 1; free(dl); free(rl); continue; }
    int64_t i = 0;
    while (i < nv) {  /* every page starts a row */
      const uint64_t kind = below(&s, 10);
      if (kind == 0) { dl[i] = 0; rl[i] = 0; i++; continue; }
      if (kind == 1) { dl[i] = 1; rl[i] = 0; i++; continue; }
      int64_t len = 1;
      while (below(&s, 4) != 0 && len < 64) len++;
      for (int64_t k = 0; k < len && i < nv; k++, i++) { dl[i] = below(&s, 10) == 0 ? 2 : 3; rl[i] = k ? 1 : 0; }
    }
    uint8_t *base = buf + pgi * slot, *p = base;
    uint8_t *q = hybrid_encode(rl, nv, 1, p + 4);
    uint32_t L = (uint32_t)(q - (p + 4));
    memcpy(p, &L, 4);
    p = q;
    q = hybrid_encode(dl, nv, 2, p + 4);
    L = (uint32_t)(q - (p + 4));
    memcpy(p, &L, 4);
    p = q;
    for (int64_t k = 0; k < nv; k++)
      if (dl[k] == 3) { uint64_t v = splitmix(&s); memcpy(p, &v, 8); p += 8; }
    free(dl); free(rl);
    pages[pgi] = (page_desc){0, 0, (int32_t)nv, 0, 0, 0, (int64_t)(base - buf), (int64_t)(p - base)};
  }
  return failed ? -2 : (int32_t)n_pages;
}

/* ---------------------------------------------------------------------------------------------------------------
 * BYTE_ARRAY chunk of key-like strings: length U[8,64]; each string shares a prefix of U[0, len-1] bytes (capped by
 * the predecessor's length) with its predecessor inside the page and continues with random [a-z0-9].
 * mode 0: DELTA_LENGTH_BYTE_ARRAY (lengths, then the strings); mode 1: DELTA_BYTE_ARRAY (prefix lengths, then the
 * suffixes as DELTA_LENGTH_BYTE_ARRAY); mode 2: PLAIN (u32 length + bytes); mode 3: DELTA_BYTE_ARRAY over sorted keys.
 */
int64_t pqgen_strings_slot_bytes(int64_t page_values) { return (page_values * 70 + 4096 + 255) / 256 * 256; }

int32_t pqgen_strings_chunk(uint64_t seed, int64_t n_values, int64_t page_values, int mode, uint8_t *buf, int64_t cap,
                            page_desc *pages, int32_t page_cap) {
  static const char alpha[] = "abcdefghijklmnopqrstuvwxyz0123456789";
  const int64_t n_pages = (n_values + page_values - 1) / page_values;
  const int64_t slot = pqgen_strings_slot_bytes(page_values);
  if (n_pages > page_cap || n_pages * slot > cap) return -1;
  int failed = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t pgi = 0; pgi < n_pages; pgi++) {
    uint64_t s = seed * 0x100000001B3ull + (uint64_t)pgi * 0x9E3779B97F4A7C15ull + 999;
    const int64_t nv = (pgi == n_pages - 1) ? n_values - pgi * page_values : page_values;
    int64_t *lens = (int64_t *)malloc(8 * (size_t)nv), *pre = (int64_t *)malloc(8 * (size_t)nv);
    uint8_t *str = (uint8_t *)malloc(64 * (size_t)nv);
    if (!lens || !pre || !str) { failed = 1; free(lens); free(pre); free(str); continue; }
    int64_t prev_len = 0;
    if (mode == 3) {
      /* sorted keys: a 16-digit zero-padded counter that grows by U[1, 1000] from key to key, then U[0, 48] characters;
       * every key shares its leading digits with its predecessor, so no prefix length inside a page is 0 */
      uint64_t ctr = (uint64_t)pgi * (uint64_t)page_values * 1000ull;
      for (int64_t i = 0; i < nv; i++) {
        ctr += 1 + below(&s, 1000);
        const int64_t len = 16 + (int64_t)below(&s, 49);
        uint8_t *d = str + 64 * i;
        uint64_t c = ctr;
        for (int k = 15; k >= 0; k--) { d[k] = (uint8_t)('0' + c % 10); c /= 10; }
        for (int64_t k = 16; k < len; k++) d[k] = (uint8_t)alpha[below(&s, 36)];
        int64_t pl = 0;
        if (i > 0) { const uint8_t *q = str + 64 * (i - 1); while (pl < len && pl < prev_len && q[pl] == d[pl]) pl++; }
        lens[i] = len; pre[i] = pl; prev_len = len;
      }
    }
    for (int64_t i = 0; i < nv && mode != 3; i++) {
      const int64_t len = 8 + (int64_t)below(&s, 57);
      int64_t pl = i == 0 ? 0 : (int64_t)below(&s, (uint64_t)len);
      if (pl > prev_len) pl = prev_len;
      uint8_t *d = str + 64 * i;
      if (pl) memcpy(d, str + 64 * (i - 1), (size_t)pl);
      for (int64_t k = pl; k < len; k++) d[k] = (uint8_t)alpha[below(&s, 36)];
      lens[i] = len; pre[i] = pl; prev_len = len;
    }
    uint8_t *base = buf + pgi * slot, *p = base;
    int enc;
    if (mode == 0) {
      enc = 6;
      p = delta_bp_encode(lens, nv, p);
      for (int64_t i = 0; i < nv; i++) { memcpy(p, str + 64 * i, (size_t)lens[i]); p += lens[i]; }
    } else if (mode == 1 || mode == 3) {
      enc = 7;
      p = delta_bp_encode(pre, nv, p);
      for (int64_t i = 0; i < nv; i++) lens[i] -= pre[i];
      p = delta_bp_encode(lens, nv, p);
      for (int64_t i = 0; i < nv; i++) { memcpy(p, str + 64 * i + pre[i], (size_t)lens[i]); p += lens[i]; }
    } else {
      enc = 0;
      for (int64_t i = 0; i < nv; i++) { uint32_t l = (uint32_t)lens[i]; memcpy(p, &l, 4); memcpy(p + 4, str + 64 * i, l); p += 4 + l; }
    }
    free(lens); free(pre); free(str);
    pages[pgi] = (page_desc){0, enc, (int32_t)nv, 0, 0, 0, (int64_t)(base - buf), (int64_t)(p - base)};
  }
  return failed ? -2 : (int32_t)n_pages;
}
