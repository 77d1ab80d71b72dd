// PLAIN encoding on gfx950: fixed-width reinterpretation copies, BOOLEAN bit unpack, dictionary page
// materialisation and the null scatter (dense values -> one slot per level entry).
//
// Reference behaviour reproduced here: encoding/encodingread.go:13-125, encoding/binaryread.go:10-92
// (PLAIN), layout/page.go:792-811 (dictionary page), :905-931 (value scatter).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "pqd_kernels.h"

namespace pqd {

struct V16Unaligned { uint32_t v[4]; } __attribute__((packed));

__device__ __forceinline__ void set_status1(int32_t *p, int code) { atomicCAS(p, 0, code); }

// Status of reading n elements of esz bytes from a section of `len` bytes.
// INT32/INT64/FLOAT/DOUBLE use io.ReadFull (binaryread.go:12-18); INT96 / FIXED_LEN_BYTE_ARRAY read each
// element leniently and fail only when an element starts at the end (encodingread.go:76, :119).
__device__ __forceinline__ int plain_fixed_status(int ptype, int64_t len, int64_t n, int esz) {
  if (n <= 0) return PQD_OK;
  const int64_t need = n * (int64_t)esz;
  if (ptype == PQD_INT96 || ptype == PQD_FIXED_LEN_BYTE_ARRAY) {
    if (len <= (n - 1) * (int64_t)esz) return PQD_ERR_EOF;  // also esz == 0 at the end of the section
    return PQD_OK;
  }
  if (need == 0) return PQD_OK;
  if (len <= 0) return PQD_ERR_EOF;
  if (len < need) return PQD_ERR_UNEXPECTED_EOF;
  return PQD_OK;
}

// Copies `n` bytes src -> dst (both arbitrarily aligned), bytes at or beyond `src_avail` read as zero - or, with
// `prev` > 0, as the byte `prev` positions earlier: ReadPlainINT96 reads every value into the same 12-byte buffer
// (encodingread.go:74-80), so a last value cut short keeps the tail of the value before it.
// Called by a whole block; `first` is the block's first byte.
__device__ __forceinline__ void block_copy_tile(const uint8_t *src, int64_t src_avail, uint8_t *dst, int64_t n,
                                                int64_t first, int tile_bytes, int prev = 0) {
  const int tid = threadIdx.x;
  for (int o = tid * 16; o < tile_bytes; o += 256 * 16) {
    const int64_t b = first + o;
    if (b >= n) break;
    if (b + 16 <= n && b + 16 <= src_avail) {
      *reinterpret_cast<V16Unaligned *>(dst + b) = *reinterpret_cast<const V16Unaligned *>(src + b);
    } else {
      for (int k = 0; k < 16 && b + k < n; k++)
        dst[b + k] = (b + k < src_avail) ? src[b + k] : (prev > 0 && b + k >= prev && b + k - prev < src_avail ? src[b + k - prev] : (uint8_t)0);
    }
  }
}

// ---- dictionary pages, fixed-width types: PLAIN values copied to an aligned table ----
// grid.y = dictionary page of the call (dict_pages lists their page indices); the entries of page dp land at
// dict_out + pages[dp].dict_off * elem_size.  BOOLEAN dictionaries are one bit per entry, read the way
// ReadPlainBOOLEAN does (encodingread.go:35-55: io.EOF on an empty section, zero fill on a short one).
__global__ __launch_bounds__(256) void k_dict_plain_fixed(ChunkCtx c, const int32_t *__restrict__ dict_pages, uint8_t *dict_out) {
  const int dict_page = dict_pages[blockIdx.y];
  const PageIn pi = c.pages[dict_page];
  const int64_t n = pi.num_values;
  uint8_t *out = dict_out + pi.dict_off * (int64_t)c.elem_size;
  if (c.physical_type == PQD_BOOLEAN) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const int st = n < 0 ? PQD_ERR_MALFORMED : (n > 0 && pi.body_len <= 0 ? PQD_ERR_EOF : PQD_OK);
      if (st) set_status1(&c.state[dict_page].status, st);
    }
    const uint8_t *src = c.in + pi.body_off;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
      out[i] = (i >> 3) < pi.body_len ? (uint8_t)((src[i >> 3] >> (i & 7)) & 1u) : (uint8_t)0;
    return;
  }
  const int64_t bytes = n * (int64_t)c.elem_size;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int st = n < 0 ? PQD_ERR_MALFORMED : plain_fixed_status(c.physical_type, pi.body_len, n, c.elem_size);
    if (st) set_status1(&c.state[dict_page].status, st);
  }
  for (int64_t first = (int64_t)blockIdx.x * kPlainTile; first < bytes; first += (int64_t)gridDim.x * kPlainTile)
    block_copy_tile(c.in + pi.body_off, pi.body_len, out, bytes, first, kPlainTile, c.physical_type == PQD_INT96 ? 12 : 0);
}

// ---- data pages, PLAIN fixed-width values ----
struct PlainArgs {
  ChunkCtx c;
  const int32_t *tile_prefix;  // per data page, nd + 1 entries
  int n_tiles;
  uint8_t *values_out;
};

__global__ __launch_bounds__(256) void k_plain_fixed(PlainArgs a) {
  const ChunkCtx &c = a.c;
  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    int lo = 0, hi = c.nd;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (a.tile_prefix[mid] <= tile) lo = mid; else hi = mid;
    }
    const int k = tile - a.tile_prefix[lo];
    const Stream *vs = &c.streams[2 * c.nd + lo];
    const int p = vs->page;
    PageState *ps = &c.state[p];
    if (ps->status != 0 || c.pages[p].encoding != PQD_ENC_PLAIN) continue;
    const int64_t cnt = ps->cnt;
    const int64_t bytes = cnt * (int64_t)c.elem_size;
    if (k == 0 && threadIdx.x == 0) {
      int st = plain_fixed_status(c.physical_type, ps->val_len, cnt, c.elem_size);
      if (st) set_status1(&ps->status, st);
    }
    const int64_t first = (int64_t)k * kPlainTile;
    if (first >= bytes) continue;
    block_copy_tile(c.in + ps->val_off, ps->val_len, a.values_out + ps->dense_base * (int64_t)c.elem_size, bytes, first,
                    kPlainTile, c.physical_type == PQD_INT96 ? 12 : 0);
  }
}

// The same with the grid laid out by page and piece (grid.x = pieces of kPlainTile bytes, enough for the longest page;
// grid.y = data page): no search for a tile's page, and workgroups that follow each other copy pieces that follow each
// other in memory.
__global__ __launch_bounds__(256) void k_plain_fixed_pages(ChunkCtx c, uint8_t *values_out) {
  const int p = c.streams[2 * c.nd + blockIdx.y].page;
  const PageState *ps = &c.state[p];
  if (ps->status != 0 || c.pages[p].encoding != PQD_ENC_PLAIN) return;
  const int64_t cnt = ps->cnt;
  const int64_t bytes = cnt * (int64_t)c.elem_size;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int st = plain_fixed_status(c.physical_type, ps->val_len, cnt, c.elem_size);
    if (st) set_status1(&c.state[p].status, st);
  }
  const int64_t first = (int64_t)blockIdx.x * kPlainTile;
  if (first >= bytes) return;
  block_copy_tile(c.in + ps->val_off, ps->val_len, values_out + ps->dense_base * (int64_t)c.elem_size, bytes, first, kPlainTile,
                  c.physical_type == PQD_INT96 ? 12 : 0);
}

// ---- data pages, BYTE_STREAM_SPLIT FLOAT / DOUBLE: byte k of value i sits at cnt*k + i ----
// encoding/encodingread.go:437-485: one io.ReadFull of cnt*size bytes, then the byte-plane transpose.
// A thread assembles four consecutive values from one 4-byte read per plane.
template <int ESZ>
__global__ __launch_bounds__(256) void k_byte_stream_split(PlainArgs a) {
  const ChunkCtx &c = a.c;
  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    int lo = 0, hi = c.nd;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (a.tile_prefix[mid] <= tile) lo = mid; else hi = mid;
    }
    const int k = tile - a.tile_prefix[lo];
    const Stream *vs = &c.streams[2 * c.nd + lo];
    const int p = vs->page;
    PageState *ps = &c.state[p];
    if (ps->status != 0 || c.pages[p].encoding != PQD_ENC_BYTE_STREAM_SPLIT) continue;
    const int64_t cnt = ps->cnt;
    const int st = plain_fixed_status(c.physical_type, ps->val_len, cnt, ESZ);
    if (st) {
      if (k == 0 && threadIdx.x == 0) set_status1(&ps->status, st);
      continue;
    }
    const uint8_t *src = c.in + ps->val_off;
    uint8_t *dst = a.values_out + ps->dense_base * (int64_t)ESZ;
    constexpr int kTileValues = kPlainTile / ESZ;
    const int64_t v0 = (int64_t)k * kTileValues;
    for (int o = threadIdx.x * 4; o < kTileValues; o += 256 * 4) {
      const int64_t i = v0 + o;
      if (i >= cnt) break;
      if (i + 4 <= cnt) {
        uint32_t plane[ESZ];
#pragma unroll
        for (int b = 0; b < ESZ; b++) plane[b] = reinterpret_cast<const V16Unaligned *>(src + cnt * b + i)->v[0];
        uint32_t out[ESZ];  // four values of ESZ bytes = ESZ dwords
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
          for (int d = 0; d < ESZ / 4; d++) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) x |= ((plane[4 * d + b] >> (8 * j)) & 0xFFu) << (8 * b);
            out[j * (ESZ / 4) + d] = x;
          }
        }
#pragma unroll
        for (int q = 0; q < ESZ / 4; q++) {
          V16Unaligned x = {{out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]}};
          *reinterpret_cast<V16Unaligned *>(dst + i * ESZ + 16 * q) = x;
        }
      } else {
        for (int64_t j = i; j < cnt; j++)
          for (int b = 0; b < ESZ; b++) dst[j * ESZ + b] = src[cnt * b + j];
      }
    }
  }
}

// ---- data pages, PLAIN BOOLEAN: bit k of byte k/8, LSB first -> one byte per value ----
// The reference routes this through ReadBitPacked with header cnt<<1 (encodingread.go:42): an empty
// section is io.EOF, a short one is zero filled.
__global__ __launch_bounds__(256) void k_plain_boolean(PlainArgs a) {
  const ChunkCtx &c = a.c;
  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    int lo = 0, hi = c.nd;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (a.tile_prefix[mid] <= tile) lo = mid; else hi = mid;
    }
    const int k = tile - a.tile_prefix[lo];
    const Stream *vs = &c.streams[2 * c.nd + lo];
    PageState *ps = &c.state[vs->page];
    if (ps->status != 0) continue;
    const int64_t cnt = ps->cnt;
    if (cnt <= 0) continue;
    if (ps->val_len <= 0) {
      if (k == 0 && threadIdx.x == 0) set_status1(&ps->status, PQD_ERR_EOF);
      continue;
    }
    const uint8_t *src = c.in + ps->val_off;
    uint8_t *dst = a.values_out + ps->dense_base;
    const int64_t first = (int64_t)k * kPlainTile;  // values
    for (int o = threadIdx.x * 16; o < kPlainTile; o += 256 * 16) {
      const int64_t v = first + o;
      if (v >= cnt) break;
      const int64_t byte = v >> 3;
      uint32_t bits = 0;
      if (byte < ps->val_len) bits |= src[byte];
      if (byte + 1 < ps->val_len) bits |= (uint32_t)src[byte + 1] << 8;
      uint32_t o4[4];
      for (int q = 0; q < 4; q++) {
        uint32_t b4 = (bits >> (4 * q)) & 0xFu;
        o4[q] = (b4 & 1u) | ((b4 & 2u) << 7) | ((b4 & 4u) << 14) | ((b4 & 8u) << 21);
      }
      if (v + 16 <= cnt) {
        V16Unaligned x = {{o4[0], o4[1], o4[2], o4[3]}};
        *reinterpret_cast<V16Unaligned *>(dst + v) = x;
      } else {
        for (int j = 0; v + j < cnt; j++) dst[v + j] = (uint8_t)((bits >> j) & 1u);
      }
    }
  }
}

// ---- null scatter: out[i] = def[i] == max_def ? dense[rank(i)] : 0 ----
constexpr int kScatterTile = 4096;

__global__ __launch_bounds__(256) void k_scatter_count(const int32_t *def, int64_t n, int max_def, int64_t *partial) {
  __shared__ int s_cnt[4];
  const int64_t base = (int64_t)blockIdx.x * kScatterTile;
  int c = 0;
  for (int i = threadIdx.x * 4; i < kScatterTile; i += 1024) {
    const int64_t s = base + i;
    if (s + 4 <= n) {
      const int4 d = *reinterpret_cast<const int4 *>(def + s);
      c += (d.x == max_def) + (d.y == max_def) + (d.z == max_def) + (d.w == max_def);
    } else {
      for (int j = 0; j < 4; j++) if (s + j < n) c += def[s + j] == max_def;
    }
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (int64_t)s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// exclusive scan of the per-tile counts, one block
__global__ __launch_bounds__(1024) void k_scatter_scan(int64_t *partial, int n_tiles) {
  __shared__ int64_t s[1024];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < n_tiles; t0 += 1024) {
    const int t = t0 + threadIdx.x;
    const int64_t x = t < n_tiles ? partial[t] : 0;
    s[threadIdx.x] = x;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int64_t y = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
      __syncthreads();
      s[threadIdx.x] += y;
      __syncthreads();
    }
    if (t < n_tiles) partial[t] = carry + s[threadIdx.x] - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry += s[1023];
    __syncthreads();
  }
}

// A block takes kScatterTile slots in four rounds of 1024: every thread reads four consecutive definition levels
// with one 16-byte load (a wave covers 1 KiB of levels), the block scans the counts of valid slots, and the thread
// writes its four slots with one or two 16-byte stores, taking its values from consecutive dense positions.
template <int ESZ>
__global__ __launch_bounds__(256) void k_scatter_apply(const int32_t *def, int64_t n, int max_def, const int64_t *partial,
                                                       const uint8_t *dense, uint8_t *out, int esz) {
  __shared__ int s_wave[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * kScatterTile;
  int64_t rank0 = partial[blockIdx.x];  // dense position of the tile's first valid slot
#pragma unroll 1
  for (int it = 0; it < kScatterTile / 1024; it++) {
    const int64_t s0 = base + it * 1024 + tid * 4;
    int4 d = make_int4(-1, -1, -1, -1);
    if (s0 + 4 <= n) d = *reinterpret_cast<const int4 *>(def + s0);
    else {
      if (s0 < n) d.x = def[s0];
      if (s0 + 1 < n) d.y = def[s0 + 1];
      if (s0 + 2 < n) d.z = def[s0 + 2];
    }
    const bool v0 = s0 < n && d.x == max_def, v1 = s0 + 1 < n && d.y == max_def, v2 = s0 + 2 < n && d.z == max_def,
               v3 = s0 + 3 < n && d.w == max_def;
    const int c = (int)v0 + (int)v1 + (int)v2 + (int)v3;
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    __syncthreads();  // the previous round's totals have been read
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { if (q < wv) wbase += s_wave[q]; total += s_wave[q]; }
    int64_t r = rank0 + wbase + incl - c;
    rank0 += total;
    if (s0 >= n) continue;
    if (ESZ == 4) {
      const uint32_t *dn = reinterpret_cast<const uint32_t *>(dense);
      uint32_t y[4];
      y[0] = v0 ? dn[r] : 0u; r += v0;
      y[1] = v1 ? dn[r] : 0u; r += v1;
      y[2] = v2 ? dn[r] : 0u; r += v2;
      y[3] = v3 ? dn[r] : 0u;
      uint32_t *o = reinterpret_cast<uint32_t *>(out) + s0;
      if (s0 + 4 <= n) { V16Unaligned x = {{y[0], y[1], y[2], y[3]}}; *reinterpret_cast<V16Unaligned *>(o) = x; }
      else for (int j = 0; s0 + j < n; j++) o[j] = y[j];
    } else if (ESZ == 8) {
      const uint64_t *dn = reinterpret_cast<const uint64_t *>(dense);
      uint64_t y[4];
      y[0] = v0 ? dn[r] : 0ull; r += v0;
      y[1] = v1 ? dn[r] : 0ull; r += v1;
      y[2] = v2 ? dn[r] : 0ull; r += v2;
      y[3] = v3 ? dn[r] : 0ull;
      uint64_t *o = reinterpret_cast<uint64_t *>(out) + s0;
      if (s0 + 4 <= n) {
        V16Unaligned x0 = {{(uint32_t)y[0], (uint32_t)(y[0] >> 32), (uint32_t)y[1], (uint32_t)(y[1] >> 32)}};
        V16Unaligned x1 = {{(uint32_t)y[2], (uint32_t)(y[2] >> 32), (uint32_t)y[3], (uint32_t)(y[3] >> 32)}};
        reinterpret_cast<V16Unaligned *>(o)[0] = x0;
        reinterpret_cast<V16Unaligned *>(o)[1] = x1;
      } else for (int j = 0; s0 + j < n; j++) o[j] = y[j];
    } else {
      const bool vv[4] = {v0, v1, v2, v3};
      for (int j = 0; j < 4 && s0 + j < n; j++) {
        for (int b = 0; b < esz; b++) out[(s0 + j) * esz + b] = vv[j] ? dense[r * esz + b] : (uint8_t)0;
        r += vv[j];
      }
    }
  }
}

// ---- null scatter straight from PLAIN pages (4- and 8-byte types, every data page PLAIN) ----
// The dense intermediate of such a column would be a plain copy of the pages' value sections, so the scatter reads
// the pages themselves: the value of the slot with rank r inside data page i sits at val_off[i] + (r - dense_base[i])
// * size.  tile_page[t] = data page that holds slot t * kScatterTile.
__global__ __launch_bounds__(256) void k_scatter_tile_pages(ChunkCtx c, int64_t n, int n_tiles, int32_t *tile_page) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  // (one thread per data page checks the page's length the way the copying kernel does, binaryread.go:12-18)
  if (t < c.nd) {
    const int p = c.streams[2 * c.nd + t].page;
    PageState *ps = &c.state[p];
    if (ps->status == 0 && c.pages[p].encoding == PQD_ENC_PLAIN) {
      const int st = plain_fixed_status(c.physical_type, ps->val_len, ps->cnt, c.elem_size);
      if (st) set_status1(&ps->status, st);
    }
  }
  if (t > n_tiles) return;
  if (t == n_tiles) { tile_page[t] = c.nd - 1; return; }
  const int64_t s = (int64_t)t * kScatterTile;
  int lo = 0, hi = c.nd;   // the last data page whose first slot is at or before s
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (c.pages[c.streams[2 * c.nd + mid].page].slot_base <= s) lo = mid; else hi = mid;
  }
  tile_page[t] = lo;
}

template <int ESZ>
__global__ __launch_bounds__(256) void k_scatter_apply_plain(ChunkCtx c, const int32_t *def, int64_t n, const int64_t *partial,
                                                             const int32_t *tile_page, uint8_t *out) {
  typedef typename std::conditional<ESZ == 8, uint64_t, uint32_t>::type T;
  struct TU { T v; } __attribute__((packed));
  __shared__ int s_wave[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int max_def = c.max_def;
  const int64_t base = (int64_t)blockIdx.x * kScatterTile;
  const int pg_lo = tile_page[blockIdx.x], pg_hi = tile_page[blockIdx.x + 1];
  const Stream *vstreams = c.streams + 2 * c.nd;
  int64_t rank0 = partial[blockIdx.x];
#pragma unroll 1
  for (int it = 0; it < kScatterTile / 1024; it++) {
    const int64_t s0 = base + it * 1024 + tid * 4;
    int4 d = make_int4(-1, -1, -1, -1);
    if (s0 + 4 <= n) d = *reinterpret_cast<const int4 *>(def + s0);
    else {
      if (s0 < n) d.x = def[s0];
      if (s0 + 1 < n) d.y = def[s0 + 1];
      if (s0 + 2 < n) d.z = def[s0 + 2];
    }
    const bool vv[4] = {s0 < n && d.x == max_def, s0 + 1 < n && d.y == max_def, s0 + 2 < n && d.z == max_def,
                        s0 + 3 < n && d.w == max_def};
    const int cv = (int)vv[0] + (int)vv[1] + (int)vv[2] + (int)vv[3];
    int incl = cv;
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    __syncthreads();  // the previous round's totals have been read
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { if (q < wv) wbase += s_wave[q]; total += s_wave[q]; }
    int64_t r = rank0 + wbase + incl - cv;
    rank0 += total;
    if (s0 >= n) continue;
    // the data page of slot s0: the last one in [pg_lo, pg_hi] that starts at or before it
    int lo = pg_lo, hi = pg_hi + 1;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (c.pages[vstreams[mid].page].slot_base <= s0) lo = mid; else hi = mid;
    }
    int pg = lo;
    int p = vstreams[pg].page;
    int64_t next_slot = pg + 1 < c.nd ? c.pages[vstreams[pg + 1].page].slot_base : n;
    int64_t dense_base = c.state[p].dense_base, val_len = c.state[p].val_len;
    const uint8_t *src = c.in + c.state[p].val_off;
    T y[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      while (s0 + j >= next_slot && pg + 1 < c.nd) {   // (pages without slots are stepped over)
        pg++;
        p = vstreams[pg].page;
        next_slot = pg + 1 < c.nd ? c.pages[vstreams[pg + 1].page].slot_base : n;
        dense_base = c.state[p].dense_base; val_len = c.state[p].val_len;
        src = c.in + c.state[p].val_off;
      }
      const int64_t off = (r - dense_base) * ESZ;
      // (a page that is too short has failed; its slots only must not be read from beyond it)
      y[j] = (vv[j] && off >= 0 && off + ESZ <= val_len) ? reinterpret_cast<const TU *>(src + off)->v : (T)0;
      r += vv[j];
    }
    T *o = reinterpret_cast<T *>(out) + s0;
    if (s0 + 4 <= n) {
      if (ESZ == 4) {
        V16Unaligned x = {{(uint32_t)y[0], (uint32_t)y[1], (uint32_t)y[2], (uint32_t)y[3]}};
        *reinterpret_cast<V16Unaligned *>(o) = x;
      } else {
        V16Unaligned x0 = {{(uint32_t)y[0], (uint32_t)((uint64_t)y[0] >> 32), (uint32_t)y[1], (uint32_t)((uint64_t)y[1] >> 32)}};
        V16Unaligned x1 = {{(uint32_t)y[2], (uint32_t)((uint64_t)y[2] >> 32), (uint32_t)y[3], (uint32_t)((uint64_t)y[3] >> 32)}};
        reinterpret_cast<V16Unaligned *>(o)[0] = x0;
        reinterpret_cast<V16Unaligned *>(o)[1] = x1;
      }
    } else for (int j = 0; s0 + j < n; j++) o[j] = y[j];
  }
}

// ---- the same from validity bits: the level expansion leaves, per definition-level tile (kScatterTile slots of one page), a
// bit per slot that holds a value and their number, so the levels are not read again: counts -> exclusive scan -> every
// thread takes 16 slots (one half word of bits), ranks by popcount, values from the page ----
static_assert(kScatterTile == kTile, "scatter tiles are the level expansion's tiles");

__global__ __launch_bounds__(1024) void k_scatter_scan_cnt(const int32_t *cnt, int64_t *partial, int n_tiles) {
  // one workgroup: every thread adds up a stretch of consecutive counts, the stretches' sums are scanned (wave scans, then
  // the sixteen wave totals), and every thread writes the exclusive prefixes of its stretch
  __shared__ int64_t s_w[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (n_tiles + 1023) / 1024;
  const int lo = tid * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
  int64_t sum = 0;
  for (int t = lo; t < hi; t++) sum += cnt[t];
  int64_t incl = sum;
  for (int o = 1; o < 64; o <<= 1) {
    const int64_t y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  int64_t base = 0, total = 0;
#pragma unroll
  for (int q = 0; q < 16; q++) { if (q < wv) base += s_w[q]; total += s_w[q]; }
  int64_t run = base + incl - sum;
  for (int t = lo; t < hi; t++) { partial[t] = run; run += cnt[t]; }
  if (tid == 0) partial[n_tiles] = total;
}

// A workgroup takes a tile, a wavefront 1024 of its slots (32 words of bits): lane l holds the popcount of word l, a wave
// scan gives every word its rank, and the wavefront then walks its slots 64 at a time - lane l takes slot 64 j + l - so that
// neighbouring lanes read neighbouring values from the page and write neighbouring slots.
// What a tile's workgroup needs, worked out for all tiles at once by k_scatter_bits_plan (page of the tile, its first tile,
// the rank of the tile's first value): the scatter itself then starts with one load instead of a chain of four.
struct ScatterPlan {
  int64_t src;        // chunk offset of the value of rank 0 of the tile (the page's value section + the values in front of the tile)
  int64_t out_slot;   // slot of the tile's first value in the column image
  int32_t left;       // bytes of the value section from `src` on (may be negative when the page is too short)
  int32_t n;          // slots of the tile (0: nothing to do)
};

__global__ __launch_bounds__(256) void k_scatter_bits_plan(ChunkCtx c, const int32_t *__restrict__ tile_prefix, const int32_t *__restrict__ tile_stream,
                                                           int def_tile0, int n_def_tiles, const int64_t *__restrict__ partial, ScatterPlan *plan) {
  const int dt = blockIdx.x * 256 + threadIdx.x;
  if (dt >= n_def_tiles) return;
  const int ls = tile_stream[def_tile0 + dt];            // the definition-level stream (index among the level streams)
  const int p = c.streams[ls].page;
  const int first = tile_prefix[ls] - def_tile0;         // the page's first definition-level tile
  const int t0 = (dt - first) * kScatterTile;
  const PageState *ps = &c.state[p];
  if (dt == first && ps->status == 0 && c.pages[p].encoding == PQD_ENC_PLAIN) {
    // (the page's length is checked the way the copying kernel does, binaryread.go:12-18)
    const int st = plain_fixed_status(c.physical_type, ps->val_len, ps->cnt, c.elem_size);
    if (st) set_status1(&c.state[p].status, st);
  }
  const int n = c.pages[p].num_values > 0 ? c.pages[p].num_values : 0;
  ScatterPlan pl{0, 0, 0, 0};
  if (t0 < n) {
    const int64_t r0 = (partial[dt] - partial[first]) * c.elem_size;
    pl.src = ps->val_off + r0;
    const int64_t left = ps->val_len - r0;
    pl.left = left > 0x7fffffff ? 0x7fffffff : (left < -0x7fffffff ? -0x7fffffff : (int32_t)left);
    pl.out_slot = c.pages[p].slot_base + t0;
    pl.n = n - t0 < kScatterTile ? n - t0 : kScatterTile;
  }
  plan[dt] = pl;
}

template <int ESZ>
__global__ __launch_bounds__(256) void k_scatter_bits_plain(ChunkCtx c, const ScatterPlan *__restrict__ plan, const uint32_t *__restrict__ bits, uint8_t *out) {
  typedef typename std::conditional<ESZ == 8, uint64_t, uint32_t>::type T;
  struct TU { T v; } __attribute__((packed));
  __shared__ int s_wave[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int dt = blockIdx.x;
  const ScatterPlan pl = plan[dt];
  const uint32_t word = lane < 32 ? bits[(size_t)dt * (kScatterTile / 32) + wv * 32 + lane] : 0u;
  if (pl.n <= 0) return;
  const int cw = __builtin_popcount(word);
  int incl = cw;
  for (int o = 1; o < 32; o <<= 1) {
    const int y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  const int wave_total = __shfl(incl, 31);
  if (lane == 0) s_wave[wv] = wave_total;
  __syncthreads();
  int wbase = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) if (q < wv) wbase += s_wave[q];
  const int excl = incl - cw;                                         // values in the wavefront's words before word `lane`
  const uint8_t *src = c.in + pl.src;
  const int64_t val_left = pl.left;
  T *o = reinterpret_cast<T *>(out) + pl.out_slot + wv * 1024;
  const int left = pl.n - wv * 1024;                                  // slots of the tile from the wavefront's first on
  // a lane takes two neighbouring slots: a store instruction of the wavefront leaves 128 slots (1 KiB of 64-bit values) without gaps
  const uint32_t sh = 2u * (uint32_t)(lane & 15);
  const uint32_t below = (1u << sh) - 1u;
  struct T2 { T a, b; } __attribute__((packed));
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int wi = 4 * j + (lane >> 4);
    const uint32_t wj = (uint32_t)__shfl((int)word, wi);
    const int ej = __shfl(excl, wi);
    const bool v0 = (wj >> sh) & 1u, v1 = (wj >> (sh + 1u)) & 1u;
    const int64_t off0 = (int64_t)(wbase + ej + __builtin_popcount(wj & below)) * ESZ;
    const int64_t off1 = off0 + (v0 ? ESZ : 0);
    // (a page that is too short has failed; its slots only must not be read from beyond it)
    const T y0 = (v0 && off0 + ESZ <= val_left) ? reinterpret_cast<const TU *>(src + off0)->v : (T)0;
    const T y1 = (v1 && off1 + ESZ <= val_left) ? reinterpret_cast<const TU *>(src + off1)->v : (T)0;
    const int s0 = 128 * j + 2 * lane;
    if (s0 + 1 < left) *reinterpret_cast<T2 *>(o + s0) = T2{y0, y1};
    else if (s0 < left) reinterpret_cast<TU *>(o + s0)->v = y0;
  }
}

size_t scatter_plan_bytes(int n_def_tiles) { return sizeof(ScatterPlan) * ((size_t)n_def_tiles + 1); }

hipError_t launch_scatter_plain_bits(const ChunkCtx &c, const int32_t *tile_prefix, const int32_t *tile_stream, int def_tile0, int n_def_tiles,
                                     const uint32_t *valid_bits, const int32_t *valid_cnt, int64_t *partial, void *plan, uint8_t *out, hipStream_t stream) {
  if (n_def_tiles <= 0) return hipSuccess;
  ScatterPlan *pl = static_cast<ScatterPlan *>(plan);
  hipLaunchKernelGGL(k_scatter_scan_cnt, dim3(1), dim3(1024), 0, stream, valid_cnt, partial, n_def_tiles);
  hipLaunchKernelGGL(k_scatter_bits_plan, dim3((n_def_tiles + 255) / 256), dim3(256), 0, stream, c, tile_prefix, tile_stream, def_tile0, n_def_tiles,
                     (const int64_t *)partial, pl);
  if (c.elem_size == 4)
    hipLaunchKernelGGL(k_scatter_bits_plain<4>, dim3(n_def_tiles), dim3(256), 0, stream, c, (const ScatterPlan *)pl, valid_bits, out);
  else
    hipLaunchKernelGGL(k_scatter_bits_plain<8>, dim3(n_def_tiles), dim3(256), 0, stream, c, (const ScatterPlan *)pl, valid_bits, out);
  return hipGetLastError();
}

// ---- row index: out[r] = slot of the r-th entry with repetition level 0 (layout/table.go:73-103 cuts rows there) ----
__global__ __launch_bounds__(256) void k_row_starts(const int32_t *rep, int64_t n, const int64_t *partial, int64_t *out,
                                                    int64_t n_rows) {
  __shared__ int s_wave[4];
  const int64_t base = (int64_t)blockIdx.x * kScatterTile;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t s0 = base + (int64_t)tid * 16;
  uint32_t first = 0;
  for (int j = 0; j < 16; j++)
    if (s0 + j < n && (rep == nullptr || rep[s0 + j] == 0)) first |= 1u << j;
  const int c = __builtin_popcount(first);
  int incl = c;
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) s_wave[wv] = incl;
  __syncthreads();
  int wbase = 0;
  for (int q = 0; q < wv; q++) wbase += s_wave[q];
  int64_t rank = partial[blockIdx.x] + wbase + incl - c;
  for (int j = 0; j < 16; j++)
    if ((first >> j) & 1u) out[rank++] = s0 + j;
  if (blockIdx.x == 0 && tid == 0) out[n_rows] = n;  // end marker
}

__global__ __launch_bounds__(256) void k_iota(int64_t *out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= n; i += (int64_t)gridDim.x * 256) out[i] = i;
}

hipError_t launch_row_starts(const int32_t *rep_levels, int64_t n_slots, int64_t n_rows, int64_t *scratch, int64_t *out,
                             hipStream_t stream) {
  if (rep_levels == nullptr) {  // flat column: every slot is a row
    hipLaunchKernelGGL(k_iota, dim3(1024), dim3(256), 0, stream, out, n_slots);
    return hipGetLastError();
  }
  const int n_tiles = (int)((n_slots + kScatterTile - 1) / kScatterTile);
  if (n_tiles <= 0) {
    hipLaunchKernelGGL(k_row_starts, dim3(1), dim3(256), 0, stream, rep_levels, (int64_t)0, scratch, out, n_rows);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_scatter_count, dim3(n_tiles), dim3(256), 0, stream, rep_levels, n_slots, 0, scratch);
  hipLaunchKernelGGL(k_scatter_scan, dim3(1), dim3(1024), 0, stream, scratch, n_tiles);
  hipLaunchKernelGGL(k_row_starts, dim3(n_tiles), dim3(256), 0, stream, rep_levels, n_slots, scratch, out, n_rows);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
hipError_t launch_dict_plain_fixed(const ChunkCtx &c, const int32_t *dict_pages, int n_dict, uint8_t *dict_out, hipStream_t stream) {
  if (n_dict <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_dict_plain_fixed, dim3(n_dict > 64 ? 2 : 16, n_dict), dim3(256), 0, stream, c, dict_pages, dict_out);
  return hipGetLastError();
}

hipError_t launch_plain_fixed(const ChunkCtx &c, const int32_t *tile_prefix, int n_tiles, int64_t max_page_bytes, uint8_t *values_out,
                              hipStream_t stream) {
  if (n_tiles <= 0) return hipSuccess;
  const int64_t pieces = (max_page_bytes + kPlainTile - 1) / kPlainTile;
  // (many small pages: most workgroups of a page-by-piece grid would find nothing to do)
  if (c.nd <= 65535 && pieces >= 1 && pieces <= 0x7FFFFFFF && pieces * (int64_t)c.nd <= 2 * (int64_t)n_tiles + 1024) {
    hipLaunchKernelGGL(k_plain_fixed_pages, dim3((unsigned)pieces, c.nd), dim3(256), 0, stream, c, values_out);
    return hipGetLastError();
  }
  PlainArgs a{c, tile_prefix, n_tiles, values_out};
  int grid = n_tiles < 256 * 16 ? n_tiles : 256 * 16;
  hipLaunchKernelGGL(k_plain_fixed, dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_byte_stream_split(const ChunkCtx &c, const int32_t *tile_prefix, int n_tiles, uint8_t *values_out,
                                    hipStream_t stream) {
  if (n_tiles <= 0) return hipSuccess;
  PlainArgs a{c, tile_prefix, n_tiles, values_out};
  int grid = n_tiles < 256 * 16 ? n_tiles : 256 * 16;
  if (c.elem_size == 4) hipLaunchKernelGGL(k_byte_stream_split<4>, dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(k_byte_stream_split<8>, dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_plain_boolean(const ChunkCtx &c, const int32_t *tile_prefix, int n_tiles, uint8_t *values_out,
                                hipStream_t stream) {
  if (n_tiles <= 0) return hipSuccess;
  PlainArgs a{c, tile_prefix, n_tiles, values_out};
  int grid = n_tiles < 256 * 16 ? n_tiles : 256 * 16;
  hipLaunchKernelGGL(k_plain_boolean, dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// per-tile (4096 slots) counts of non-null slots, exclusive-scanned in place; n_tiles may exceed the slots
hipError_t launch_scatter_rank(const int32_t *def_levels, int64_t n_slots, int max_def, int64_t *partial, int n_tiles,
                               hipStream_t stream) {
  if (n_tiles <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_scatter_count, dim3(n_tiles), dim3(256), 0, stream, def_levels, n_slots, max_def, partial);
  hipLaunchKernelGGL(k_scatter_scan, dim3(1), dim3(1024), 0, stream, partial, n_tiles);
  return hipGetLastError();
}

hipError_t launch_scatter_fixed(const ChunkCtx &c, const int32_t *def_levels, const uint8_t *dense, uint8_t *out,
                                int64_t n_slots, int64_t *scratch, hipStream_t stream) {
  if (n_slots <= 0) return hipSuccess;
  const int n_tiles = (int)((n_slots + kScatterTile - 1) / kScatterTile);
  hipLaunchKernelGGL(k_scatter_count, dim3(n_tiles), dim3(256), 0, stream, def_levels, n_slots, c.max_def, scratch);
  hipLaunchKernelGGL(k_scatter_scan, dim3(1), dim3(1024), 0, stream, scratch, n_tiles);
  if (c.elem_size == 4)
    hipLaunchKernelGGL(k_scatter_apply<4>, dim3(n_tiles), dim3(256), 0, stream, def_levels, n_slots, c.max_def, scratch, dense, out, 4);
  else if (c.elem_size == 8)
    hipLaunchKernelGGL(k_scatter_apply<8>, dim3(n_tiles), dim3(256), 0, stream, def_levels, n_slots, c.max_def, scratch, dense, out, 8);
  else
    hipLaunchKernelGGL(k_scatter_apply<1>, dim3(n_tiles), dim3(256), 0, stream, def_levels, n_slots, c.max_def, scratch, dense, out, c.elem_size);
  return hipGetLastError();
}

// scratch: n_tiles + 1 ranks (int64) followed by n_tiles + 1 page indices (int32)
hipError_t launch_scatter_plain(const ChunkCtx &c, const int32_t *def_levels, uint8_t *out, int64_t n_slots, int64_t *scratch,
                                hipStream_t stream) {
  if (n_slots <= 0 || c.nd <= 0) return hipSuccess;
  const int n_tiles = (int)((n_slots + kScatterTile - 1) / kScatterTile);
  int32_t *tile_page = reinterpret_cast<int32_t *>(scratch + n_tiles + 1);
  const int n_thr = (n_tiles + 1 > c.nd ? n_tiles + 1 : c.nd);
  hipLaunchKernelGGL(k_scatter_tile_pages, dim3((n_thr + 255) / 256), dim3(256), 0, stream, c, n_slots, n_tiles, tile_page);
  hipLaunchKernelGGL(k_scatter_count, dim3(n_tiles), dim3(256), 0, stream, def_levels, n_slots, c.max_def, scratch);
  hipLaunchKernelGGL(k_scatter_scan, dim3(1), dim3(1024), 0, stream, scratch, n_tiles);
  if (c.elem_size == 4)
    hipLaunchKernelGGL(k_scatter_apply_plain<4>, dim3(n_tiles), dim3(256), 0, stream, c, def_levels, n_slots, scratch, tile_page, out);
  else
    hipLaunchKernelGGL(k_scatter_apply_plain<8>, dim3(n_tiles), dim3(256), 0, stream, c, def_levels, n_slots, scratch, tile_page, out);
  return hipGetLastError();
}

}  // namespace pqd
