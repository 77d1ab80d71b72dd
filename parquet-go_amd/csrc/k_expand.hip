// Tile expansion of RLE / bit-packed hybrid streams on gfx950: run table -> values.
// Levels become int32 arrays (+ null and row counts); dictionary indices are gathered straight into the
// value array.
//
// Reference behaviour reproduced here: encoding/encodingread.go:146-167 (RLE run), :170-228 (bit-packed run),
// layout/page.go:139-164 (dictionary gather), :835-888 (levels, null count), :925-927 (row count).
#include <hip/hip_runtime.h>
#include "pqd_kernels.h"
#include "pqd_device.h"

namespace pqd {

__device__ __forceinline__ void set_status(int32_t *p, int code) { atomicCAS(p, 0, code); }  // first error wins

constexpr int kSegBytes = 128;  // run index of the first item of every 64-item segment of a tile (32 entries)

struct ExpandArgs {
  ChunkCtx c;
  const int32_t *tile_prefix;  // n_streams + 1 entries
  const int32_t *tile_stream;  // n_tiles entries: stream (relative to stream0) of each tile
  int stream0;
  int tile_lo;                 // this launch expands the tiles [tile_lo, n_tiles)
  int n_tiles;
  int32_t *rep_out, *def_out;  // levels
  const uint8_t *dict;         // dictionary, elem_size bytes per entry, 16-byte aligned and padded
  int32_t dict_n;
  int32_t no_dict;             // the pages of this launch have no dictionary: their indices pass through unchecked
  int32_t dict_lds_n;          // entries [0, dict_lds_n) are staged in LDS
  int32_t dict_lds_bytes;      // LDS reserved for them (multiple of 16)
  int32_t runs_cap;            // most run-slice entries a tile may keep in LDS (+ 3)
  int32_t mark_cap;            // items per tile the mark array holds
  int32_t pool_bytes;          // LDS a tile's run slice and packed bytes share (multiple of 16): a tile of many runs has
                               // few packed bytes (an RLE run carries none) and a tile of many packed bytes few runs
  int32_t group_bytes;         // LDS of one group: segments | marks | pool (run slice, then packed bytes)
  uint8_t *values_out;
  TileDesc *descs;             // n_tiles entries, written by k_tile_desc
  const int32_t *work_flag;    // not null: the launch has work only if this word is set (every stream may have been expanded already)
  // levels: which slots of every definition-level tile hold a value (def == max_def), one bit per slot (128 words per tile,
  // tile = index among the definition-level tiles), and how many; nullptr: not wanted
  uint32_t *valid_bits;
  int32_t *valid_cnt;          // zero when the launch starts
  int32_t def_tile0;           // index of the first definition-level tile among the tiles of the launch
};

constexpr int kBitsBytes = kTile / 8;   // LDS of a tile's validity bits (levels only)

__host__ __device__ inline int runs_lds_bytes(int runs_cap) { return (runs_cap + 3) * (int)sizeof(RunEntry); }
__host__ __device__ inline int marks_lds_bytes(int mark_cap) { return (mark_cap * 2 + 15) / 16 * 16; }

// w bits at bit position `bit` (relative to `payload`) of the stream; bytes at or beyond `avail` are zero.
__device__ __forceinline__ uint32_t extract_bits(const uint8_t *sbase, uint32_t avail, uint32_t payload, uint64_t bit,
                                                 int w) {
  uint64_t byte = (uint64_t)payload + (bit >> 3);
  uint32_t sh = (uint32_t)bit & 7u;
  uint64_t x;
  if (byte + 8 <= (uint64_t)avail) {
    x = reinterpret_cast<const U64Unaligned *>(sbase + byte)->v;
  } else {
    x = 0;
    for (int k = 0; k < 8; k++)
      if (byte + k < (uint64_t)avail) x |= (uint64_t)sbase[byte + k] << (8 * k);
  }
  uint32_t mask = w >= 32 ? 0xFFFFFFFFu : ((1u << w) - 1u);
  return (uint32_t)(x >> sh) & mask;
}

// Values v .. v+n4-1 of a stream straight from global memory (any number of runs): binary search for the run
// holding v, then a walk.
__device__ __forceinline__ void decode4(const RunEntry *R, int nr, int v, int n4, const uint8_t *sbase, uint32_t avail,
                                        int w, uint32_t out[4]) {
  int lo = 0, hi = nr - 1;  // largest i in [0, nr-2] with start(i) <= v
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if ((int)(R[mid].x & 0x7FFFFFFFu) <= v) lo = mid; else hi = mid;
  }
  int i = lo;
  uint32_t ex = R[i].x, ey = R[i].y;
  int start = (int)(ex & 0x7FFFFFFFu);
  int next = (int)(R[i + 1].x & 0x7FFFFFFFu);
  for (int j = 0; j < n4; j++) {
    const int vv = v + j;
    while (vv >= next) {
      i++;
      ex = R[i].x; ey = R[i].y;
      start = (int)(ex & 0x7FFFFFFFu);
      next = (int)(R[i + 1].x & 0x7FFFFFFFu);
    }
    if ((ex >> 31) == 0) out[j] = ey;
    else out[j] = extract_bits(sbase, avail, ey, (uint64_t)(uint32_t)(vv - start) * (uint32_t)w, w);
  }
}

// ---------------------------------------------------------------------------------------------
// k_tile_desc: one thread per tile, after the prescan: which runs, items and stream bytes the tile covers.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_desc(ExpandArgs a) {
  if (a.work_flag && *a.work_flag == 0) return;
  const int tile = a.tile_lo + blockIdx.x * 256 + threadIdx.x;
  if (tile >= a.n_tiles) return;
  const ChunkCtx &c = a.c;
  TileDesc d;
  d.flags = 0;
  const int ls = a.tile_stream[tile];
  const int k = tile - a.tile_prefix[ls];
  const Stream *sp = &c.streams[a.stream0 + ls];
  const int cnt = sp->cnt;
  d.t0 = k * kTile;
  d.page = sp->page;
  d.pad = 0;
  if (sp->status != 0 || sp->direct || d.t0 >= cnt) {
    d.in_off = 0; d.out_elem = 0; d.run0 = 0; d.item_lo = 0; d.n_items = 0; d.t1 = d.t0; d.org = 0; d.span = 0; d.nr = 0;
    d.avail = 0; d.w = 0;
    a.descs[tile] = d;
    return;
  }
  d.t1 = cnt < d.t0 + kTile ? cnt : d.t0 + kTile;
  const int2 *tile_run = reinterpret_cast<const int2 *>(c.tile_run) + sp->tile_base;
  const int2 tk = tile_run[k];
  const bool last_tile = !(d.t0 + kTile < cnt);
  const int2 tk1 = last_tile ? make_int2(sp->n_runs - 1, sp->pad) : tile_run[k + 1];
  const int r0 = tk.x, r1 = tk1.x;
  d.nr = r1 - r0 + 2;
  d.run0 = sp->run_base + r0;
  const RunEntry *gruns = c.runs + d.run0;
  d.in_off = sp->off;
  d.avail = sp->avail;
  d.w = sp->w;
  d.out_elem = sp->out_base;
  const uint32_t w = (uint32_t)d.w;
  // items of the tile: from the item holding value t0 to the one holding value t1 - 1
  const RunEntry e0 = gruns[0];
  const int start0 = (int)(e0.x & 0x7FFFFFFFu);
  const uint32_t k0 = (uint32_t)(d.t0 - start0) >> 3;
  d.item_lo = e0.z + k0;
  RunEntry eL = gruns[d.nr - 2];
  if ((int)(eL.x & 0x7FFFFFFFu) >= d.t1) eL = gruns[d.nr - 3];  // the next tile starts with a run of its own
  const uint32_t item_hi = eL.z + ((uint32_t)(d.t1 - 1 - (int)(eL.x & 0x7FFFFFFFu)) >> 3) + 1u;
  d.n_items = (int)(item_hi - d.item_lo);
  // stream bytes the tile can touch: from its first group (or the header of its first run) to the position of
  // the next tile's first value plus one group; org is 16-byte aligned in memory
  const int b_lo = (e0.x >> 31) ? (int)(e0.y + k0 * w) : tk.y;
  const int b_hi = tk1.y + (int)w + 1;
  d.org = b_lo - (int)((reinterpret_cast<uintptr_t>(c.in + d.in_off) + (uintptr_t)(uint32_t)b_lo) & 15u);
  // (a tile deep inside a bit-packed run that the stream holds only the beginning of starts behind the stream's last
  //  byte: it touches no bytes at all, and every value of it is the zero fill)
  d.span = b_hi > d.org ? b_hi - d.org : 0;
  const bool in_lds = d.nr <= a.runs_cap + 2 && (d.nr + 1) * (int)sizeof(RunEntry) + ((d.span + 8 + 15) & ~15) <= a.pool_bytes && d.n_items <= a.mark_cap;
  // bit 3: the values are the unpacked numbers themselves (Encoding_RLE data pages, layout/page.go:657-667)
  // (also set for dictionary-encoded pages that have no dictionary page in front of them: the call fails for them
  //  with its own code, and the stream's own errors, which come first in the reference, are still found)
  const bool ident = a.stream0 >= 2 * c.nd && (c.pages[sp->page].encoding == PQD_ENC_RLE || a.no_dict);
  d.flags = 1 | (in_lds ? 2 : 0) | ((a.stream0 + ls) >= c.nd ? 4 : 0) | (ident ? 8 : 0);
  a.descs[tile] = d;
}

// inclusive running maximum over the 64 lanes (DPP row shifts, then the two row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x) {
  uint32_t y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false); x = x > y ? x : y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false); x = x > y ? x : y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false); x = x > y ? x : y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false); x = x > y ? x : y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false); x = x > y ? x : y;  // row_bcast:15
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false); x = x > y ? x : y;  // row_bcast:31
  return x;
}

// DMODE 0: dictionary in global memory; 1: whole dictionary in LDS; 2: entries below dict_lds_n in LDS, the
// rest in global memory.  (Separate code paths: a run-time choice of address space would turn every lookup
// into a flat load.)
template <int DMODE, class T>
__device__ __forceinline__ T dict_lookup(const ExpandArgs &a, const uint8_t *s_dict, uint32_t idx) {
  if (DMODE == 0) return reinterpret_cast<const T *>(a.dict)[idx];
  if (DMODE == 1) return reinterpret_cast<const T *>(s_dict)[idx];
  typedef const __attribute__((address_space(3))) T *LdsPtr;
  typedef const __attribute__((address_space(1))) T *GlobalPtr;
  const bool in = idx < (uint32_t)a.dict_lds_n;
  T v = ((LdsPtr)s_dict)[in ? idx : 0u];
  if (!in) v = ((GlobalPtr)a.dict)[idx];
  return v;
}

// The eight lookups of an item.  With part of the dictionary in LDS (DMODE 2) every lane reads LDS for all eight
// (index 0 where the entry is not there) and only the lanes beyond the LDS part load from global memory, into
// registers of their own: the global loads of an item are all in flight before the first result is needed, and
// none of them waits for an LDS read.  (One access through a pointer picked per lane would be a flat load, which
// ties up both memory paths.)
template <int DMODE, class T>
__device__ __forceinline__ void dict_lookup8(const ExpandArgs &a, const uint8_t *s_dict, const uint32_t x[8], T y[8]) {
  if (DMODE != 2) {
#pragma unroll
    for (int j = 0; j < 8; j++) y[j] = dict_lookup<DMODE, T>(a, s_dict, x[j]);
    return;
  }
  typedef const __attribute__((address_space(3))) T *LdsPtr;
  typedef const __attribute__((address_space(1))) T *GlobalPtr;
  T yg[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    yg[j] = 0;
    if (x[j] >= (uint32_t)a.dict_lds_n) yg[j] = ((GlobalPtr)a.dict)[x[j]];
  }
  T yl[8];
#pragma unroll
  for (int j = 0; j < 8; j++) yl[j] = ((LdsPtr)s_dict)[x[j] < (uint32_t)a.dict_lds_n ? x[j] : 0u];
#pragma unroll
  for (int j = 0; j < 8; j++) y[j] = x[j] < (uint32_t)a.dict_lds_n ? yl[j] : yg[j];
}

// The items of one staged tile.  GT threads; a wave takes 64 consecutive items per step.  The run holding an
// item: the running maximum, over the items before it, of the marks the runs left at their first item.
template <int MODE, int DMODE, int GT, int W>
__device__ __forceinline__ void expand_items(const ExpandArgs &a, const TileDesc &d, const RunEntry *s_runs,
                                             const int *s_seg, uint16_t *s_mark, const uint32_t *s_stage,
                                             const uint8_t *s_dict, int gtid, uint32_t staged, int &n_flag,
                                             bool &bad_index, uint32_t *s_bits) {
  const ChunkCtx &c = a.c;
  const int lane = gtid & 63;
  const int tn = d.t1 - d.t0;
  const bool is_def = (d.flags & 4) != 0;
  const bool ident = MODE != 0 && (d.flags & 8) != 0;
  const bool check_bp = MODE != 0 && !(W < 32 && (1u << (W & 31)) <= (uint32_t)a.dict_n);  // can a packed index be out of range
  for (int q0 = gtid - lane; q0 < d.n_items; q0 += GT) {
    const int q = q0 + lane;
    const bool active = q < d.n_items;
    uint32_t r = (uint32_t)s_seg[q0 >> 6];
    if (active) {
      const uint32_t m = s_mark[q];
      s_mark[q] = 0;  // left clean for the next tile
      r = r > m ? r : m;
    }
    r = wave_incl_max(r);
    uint32_t out4[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int j0 = 0, j1 = 0;
    int64_t oidx = 0;
    if (active) {
      const RunEntry e = s_runs[r];
      const int next_rel = (int)(s_runs[r + 1].x & 0x7FFFFFFFu) - d.t0;
      const uint32_t kk = d.item_lo + (uint32_t)q - e.z;
      const int vrel = (int)(e.x & 0x7FFFFFFFu) - d.t0 + (int)(kk << 3);  // tile-relative position of the item's value 0
      j0 = vrel < 0 ? -vrel : 0;
      const int lim = next_rel < tn ? next_rel : tn;
      j1 = lim - vrel < 8 ? lim - vrel : 8;
      uint32_t x[8];
      const bool bp = e.x >> 31;
      if (bp) {
        const uint32_t rel = e.y - (uint32_t)d.org + kk * (uint32_t)W;
        if (rel + (uint32_t)W + 8u <= staged) {
          unpack8<W>(s_stage, rel, x);
        } else {
          // a payload cut short by the end of the stream: bytes that are not there read as zero
          for (int j = 0; j < 8; j++)
            x[j] = extract_bits(c.in + d.in_off, (uint32_t)d.avail, e.y, (uint64_t)(kk * 8u + (uint32_t)j) * (uint32_t)W, W);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = e.y;
      }
      oidx = d.out_elem + d.t0 + vrel;

      if (MODE == 0) {
        const uint32_t cmp = is_def ? (uint32_t)c.max_def : 0u;
        // (one bit per value: which of the item's values are inside the run and the tile, which equal the level that is counted)
        const uint32_t inmask = ((1u << (j1 < 0 ? 0 : j1)) - 1u) & ~((1u << j0) - 1u);
        uint32_t eq = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          eq |= x[j] == cmp ? 1u << j : 0u;
          out4[j] = x[j];
        }
        eq &= inmask;
        n_flag += __builtin_popcount(is_def ? inmask ^ eq : eq);
        if (s_bits && is_def && j1 > j0) {
          // the item's slots that hold a value, at their places in the tile's bitmap
          const uint32_t b0 = (uint32_t)(vrel + j0), mm = eq >> j0, sh = b0 & 31u;
          if (mm) {
            atomicOr(&s_bits[b0 >> 5], mm << sh);
            if (sh > 24u && (mm >> (32u - sh)) != 0u) atomicOr(&s_bits[(b0 >> 5) + 1u], mm >> (32u - sh));
          }
        }
      } else if (MODE == 4 || MODE == 8) {
        if (ident) {
          // int32(value) / int64(value): the unpacked numbers are zero extended (encodingread.go:160, :202)
          if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 8; j++) out4[j] = x[j];
          } else {
            uint64_t y[8];
#pragma unroll
            for (int j = 0; j < 8; j++) y[j] = x[j];
            store_item(reinterpret_cast<uint64_t *>(a.values_out) + oidx, y, j0, j1);
          }
        } else {
          if (bp ? check_bp : true) {
#pragma unroll
            for (int j = 0; j < 8; j++)
              if (x[j] >= (uint32_t)a.dict_n) { bad_index = bad_index || (j >= j0 && j < j1); x[j] = 0; }
          }
          if (MODE == 4) {
            dict_lookup8<DMODE, uint32_t>(a, s_dict, x, out4);
          } else {
            uint64_t y[8];
            dict_lookup8<DMODE, uint64_t>(a, s_dict, x, y);
            store_item(reinterpret_cast<uint64_t *>(a.values_out) + oidx, y, j0, j1);
          }
        }
      } else {
        const int esz = c.elem_size;
        if (ident) {
          for (int j = j0; j < j1; j++) {
            uint8_t *dst = a.values_out + (size_t)(oidx + j) * esz;
            for (int bb = 0; bb < esz; bb++) dst[bb] = bb < 4 ? (uint8_t)(x[j] >> (8 * bb)) : (uint8_t)0;
          }
        } else {
          for (int j = j0; j < j1; j++) {
            uint32_t idx = x[j];
            if (idx >= (uint32_t)a.dict_n) { bad_index = true; idx = 0; }
            uint8_t *dst = a.values_out + (size_t)(oidx + j) * esz;
            const bool in_lds = DMODE == 1 || (DMODE == 2 && idx < (uint32_t)a.dict_lds_n);
            if (in_lds) { const uint8_t *src = s_dict + (size_t)idx * esz; for (int bb = 0; bb < esz; bb++) dst[bb] = src[bb]; }
            else { const uint8_t *src = a.dict + (size_t)idx * esz; for (int bb = 0; bb < esz; bb++) dst[bb] = src[bb]; }
          }
        }
      }
    }
    if ((MODE == 0 || MODE == 4) && active) {
      uint32_t *arr = MODE == 0 ? reinterpret_cast<uint32_t *>(is_def ? a.def_out : a.rep_out) : reinterpret_cast<uint32_t *>(a.values_out);
      store_item(arr + oidx, out4, j0, j1);
    }
  }
}

// The work on one tile once its run slice and packed bytes are in LDS (or, for a tile beyond the LDS budget, straight
// from global memory): the items, then the tile's counts or its error.
template <int MODE, int DMODE, int GT>
__device__ __forceinline__ void process_tile(const ExpandArgs &a, const TileDesc &d, const RunEntry *s_runs, const int *s_seg,
                                             uint16_t *s_mark, const uint32_t *s_stage, const uint8_t *s_dict, int gtid, uint32_t *s_bits,
                                             int tile) {
  const ChunkCtx &c = a.c;
  const int esz = c.elem_size;
  const bool valid = (d.flags & 1) != 0;
  const bool in_lds = (d.flags & 2) != 0;
  const RunEntry *gruns = c.runs + d.run0;
  const uint8_t *sbase = c.in + d.in_off;
  const uint32_t staged = (uint32_t)((d.span + 8 + 15) >> 4) * 16u;
  int n_flag = 0;  // levels: nulls or rows counted by this thread
  bool bad_index = false;
  if (in_lds) {
    switch (__builtin_amdgcn_readfirstlane(d.w)) {
#define PQD_CASE(W) case W: expand_items<MODE, DMODE, GT, W>(a, d, s_runs, s_seg, s_mark, s_stage, s_dict, gtid, staged, n_flag, bad_index, s_bits); break;
      PQD_CASE(0) PQD_CASE(1) PQD_CASE(2) PQD_CASE(3) PQD_CASE(4) PQD_CASE(5) PQD_CASE(6) PQD_CASE(7) PQD_CASE(8)
      PQD_CASE(9) PQD_CASE(10) PQD_CASE(11) PQD_CASE(12) PQD_CASE(13) PQD_CASE(14) PQD_CASE(15) PQD_CASE(16)
      PQD_CASE(17) PQD_CASE(18) PQD_CASE(19) PQD_CASE(20) PQD_CASE(21) PQD_CASE(22) PQD_CASE(23) PQD_CASE(24)
      PQD_CASE(25) PQD_CASE(26) PQD_CASE(27) PQD_CASE(28) PQD_CASE(29) PQD_CASE(30) PQD_CASE(31) PQD_CASE(32)
#undef PQD_CASE
      default: break;
    }
  } else if (valid) {
    // tiles with more runs or bytes than the LDS budget: four values per lane straight from global memory
    for (int it = 0; it < kTile / (GT * 4); it++) {
      const int rel = it * GT * 4 + gtid * 4;
      const int v = d.t0 + rel;
      if (v >= d.t1) continue;
      const int n4 = d.t1 - v < 4 ? d.t1 - v : 4;
      uint32_t x[4] = {0, 0, 0, 0};
      decode4(gruns, d.nr, v, n4, sbase, (uint32_t)d.avail, d.w, x);
      if (MODE == 0) {
        const bool is_def = (d.flags & 4) != 0;
        int32_t *outp = (is_def ? a.def_out : a.rep_out) + d.out_elem + v;
        const uint32_t cmp = is_def ? (uint32_t)c.max_def : 0u;
        for (int j = 0; j < n4; j++) {
          n_flag += is_def ? (x[j] != cmp) : (x[j] == cmp);
          outp[j] = (int32_t)x[j];
          if (s_bits && is_def && x[j] == cmp) atomicOr(&s_bits[(uint32_t)(rel + j) >> 5], 1u << ((uint32_t)(rel + j) & 31u));
        }
      } else {
        for (int j = 0; j < n4; j++) {
          uint32_t idx = x[j];
          uint8_t *dst = a.values_out + (size_t)(d.out_elem + v + j) * esz;
          if (d.flags & 8) {
            for (int bb = 0; bb < esz; bb++) dst[bb] = bb < 4 ? (uint8_t)(idx >> (8 * bb)) : (uint8_t)0;
            continue;
          }
          if (idx >= (uint32_t)a.dict_n) { bad_index = true; idx = 0; }
          const uint8_t *src = a.dict + (size_t)idx * esz;
          if (MODE == 4) *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(src);
          else if (MODE == 8) *reinterpret_cast<uint64_t *>(dst) = *reinterpret_cast<const uint64_t *>(src);
          else for (int bb = 0; bb < esz; bb++) dst[bb] = src[bb];
        }
      }
    }
  }

  if (MODE == 0) {
    // wave reduction, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) n_flag += __shfl_down(n_flag, o);
    if ((gtid & 63) == 0 && n_flag) atomicAdd((d.flags & 4) ? &c.state[d.page].n_nulls : &c.state[d.page].n_rows, n_flag);
    if (s_bits) {
      // the tile's validity bits leave for memory (a definition-level tile) and the bitmap is left clean for the next tile
      __syncthreads();
      int n = 0;
      for (int i = gtid; i < kTile / 32; i += GT) {
        const uint32_t b = s_bits[i];
        s_bits[i] = 0u;
        if (tile >= a.def_tile0) {     // (a tile that holds nothing - a failed stream - leaves zeros)
          a.valid_bits[(size_t)(tile - a.def_tile0) * (kTile / 32) + i] = b;
          n += __builtin_popcount(b);
        }
      }
      if (gtid < kTile / 32 && tile >= a.def_tile0) {     // (whole wavefronts: kTile / 32 is a multiple of 64)
        for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
        if ((gtid & 63) == 0 && n) atomicAdd(&a.valid_cnt[tile - a.def_tile0], n);
      }
    }
  } else {
    if (bad_index) set_status(&c.state[d.page].status, PQD_ERR_MALFORMED);  // dict.Values[index] out of range
  }
}

// What a thread asks for first when a tile is staged: its first run entry (and the item count behind it) and its first
// two vectors of packed bytes - all there is, for most tiles.  Asked for together, before anything waits: one trip to
// memory per tile, not one per kind of thing.
struct TileLoads {
  RunEntry e0, e1;      // run entries gtid and gtid + GT of the slice (level tiles of short runs hold more than GT)
  uint32_t zn0, zn1;
  uint4 v0, v1;
};

__device__ __forceinline__ uint4 stage_raw(const TileDesc &d, const uint8_t *sbase, int nvec, int i) {
  uint4 x = make_uint4(0, 0, 0, 0);
  if (i < nvec && d.org + 16 * i < d.avail) x = *reinterpret_cast<const uint4 *>(sbase + d.org + 16 * i);
  return x;
}

template <int GT>
__device__ __forceinline__ void tile_issue_loads(const ExpandArgs &a, const TileDesc &d, int gtid, TileLoads &pf) {
  pf.e0 = pf.e1 = RunEntry{0u, 0u, 0u, 0u};
  pf.zn0 = pf.zn1 = 0;
  pf.v0 = pf.v1 = make_uint4(0, 0, 0, 0);
  if ((d.flags & 2) == 0) return;
  const RunEntry *gruns = a.c.runs + d.run0;
  const uint8_t *sbase = a.c.in + d.in_off;
  const int nvec = (d.span + 8 + 15) >> 4;
  if (gtid < d.nr) pf.e0 = gruns[gtid];
  if (gtid < d.nr - 1) pf.zn0 = gruns[gtid + 1].z;
  if (gtid + GT < d.nr) pf.e1 = gruns[gtid + GT];
  if (gtid + GT < d.nr - 1) pf.zn1 = gruns[gtid + GT + 1].z;
  pf.v0 = stage_raw(d, sbase, nvec, gtid);
  pf.v1 = stage_raw(d, sbase, nvec, gtid + GT);
}

// Run slice, segment table, marks and packed bytes of a tile into LDS: what tile_issue_loads asked for, then whatever a
// tile of many runs or many bytes holds beyond that.
template <int GT>
__device__ __forceinline__ void tile_stage(const ExpandArgs &a, const TileDesc &d, int gtid, const TileLoads &pf, RunEntry *s_runs,
                                           int *s_seg, uint16_t *s_mark) {
  if ((d.flags & 2) == 0) return;
  const RunEntry *gruns = a.c.runs + d.run0;
  const uint8_t *sbase = a.c.in + d.in_off;
  const int nvec = (d.span + 8 + 15) >> 4;
  // the packed bytes follow the tile's run slice (nr entries and the sentinel) in the pool
  uint32_t *s_stage = reinterpret_cast<uint32_t *>(s_runs + d.nr + 1);
  const uint32_t item_lo = d.item_lo, item_hi = d.item_lo + (uint32_t)d.n_items;
  auto note_run = [&](int i, const RunEntry &e, uint32_t zn) {
    // the 64-item segments whose first item run i holds
    const int ia = e.z <= item_lo ? 0 : (int)(e.z - item_lo);
    const int ib = zn >= item_hi ? d.n_items : (zn <= item_lo ? 0 : (int)(zn - item_lo));
    for (int j = (ia + 63) >> 6; (j << 6) < ib; j++) s_seg[j] = i;
    // a mark at the run's first item (the previous tile's readers cleared theirs before the barrier in front of this)
    if (i >= 1 && e.z > item_lo && e.z < item_hi) s_mark[e.z - item_lo] = (uint16_t)i;
  };
  auto stage_put = [&](int i, uint4 x) {
    if (i >= nvec) return;
    const int left = d.avail - (d.org + 16 * i);   // bytes of the stream in this vector; the rest reads as zero
    if (left < 16) {
      uint32_t wv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int lb = left - 4 * q;
        if (lb <= 0) wv[q] = 0;
        else if (lb < 4) wv[q] &= (1u << (8 * lb)) - 1u;
      }
      x = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
    reinterpret_cast<uint4 *>(s_stage)[i] = x;
  };
  if (gtid < d.nr) {
    s_runs[gtid] = pf.e0;
    if (gtid < d.nr - 1) note_run(gtid, pf.e0, pf.zn0);
  }
  if (gtid + GT < d.nr) {
    s_runs[gtid + GT] = pf.e1;
    if (gtid + GT < d.nr - 1) note_run(gtid + GT, pf.e1, pf.zn1);
  }
  for (int i = gtid + 2 * GT; i < d.nr; i += GT) {
    const RunEntry e = gruns[i];
    s_runs[i] = e;
    if (i < d.nr - 1) note_run(i, e, gruns[i + 1].z);
  }
  if (gtid == 0) s_runs[d.nr] = RunEntry{0x7FFFFFFFu, 0u, 0xFFFFFFFFu, 0u};
  stage_put(gtid, pf.v0);
  stage_put(gtid + GT, pf.v1);
  for (int i = gtid + 2 * GT; i < nvec; i += GT) stage_put(i, stage_raw(d, sbase, nvec, i));
}

// Descriptors reach the expanding threads through a small ring in LDS, filled by plain loads two tiles ahead (a scalar
// load would have to be waited for on the spot: the kernel keeps more scalars than there are registers for).
constexpr int kRingSlots = 4;
constexpr int kDescDwords = (int)(sizeof(TileDesc) / 4);
constexpr int kDescUsed = 17;   // dwords of a descriptor that carry something
static_assert(sizeof(TileDesc) == 128, "descriptor ring slots are 128 bytes");

__device__ __forceinline__ TileDesc ring_read(const uint32_t *slot) {
  uint32_t w[kDescUsed];
#pragma unroll
  for (int k = 0; k < kDescUsed; k++) w[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot[k]);
  TileDesc d;
  d.in_off = (int64_t)(((uint64_t)w[1] << 32) | w[0]);
  d.out_elem = (int64_t)(((uint64_t)w[3] << 32) | w[2]);
  d.run0 = (int64_t)(((uint64_t)w[5] << 32) | w[4]);
  d.item_lo = w[6]; d.n_items = (int32_t)w[7]; d.t0 = (int32_t)w[8]; d.t1 = (int32_t)w[9]; d.org = (int32_t)w[10];
  d.span = (int32_t)w[11]; d.nr = (int32_t)w[12]; d.avail = (int32_t)w[13]; d.w = (int32_t)w[14]; d.flags = (int32_t)w[15];
  d.page = (int32_t)w[16]; d.pad = 0;
  return d;
}

// MODE 0: levels -> int32 (+ counts); MODE 4 / 8: dictionary gather of 4- / 8-byte elements;
// MODE 1: dictionary gather of elem_size-byte elements (BOOLEAN, INT96, FIXED_LEN_BYTE_ARRAY)
// G groups of GT threads per block; a group expands one tile at a time.
//
// Per tile (kTile consecutive values of one stream): the run slice and the packed bytes the tile needs are
// staged in LDS with coalesced 128-bit loads (every input byte is read from HBM once).  The tile's work is then
// cut into items - up to eight consecutive values of one run, a whole bit-packed group where the run is bit
// packed - and each lane takes one item at a time: either a fill (RLE) or the unpack of one byte-aligned group
// with compile-time bit positions, then the dictionary lookups and two 16-byte stores.
template <int MODE, int DMODE, int G, int GT>
__global__ __launch_bounds__(G * GT) void k_hybrid_expand(ExpandArgs a, const TileDesc *__restrict__ descs) {
  extern __shared__ uint4 smem[];
  if (a.work_flag && *a.work_flag == 0) return;   // (uniform)
  uint8_t *lds = reinterpret_cast<uint8_t *>(smem);
  const int g = G == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / GT);
  const int gtid = G == 1 ? (int)threadIdx.x : (int)threadIdx.x % GT;
  uint8_t *area = lds + a.dict_lds_bytes + g * a.group_bytes;
  int *s_seg = reinterpret_cast<int *>(area);
  uint16_t *s_mark = reinterpret_cast<uint16_t *>(area + kSegBytes);
  RunEntry *s_runs = reinterpret_cast<RunEntry *>(area + kSegBytes + marks_lds_bytes(a.mark_cap));
  const uint8_t *s_dict = lds;
  const ChunkCtx &c = a.c;
  const int esz = c.elem_size;

  if (DMODE != 0) {
    const int nvec = (a.dict_lds_n * esz + 15) / 16;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.dict);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (int i = threadIdx.x; i < nvec; i += G * GT) dst[i] = src[i];
  }
  for (int i = gtid; i < marks_lds_bytes(a.mark_cap) / 16; i += GT) reinterpret_cast<uint4 *>(s_mark)[i] = make_uint4(0, 0, 0, 0);

  // The tiles of this group: first, first + step, ...  Tile i's descriptor sits in ring slot i % kRingSlots.
  uint32_t *s_ring = reinterpret_cast<uint32_t *>(area + a.group_bytes - kRingSlots * (int)sizeof(TileDesc));
  // (levels with validity bits wanted: the bitmap sits behind the ring, beyond group_bytes)
  uint32_t *s_bits = MODE == 0 && a.valid_bits ? reinterpret_cast<uint32_t *>(area + a.group_bytes) : nullptr;
  if (s_bits) for (int i = gtid; i < kTile / 32; i += GT) s_bits[i] = 0u;
  const int first = a.tile_lo + (int)blockIdx.x * G + g, step = (int)gridDim.x * G;
  const int n_my = first < a.n_tiles ? (a.n_tiles - first + step - 1) / step : 0;
  // (the groups of a workgroup share its barriers: all of them make the rounds of the one with the most tiles)
  const int first0 = a.tile_lo + (int)blockIdx.x * G;
  const int n_rounds = first0 < a.n_tiles ? (a.n_tiles - first0 + step - 1) / step : 0;
  auto desc_words = [&](int i) { return reinterpret_cast<const uint32_t *>(descs + (first + (int64_t)i * step)); };
  if (gtid < kDescUsed && n_my > 0) s_ring[gtid] = desc_words(0)[gtid];
  __syncthreads();   // (dictionary and marks are in place too)

  // Software pipeline, one tile deep: the run entries and packed bytes of tile i + 1 are asked for before the values of
  // tile i are worked out and stored, and the descriptor of tile i + 2 with them; both are looked at only at the top
  // of the next round.  (Loads and stores of a wavefront complete in order: whatever a wavefront waits for, it also
  // waits for the stores it issued before.  Asking after the stores would put a full trip to memory behind that wait.)
  TileLoads pf;
  tile_issue_loads<GT>(a, n_my > 0 ? ring_read(s_ring) : TileDesc{}, gtid, pf);
  uint32_t dn = 0;
  if (gtid < kDescUsed && 1 < n_my) dn = desc_words(1)[gtid];
  for (int i = 0; i < n_rounds; i++) {
    TileDesc d = {};
    if (i < n_my) d = ring_read(s_ring + (i & (kRingSlots - 1)) * kDescDwords);
    uint32_t *s_stage = reinterpret_cast<uint32_t *>(s_runs + d.nr + 1);
    __syncthreads();   // the previous tile is done with LDS
    tile_stage<GT>(a, d, gtid, pf, s_runs, s_seg, s_mark);
    if (gtid < kDescUsed && i + 1 < n_my) s_ring[((i + 1) & (kRingSlots - 1)) * kDescDwords + gtid] = dn;
    __syncthreads();
    TileDesc d1 = {};
    if (i + 1 < n_my) d1 = ring_read(s_ring + ((i + 1) & (kRingSlots - 1)) * kDescDwords);
    tile_issue_loads<GT>(a, d1, gtid, pf);
    dn = 0;
    if (gtid < kDescUsed && i + 2 < n_my) dn = desc_words(i + 2)[gtid];
    process_tile<MODE, DMODE, GT>(a, d, s_runs, s_seg, s_mark, s_stage, s_dict, gtid, s_bits, i < n_my ? first + i * step : -1);
  }
}

constexpr int kLdsTotal = 160 * 1024;
constexpr int kStageMax = 18 * 1024;  // packed bytes of one tile, at most

static int clamp_stage(int stage_bytes) {
  int sb = (stage_bytes + 15) / 16 * 16;
  if (sb < 1024) sb = 1024;
  if (sb > kStageMax) sb = kStageMax;
  return sb;
}

// run_room: run-slice entries the pool has room for beside a tile's full complement of packed bytes
static void set_group_layout(ExpandArgs &a, int run_room, int stage_bytes) {
  a.pool_bytes = runs_lds_bytes(run_room) + clamp_stage(stage_bytes);
  a.runs_cap = a.pool_bytes / (int)sizeof(RunEntry) - 3;
  if (a.runs_cap > kRunSlice) a.runs_cap = kRunSlice;
  a.mark_cap = kTile / 8 + a.runs_cap + 8;
  a.group_bytes = kSegBytes + marks_lds_bytes(a.mark_cap) + a.pool_bytes + kRingSlots * (int)sizeof(TileDesc);   // the descriptor ring comes last
}

template <class K>
static hipError_t launch_expand(K kern, const ExpandArgs &a, int grid, int threads, size_t shmem, hipStream_t stream) {
  if (shmem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_tile_desc, dim3((a.n_tiles - a.tile_lo + 255) / 256), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), shmem, stream, a, a.descs);
  return hipGetLastError();
}

hipError_t launch_expand_levels(const ChunkCtx &c, const int32_t *tile_prefix, const int32_t *tile_stream, int n_tiles,
                                int stage_bytes, int32_t *rep_out, int32_t *def_out, TileDesc *descs, int n_cu,
                                uint32_t *valid_bits, int32_t *valid_cnt, int def_tile0, hipStream_t stream) {
  if (n_tiles <= 0) return hipSuccess;
  ExpandArgs a;
  a.valid_bits = valid_bits; a.valid_cnt = valid_cnt; a.def_tile0 = def_tile0;
  a.c = c; a.tile_prefix = tile_prefix; a.tile_stream = tile_stream; a.stream0 = 0; a.tile_lo = 0; a.n_tiles = n_tiles;
  a.rep_out = rep_out; a.def_out = def_out; a.dict = nullptr; a.dict_n = 0; a.no_dict = 0; a.dict_lds_n = 0; a.dict_lds_bytes = 0;
  a.values_out = nullptr; a.descs = descs; a.work_flag = nullptr;
  set_group_layout(a, kRunSlice, stage_bytes);
  const int lds = a.group_bytes + (valid_bits ? kBitsBytes : 0);
  int per_cu = kLdsTotal / (lds + 512);
  if (per_cu > 16) per_cu = 16;
  int grid = n_cu * per_cu;
  if (grid > n_tiles) grid = n_tiles;
  // 128 threads per tile: what a tile costs besides its items (descriptor, staging, counts) is spread over four items per
  // thread, not two (list levels, 128 Mi slots: 0.48 against 0.51 ms)
  return launch_expand(&k_hybrid_expand<0, 0, 1, 128>, a, grid, 128, (size_t)lds, stream);
}

hipError_t launch_expand_dict(const ChunkCtx &c, const int32_t *tile_prefix, const int32_t *tile_stream, int tile_lo, int tile_hi,
                              int stage_bytes, const uint8_t *dict, int32_t dict_n, uint8_t *values_out, TileDesc *descs,
                              int n_cu, int split_mode, hipStream_t stream) {
  if (tile_hi <= tile_lo) return hipSuccess;
  const int n_tiles = tile_hi - tile_lo;   // tiles of this launch (grid sizing)
  ExpandArgs a;
  a.c = c; a.tile_prefix = tile_prefix; a.tile_stream = tile_stream; a.stream0 = 2 * c.nd; a.tile_lo = tile_lo; a.n_tiles = tile_hi;
  a.no_dict = dict_n < 0;
  a.valid_bits = nullptr; a.valid_cnt = nullptr; a.def_tile0 = 0;
  if (dict_n < 0) dict_n = 0;
  a.rep_out = nullptr; a.def_out = nullptr; a.dict = dict; a.dict_n = dict_n; a.values_out = values_out; a.descs = descs;
  a.work_flag = c.chunk_status + kWorkFlag;
  const int esz = c.elem_size;
  const size_t dict_bytes = ((size_t)dict_n * esz + 15) / 16 * 16;
  const bool fixed48 = esz == 4 || esz == 8;

  // split_mode 0 / 1 pins the choice between the plain gather and the variant below that keeps the head of a mid-size
  // dictionary in LDS (tests); -1 takes the latter, which also finishes a batch of chunks decoded side by side a little
  // sooner (4.65 against 4.75 ms for the seven chunks of the benchmark) although each of its workgroups needs a whole CU.
  const bool no_split = split_mode == 0;
  if (dict_bytes <= 48 * 1024 || !fixed48 || dict_bytes > 1024 * 1024 || no_split) {
    // small dictionaries: whole dictionary in the LDS of every block, several 256-thread blocks per CU.
    // Large ones of other element sizes, and very large ones: gathered from L2 / HBM.
    const bool lds = dict_bytes > 0 && dict_bytes <= 48 * 1024;
    a.dict_lds_n = lds ? dict_n : 0;
    a.dict_lds_bytes = lds ? (int)dict_bytes : 0;
    set_group_layout(a, 254, stage_bytes);
    const size_t shmem = (size_t)a.dict_lds_bytes + a.group_bytes;
    int per_cu = (int)(kLdsTotal / (shmem + 512));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int grid = n_cu * per_cu;

    if (grid > n_tiles) grid = n_tiles;
    if (esz == 4) return lds ? launch_expand(&k_hybrid_expand<4, 1, 1, 256>, a, grid, 256, shmem, stream)
                             : launch_expand(&k_hybrid_expand<4, 0, 1, 256>, a, grid, 256, shmem, stream);
    if (esz == 8) return lds ? launch_expand(&k_hybrid_expand<8, 1, 1, 256>, a, grid, 256, shmem, stream)
                             : launch_expand(&k_hybrid_expand<8, 0, 1, 256>, a, grid, 256, shmem, stream);
    return lds ? launch_expand(&k_hybrid_expand<1, 1, 1, 256>, a, grid, 256, shmem, stream)
               : launch_expand(&k_hybrid_expand<1, 0, 1, 256>, a, grid, 256, shmem, stream);
  }

  // Mid-size dictionaries (48 KiB .. 1 MiB of 4- or 8-byte entries): one 1024-thread block per CU, two groups of
  // 512 threads, and all the LDS the tiles do not need holds the head of the dictionary; indices beyond it
  // are gathered from L2.  A random 4-byte gather is bound by the L2 request rate, so every entry that sits
  // in LDS counts.
  constexpr int G = 2, GT = 512;
  set_group_layout(a, 254, stage_bytes);
  int room = kLdsTotal - G * a.group_bytes - 256;
  room = room / 16 * 16;
  if (room < 0) room = 0;
  a.dict_lds_bytes = (size_t)room < dict_bytes ? room : (int)dict_bytes;
  a.dict_lds_n = (size_t)room >= dict_bytes ? dict_n : a.dict_lds_bytes / esz;
  const size_t shmem = (size_t)a.dict_lds_bytes + (size_t)G * a.group_bytes;
  int grid = n_cu;
  if (grid > (n_tiles + G - 1) / G) grid = (n_tiles + G - 1) / G;
  const bool whole = a.dict_lds_n >= dict_n;
  if (esz == 4) return whole ? launch_expand(&k_hybrid_expand<4, 1, G, GT>, a, grid, G * GT, shmem, stream)
                             : launch_expand(&k_hybrid_expand<4, 2, G, GT>, a, grid, G * GT, shmem, stream);
  return whole ? launch_expand(&k_hybrid_expand<8, 1, G, GT>, a, grid, G * GT, shmem, stream)
               : launch_expand(&k_hybrid_expand<8, 2, G, GT>, a, grid, G * GT, shmem, stream);
}

}  // namespace pqd
