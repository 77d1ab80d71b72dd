// RLE / bit-packed hybrid streams on gfx950: page setup, run-header prescan and the per-page bookkeeping
// around the tile expansion (k_expand.hip).
//
// Reference behaviour reproduced here:
//   encoding/encodingread.go:127-143 (varint), :146-167 (RLE run), :170-228 (bit-packed run),
//   :231-268 (hybrid), layout/page.go:644-667 (dictionary / level streams), :835-888 (levels and null
//   count), :139-164 (dictionary gather).
#include <hip/hip_runtime.h>
#include "pqd_kernels.h"

namespace pqd {

__device__ __forceinline__ int bit_len32(uint32_t x) { return x ? 32 - __builtin_clz(x) : 0; }

__device__ __forceinline__ void set_status(int32_t *p, int code) {
  // keep the first error a stream / page saw
  atomicCAS(p, 0, code);
}

// ---------------------------------------------------------------------------------------------
// k_page_setup: one thread per page.  Splits a data page body into its three sections exactly as
// layout.ReadPage does (v1: [i32 len][rep] [i32 len][def] values; v2: rep | def | values with the
// lengths in the header, layout/page.go:736-784, :835-881) and fills the level stream descriptors.
// ---------------------------------------------------------------------------------------------
__device__ int take_prefixed_v1(const uint8_t *in, int64_t &pos, int64_t end, Stream &s) {
  int64_t rem = end - pos;
  if (rem <= 0) return PQD_ERR_EOF;               // io.ReadFull with nothing left
  if (rem < 4) return PQD_ERR_UNEXPECTED_EOF;
  int32_t L = (int32_t)((uint32_t)in[pos] | (uint32_t)in[pos + 1] << 8 | (uint32_t)in[pos + 2] << 16 |
                        (uint32_t)in[pos + 3] << 24);
  pos += 4;
  if (L < 0) return PQD_ERR_MALFORMED;            // make([]byte, negative) panics
  if (pos >= end) return PQD_ERR_EOF;             // bytes.Reader.Read at the end, even for L == 0
  int64_t avail = end - pos < (int64_t)L ? end - pos : (int64_t)L;
  s.off = pos;
  s.len = L;
  s.avail = (int32_t)avail;
  pos += avail;
  return PQD_OK;
}

__global__ void k_page_setup(ChunkCtx c) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= c.n_pages) return;
  PageIn pi = c.pages[p];
  PageState ps;
  ps.status = 0; ps.n_nulls = 0; ps.n_rows = 0; ps.cnt = 0;
  ps.dense_base = 0; ps.byte_base = 0; ps.byte_len = 0; ps.val_off = 0; ps.val_len = 0; ps.aux0 = 0; ps.aux1 = 0;
  ps.delta_fast = 0; ps.spare = 0;
  if (pi.data_idx < 0) {
    if (pi.page_type != PQD_DICTIONARY_PAGE) ps.status = PQD_ERR_UNSUPPORTED_PAGE_TYPE;
    c.state[p] = ps;
    return;
  }
  const int d = pi.data_idx;
  Stream rs, ds, vs;
  Stream *all[3] = {&rs, &ds, &vs};
  for (int i = 0; i < 3; i++) {
    Stream &s = *all[i];
    const StreamSeed sd = c.seeds[i * c.nd + d];
    s.run_base = sd.run_base; s.run_cap = sd.run_cap; s.tile_base = sd.tile_base; s.pad = 0;
    s.off = 0; s.len = 0; s.avail = 0; s.w = 0; s.cnt = 0; s.out_base = pi.slot_base;
    s.page = p; s.n_runs = 0; s.status = 0; s.decoded = 0; s.direct = 0; s.spare = 0;
    s.dict_off = 0; s.dict_n = 0; s.dict_ok = 0;
  }
  int64_t pos = pi.body_off, end = pi.body_off + pi.body_len;
  const int64_t nv = pi.num_values;
  // status: what goes wrong up to and including the framing of the repetition levels; def_status: the framing of
  // the definition levels.  Kept apart because the reference decodes the repetition levels completely before it
  // looks at the definition levels (layout/page.go:835-881): an error inside the repetition level runs comes first.
  int status = PQD_OK, def_status = PQD_OK;
  if (nv < 0) status = PQD_ERR_MALFORMED;
  // a v2 page is cut into its sections before anything is read (layout/page.go:736-773): sections longer than the page
  // are a slice out of range there, whatever the page's value count
  if (pi.page_type == PQD_DATA_PAGE_V2 && (pi.rep_len < 0 || pi.def_len < 0 || (int64_t)pi.rep_len + pi.def_len > pi.body_len))
    status = PQD_ERR_MALFORMED;
  if (status == PQD_OK && nv > 0) {  // with num_values == 0 ReadDataPageValues returns before reading anything
    if (pi.page_type == PQD_DATA_PAGE) {
      if (c.max_rep > 0) {
        status = take_prefixed_v1(c.in, pos, end, rs);
        rs.w = bit_len32((uint32_t)c.max_rep);
        rs.cnt = (int32_t)nv;
      }
      if (status == PQD_OK && c.max_def > 0) {
        def_status = take_prefixed_v1(c.in, pos, end, ds);
        ds.w = bit_len32((uint32_t)c.max_def);
        ds.cnt = (int32_t)nv;
      }
    } else {  // DATA_PAGE_V2
      int64_t rll = pi.rep_len, dll = pi.def_len;
      {
        // The reference re-frames a non-empty section with a synthetic length prefix and then reads prefixed blocks
        // only for the levels the column has (layout/page.go:760-773, :835-881): a section that is empty although
        // the column has that level, or present although it has not, makes it read the next section (or the values)
        // in its place.  That reinterpretation is not reproduced: the page is refused.
        if ((c.max_rep > 0) != (rll > 0)) status = PQD_ERR_REFUSED;
        if (c.max_rep > 0) {
          rs.off = pos; rs.len = (int32_t)rll; rs.avail = (int32_t)rll;
          rs.w = bit_len32((uint32_t)c.max_rep); rs.cnt = (int32_t)nv;
        }
        if ((c.max_def > 0) != (dll > 0)) def_status = PQD_ERR_REFUSED;
        if (c.max_def > 0) {
          ds.off = pos + rll; ds.len = (int32_t)dll; ds.avail = (int32_t)dll;
          ds.w = bit_len32((uint32_t)c.max_def); ds.cnt = (int32_t)nv;
        }
        pos += rll + dll;
      }
    }
  }
  // (without definition levels there is no level stream to carry def_status to the page)
  ps.status = status != PQD_OK ? status : (c.max_def > 0 ? PQD_OK : def_status);
  ps.val_off = pos;
  ps.val_len = end - pos;
  rs.status = status;
  ds.status = status != PQD_OK ? status : def_status;
  vs.status = ds.status;
  c.streams[d] = rs;
  c.streams[c.nd + d] = ds;
  c.streams[2 * c.nd + d] = vs;
  c.state[p] = ps;
}

// ---------------------------------------------------------------------------------------------
// k_stream_tables: what the host otherwise writes per page and uploads - the three stream seeds of every data page,
// the tile prefixes of the level, index and plain expansions and the two tile -> stream maps - worked out from the
// uploaded page table by one workgroup: a running sum over the pages of the run-table room and the tiles each takes.
// For chunks of very many small pages (the reference's writer cuts pages at 8 KiB, writer/writer.go:63-65) the host's
// pass over the pages and the upload of its tables cost more than the kernels that decode the chunk.
// ---------------------------------------------------------------------------------------------
constexpr int kTablesThreads = 256;
constexpr int kTablesPer = 4;      // consecutive pages per thread
constexpr int kTablesPages = kTablesThreads * kTablesPer;   // pages per workgroup

struct TableAdds {
  int64_t x[4];      // what a page adds to: run room, tiles, run room of the pages whose values are an index stream, and - in one word, both
                     // fit 32 bits with their totals (the host has checked them) - tiles of those pages << 32 | plain tiles
  int64_t tiles, rcap;
  int d;
  bool want;
};

__device__ __forceinline__ TableAdds table_adds(const ChunkCtx &c, const TableBases &b, int64_t p) {
  TableAdds a;
  a.d = -1;
  int e = 0;
  int64_t nv = 0, body_len = 0;
  if (p < c.n_pages) {
    const PageIn &pi = c.pages[p];
    a.d = pi.data_idx; e = pi.encoding; nv = pi.num_values > 0 ? pi.num_values : 0; body_len = pi.body_len;
  }
  const bool data = a.d >= 0;
  a.tiles = data ? (nv + kTile - 1) / kTile : 0;
  a.rcap = data ? (nv < body_len ? nv : body_len) + 2 : 0;
  a.want = data && b.wanted[2] && (e == PQD_ENC_PLAIN_DICTIONARY || e == PQD_ENC_RLE_DICTIONARY || (e == PQD_ENC_RLE && b.rle_typed));
  int64_t plain_tiles = 0;
  if (data && b.wanted[2] && (e == PQD_ENC_PLAIN || (e == PQD_ENC_BYTE_STREAM_SPLIT && b.is_real))) {
    const int64_t units = c.physical_type == PQD_BOOLEAN ? nv : nv * (int64_t)c.elem_size;
    plain_tiles = (units + kPlainTile - 1) / kPlainTile;
  }
  a.x[0] = a.rcap; a.x[1] = a.tiles; a.x[2] = a.want ? a.rcap : 0; a.x[3] = ((a.want ? a.tiles : 0) << 32) | plain_tiles;
  return a;
}

// the four sums over the workgroup's threads, in every thread
__device__ __forceinline__ void tables_block_sum(int64_t (&v)[4], int64_t (*s_part)[4]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    if (lane == 0) s_part[wv][k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int64_t t = 0;
    for (int j = 0; j < kTablesThreads / 64; j++) t += s_part[j][k];
    v[k] = t;
  }
  __syncthreads();
}

// first the sums of every kTablesPages pages ...
__global__ __launch_bounds__(kTablesThreads) void k_stream_table_sums(ChunkCtx c, TableBases b, int64_t *sums) {
  __shared__ int64_t s_part[kTablesThreads / 64][4];
  int64_t mine[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < kTablesPer; i++) {
    const TableAdds a = table_adds(c, b, (int64_t)blockIdx.x * kTablesPages + (int64_t)threadIdx.x * kTablesPer + i);
#pragma unroll
    for (int k = 0; k < 4; k++) mine[k] += a.x[k];
  }
  tables_block_sum(mine, s_part);
  if (threadIdx.x < 4) sums[4 * (size_t)blockIdx.x + threadIdx.x] = mine[threadIdx.x];
}

// ... then the tables: a workgroup adds up the sums in front of its pages and numbers its own pages from there
__global__ __launch_bounds__(kTablesThreads) void k_stream_tables(ChunkCtx c, TableBases b, const int64_t *sums, StreamSeed *seeds, int32_t *tp_levels,
                                                                  int32_t *tp_dict, int32_t *tp_plain, int32_t *ts_levels, int32_t *ts_dict) {
  __shared__ int64_t s_part[kTablesThreads / 64][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nd = c.nd;
  int64_t before[4] = {0, 0, 0, 0};
  for (int j = tid; j < (int)blockIdx.x; j += kTablesThreads) {
#pragma unroll
    for (int k = 0; k < 4; k++) before[k] += sums[4 * (size_t)j + k];
  }
  tables_block_sum(before, s_part);
  TableAdds a[kTablesPer];
  int64_t mine[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < kTablesPer; i++) {
    a[i] = table_adds(c, b, (int64_t)blockIdx.x * kTablesPages + (int64_t)tid * kTablesPer + i);
#pragma unroll
    for (int k = 0; k < 4; k++) mine[k] += a[i].x[k];
  }
  int64_t ex[4];       // the sums in front of this thread's first page
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int64_t v = mine[k];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t up = __shfl_up(v, o, 64);
      if (lane >= o) v += up;
    }
    ex[k] = v - mine[k];
    if (lane == 63) s_part[wv][k] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int64_t t = before[k];
    for (int j = 0; j < wv; j++) t += s_part[j][k];
    ex[k] += t;
  }
#pragma unroll
  for (int i = 0; i < kTablesPer; i++) {
    if (a[i].d >= 0) {
      const int dd = a[i].d;
      const int64_t tiles = a[i].tiles, rcap = a[i].rcap;
      const int64_t acc_runs = ex[0], acc_tiles = ex[1], v_runs = ex[2], v_tiles = ex[3] >> 32, t_pl = ex[3] & 0xFFFFFFFFll;
      for (int g = 0; g < 2; g++) {
        StreamSeed sd;
        int32_t tp;
        if (b.wanted[g]) {
          sd.run_base = b.run_base[g] + acc_runs; sd.run_cap = (int32_t)rcap; sd.tile_base = (int32_t)(b.tile_base[g] + acc_tiles + dd);
          tp = (int32_t)(b.tlv_base[g] + acc_tiles);
          for (int64_t t = 0; t < tiles; t++) ts_levels[tp + t] = g * nd + dd;
        } else {
          sd.run_base = b.run_base[g]; sd.run_cap = 0; sd.tile_base = (int32_t)(b.tile_base[g] + dd);
          tp = (int32_t)b.tlv_base[g];
        }
        seeds[(size_t)g * nd + dd] = sd;
        tp_levels[(size_t)g * nd + dd] = tp;
      }
      StreamSeed sv;
      if (b.wanted[2]) {
        sv.run_base = b.run_base[2] + v_runs; sv.run_cap = (int32_t)(a[i].want ? rcap : 0); sv.tile_base = (int32_t)(b.tile_base[2] + v_tiles + dd);
        tp_dict[dd] = (int32_t)v_tiles;
        tp_plain[dd] = (int32_t)t_pl;
        if (a[i].want) for (int64_t t = 0; t < tiles; t++) ts_dict[v_tiles + t] = dd;
      } else {
        sv.run_base = b.run_base[2]; sv.run_cap = 0; sv.tile_base = (int32_t)(b.tile_base[2] + dd);
        tp_dict[dd] = 0; tp_plain[dd] = 0;
      }
      seeds[(size_t)2 * nd + dd] = sv;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) ex[k] += a[i].x[k];
  }
  // the end entries: by the thread that holds the chunk's last page (after it, ex holds the sums over all pages)
  if (blockIdx.x == gridDim.x - 1 && tid == kTablesThreads - 1) {
    const int64_t acc_tiles = ex[1];
    tp_levels[2 * (size_t)nd] = (int32_t)((b.wanted[0] ? acc_tiles : 0) + (b.wanted[1] ? acc_tiles : 0));
    tp_dict[nd] = b.wanted[2] ? (int32_t)(ex[3] >> 32) : 0;
    tp_plain[nd] = b.wanted[2] ? (int32_t)(ex[3] & 0xFFFFFFFFll) : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// k_hybrid_prescan: one wavefront per stream walks the varint run headers (the part of
// ReadRLEBitPackedHybrid that is inherently serial) and writes the run table, the tile -> run
// index and the stream totals.  The stream is staged through LDS in 4 KiB windows loaded with
// 128-bit accesses, the next window prefetched into registers while the current one is walked;
// every lane executes the same walk on wave-uniform scalars.
// ---------------------------------------------------------------------------------------------
constexpr int kWin = 4096;           // window bytes
constexpr int kWinVec = kWin / 16 + 4;  // 16-byte vectors per window incl. 64 bytes look-ahead
constexpr int kWinPerLane = (kWinVec + 63) / 64;

struct WinRegs {
  uint4 v[kWinPerLane];
};

// Loads the window starting at q-space position wq (q = stream position + a, a = misalignment of the
// stream start) into registers, zero beyond the stream's available bytes.
__device__ __forceinline__ void win_load(WinRegs &r, const uint8_t *base16, uint32_t wq, uint32_t q_avail_end,
                                          int lane) {
#pragma unroll
  for (int j = 0; j < kWinPerLane; j++) {
    int i = lane + 64 * j;
    uint4 x = make_uint4(0, 0, 0, 0);
    uint32_t qp = wq + 16u * (uint32_t)i;
    if (i < kWinVec && qp < q_avail_end) {
      x = *reinterpret_cast<const uint4 *>(base16 + qp);
      uint32_t left = q_avail_end - qp;  // valid bytes in this vector
      if (left < 16) {
        uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          int lb = (int)left - 4 * k;  // valid bytes in dword k
          if (lb <= 0) w[k] = 0;
          else if (lb < 4) w[k] &= (1u << (8 * lb)) - 1u;
        }
        x = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    r.v[j] = x;
  }
}

__device__ __forceinline__ void win_store(const WinRegs &r, uint4 *win, int lane) {
#pragma unroll
  for (int j = 0; j < kWinPerLane; j++) {
    int i = lane + 64 * j;
    if (i < kWinVec) win[i] = r.v[j];
  }
}

namespace {
__device__ __forceinline__ uint32_t dpp_shr32(uint32_t x, int n) {
  switch (n) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
  }
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_sum32(uint32_t x) {
#pragma unroll
  for (int n = 1; n <= 8; n <<= 1) x += dpp_shr32(x, n);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31
  return x;
}
}  // namespace

// Jump stretches (kJump): resolving a span of short runs at once
constexpr int kSpan = 2048;          // byte positions one jump step covers: 32 per lane
constexpr uint32_t kJumpRunBytes = 64;  // the step is taken while runs average fewer bytes than this ...
constexpr uint32_t kJumpStayBytes = 96; // ... and kept up while they average fewer than this

template <bool kJump>
__global__ __launch_bounds__(64) void k_hybrid_prescan(ChunkCtx c, int s0) {
  __shared__ uint4 win4[kWinVec];
  __shared__ __align__(16) uint16_t jt[kJump ? kSpan + 8 : 8];  // successor of every position of a span (jt[kSpan]: beyond it)
  __shared__ uint16_t lst[kJump ? kSpan / 2 + 64 : 8];            // the headers the walk visits, in order
  __shared__ uint32_t mk[kJump ? 64 : 1];                        // bit b of mk[l]: a run the successor table covers starts at position 32 l + b
  const uint32_t *win = reinterpret_cast<const uint32_t *>(win4);
  const int lane = threadIdx.x;
  Stream *sp = &c.streams[s0 + blockIdx.x];
  const int status_in = sp->status;
  const int cnt = sp->cnt;
  if (sp->direct) return;  // already expanded
  if (status_in != 0 || cnt <= 0) {
    if (lane == 0) { sp->n_runs = 0; sp->decoded = 0; }
    return;
  }
  if (lane == 0) c.chunk_status[kWorkFlag] = 1;   // a stream for the tile expansion
  const int w = sp->w;
  const int64_t off = sp->off;
  const uint32_t L = (uint32_t)sp->len, avail = (uint32_t)sp->avail;
  RunEntry *runs = c.runs + sp->run_base;
  const int run_cap = sp->run_cap;
  int2 *tile_run = reinterpret_cast<int2 *>(c.tile_run) + sp->tile_base;  // per tile: {run index, stream byte position}

  if (w > 32) {
    if (lane == 0) { sp->status = PQD_ERR_BIT_WIDTH; sp->n_runs = 0; }
    return;
  }

  const uint8_t *abs0 = c.in + off;
  const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(abs0) & 15u);
  const uint8_t *base16 = abs0 - a;
  const uint32_t q_avail_end = a + avail;
  const uint32_t qend = a + L;  // L < 2^31
  const uint32_t nb_rle = (uint32_t)(w + 7) / 8;
  const uint32_t ucnt = (uint32_t)cnt;

  uint32_t q = a;
  uint32_t pf_wq = a + 0x80000000u; // window held in the prefetch registers (none yet: q - pf_wq is huge)
  WinRegs pf;
  uint64_t out = 0;                // values decoded so far (saturates the walk at 2^40)
  uint32_t emit_end = 0;
  uint32_t emit_q = a;             // stream position (q-space) after the last emitted run
  int nr = 0;
  uint32_t ex = 0, ey = 0, ez = 0; // this lane's pending run entry
  uint32_t items = 0;              // work items of the runs emitted so far
  int status = PQD_OK;
  bool dense = false;              // most of the last 64 runs were one-byte-header RLE runs: try whole strides of them at once
  bool jumpy = false;              // the last 64 runs were short: resolve whole spans of them at once
  uint32_t flush_q = a;            // position after the run that completed the last group of 64

  // Outer loop: one LDS window [wq, wq + kWin + 64); inner loop: the runs whose header starts in
  // [wq, wq + kWin).  The next window is in flight (registers) while the current one is walked.
  while (q < qend && status == PQD_OK) {
    // the prefetched window when q falls inside it, else (first window, or a jump past it) a fresh one
    const uint32_t wq = (q - pf_wq < (uint32_t)kWin) ? pf_wq : (q & ~15u);
    if (wq != pf_wq) win_load(pf, base16, wq, q_avail_end, lane);
    __syncthreads();
    win_store(pf, win4, lane);
    pf_wq = wq + kWin;
    win_load(pf, base16, pf_wq, q_avail_end, lane);
    __syncthreads();

    // Hot loop: runs with a 1- or 2-byte header that start in this window at least 12 bytes before the end of
    // the stream, while values are still wanted.  Branch-free apart from the rare exits; everything is
    // wave-uniform and stays on the scalar unit.  Anything else falls through to the general loop below.
    if (w > 0) {
      uint32_t lim = (qend - 12u < wq + (uint32_t)kWin && qend >= 12u) ? qend - 12u : (qend >= 12u ? wq + (uint32_t)kWin : 0u);
      if (lim > q_avail_end) lim = q_avail_end;  // headers inside the zero fill of a short stream go the general way
      const uint32_t rle_mask = nb_rle >= 4 ? 0xFFFFFFFFu : ((1u << (8u * nb_rle)) - 1u);
      uint32_t o32 = out < (uint64_t)ucnt ? (uint32_t)out : ucnt;
      // (nr cannot reach run_cap here: every emitted run holds a value below cnt and takes a byte of the stream)
      for (;;) {
        // one run at a time; the loop bound drops to 0 when the 64 runs of a flush say the stream is dense
        uint32_t hot_lim = (dense || jumpy) ? 0u : lim;
        bool long_header = false;
        // Two phases per batch of up to 64 runs.  The chain itself - header at q, so the next header is at q + its
        // length + the run's bytes - is all that has to happen one run after the other, so the serial phase does
        // nothing else: it notes the positions (and keeps count of the values, to stop at the run that completes
        // the count).  What a run is and where its values and items start is then worked out for all noted runs at
        // once, one run per lane, with wave prefix sums.
        while (q < hot_lim && o32 < ucnt) {
          const int slot0 = nr & 63;                 // pending entries sit in lanes [0, slot0)
          const int cap = 64 - slot0;
          int n = 0;
          uint32_t qq = q, oo = o32;
          uint32_t posv = 0;                          // lane slot0 + k: position of the k-th run of the batch
          while (n < cap && qq < hot_lim && oo < ucnt) {
            const uint32_t di = (qq - wq) >> 2;
            const uint32_t sh = (qq & 3u) * 8u;
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(win[di]);
            const uint32_t d1 = __builtin_amdgcn_readfirstlane(win[di + 1]);
            const uint32_t x0 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh);
            if (__builtin_expect((x0 & 0x8080u) == 0x8080u, 0)) { long_header = true; break; }  // header of 3 or more bytes
            const uint32_t two = (x0 >> 7) & 1u;
            const uint32_t header = (x0 & 0x7Fu) | (two ? ((x0 >> 1) & 0x3F80u) : 0u);
            const uint32_t isbp = header & 1u;
            const uint32_t cg = header >> 1;
            const uint32_t nbytes = isbp ? cg * (uint32_t)w : nb_rle;
            const uint32_t nq = qq + 1u + two + nbytes;
            if (cg != 0u) {                           // (an empty run is stepped over)
              if (lane == slot0 + n) posv = qq;
              n++;
              oo += cg << (isbp * 3u);
            }
            qq = nq < qend ? nq : qend;  // only a bit-packed payload can reach past the end: it is zero filled
          }
          q = qq;
          if (n > 0) {
            const int i = lane - slot0;
            const bool in = i >= 0 && i < n;
            const uint32_t qi = in ? posv : wq;   // (idle lanes read the start of the window)
            const uint32_t dj = (qi - wq) >> 2, sj = (qi & 3u) * 8u;
            const uint32_t e0 = win[dj], e1 = win[dj + 1], e2 = win[dj + 2];
            const uint32_t y0 = __builtin_amdgcn_alignbit(e1, e0, sj), y1 = __builtin_amdgcn_alignbit(e2, e1, sj);
            const uint32_t two = (y0 >> 7) & 1u;
            const uint32_t header = (y0 & 0x7Fu) | (two ? ((y0 >> 1) & 0x3F80u) : 0u);
            const uint32_t hlen = 1u + two, isbp = header & 1u, cg = header >> 1;
            const uint32_t adv = hlen + (isbp ? cg * (uint32_t)w : nb_rle);
            const uint32_t val = (uint32_t)((((uint64_t)y1 << 32) | y0) >> (8u * hlen)) & rle_mask;
            const uint32_t e_y = isbp ? qi + hlen - a : val;
            const uint32_t nvi = in ? cg << (isbp * 3u) : 0u;
            const uint32_t incl_v = wave_incl_sum32(nvi);
            const uint32_t iti = in ? (nvi + 7u) >> 3 : 0u;
            const uint32_t incl_i = wave_incl_sum32(iti);
            if (in) {
              const uint32_t start = o32 + incl_v - nvi;
              ex = (isbp << 31) | start; ey = e_y; ez = items + incl_i - iti;
              const uint32_t endv = start + nvi, endc = endv < ucnt ? endv : ucnt;
              for (uint32_t kt = (start + kTile - 1) / kTile; kt * (uint32_t)kTile < endc; kt++)
                tile_run[kt] = make_int2(nr + i, (int)(isbp ? e_y + (uint32_t)(((uint64_t)(kt * (uint32_t)kTile - start) * (uint32_t)w) >> 3) : qi - a));
            }
            const int last = slot0 + n - 1;
            o32 += (uint32_t)__builtin_amdgcn_readlane((int)incl_v, last);
            items += (uint32_t)__builtin_amdgcn_readlane((int)incl_i, last);
            const uint32_t after = (uint32_t)__builtin_amdgcn_readlane((int)(qi + adv), last);
            nr += n;
            emit_end = o32;
            emit_q = after < qend ? after : qend;
            if ((nr & 63) == 0) {
              runs[nr - 64 + lane] = RunEntry{ex, ey, ez, 0u};
              // how many of these 64 runs were RLE runs of fewer than 64 values (a one-byte header)?
              const uint32_t nxt = (uint32_t)__shfl_down((int)(ex & 0x7FFFFFFFu), 1);
              const uint32_t nvl = (lane == 63 ? o32 : nxt) - (ex & 0x7FFFFFFFu);
              dense = __builtin_popcountll(__ballot((ex >> 31) == 0u && nvl < 64u)) >= 48;
              if (kJump) { jumpy = !dense && q - flush_q < 64u * kJumpRunBytes; flush_q = q; }
              if (dense || jumpy) hot_lim = 0u;
            }
          }
          if (long_header) break;
        }
        if (long_header || !(dense || jumpy)) break;
        if (kJump && !dense) {
          // Jump stretch.  Many short runs of any shape (Arrow-style level streams: a few values per run): the chain
          // of headers inside a span of kSpan bytes is resolved by all lanes together instead of one header at a
          // time.  Every lane works out, for each of its 32 byte positions, where the next header would be if a
          // header started there (1- or 2-byte headers of non-empty runs that end inside the stream; any other
          // position is its own successor and ends the chain).  Starting from the one position known to be a header,
          // the list of visited headers doubles in length with every squaring of that successor table (the header
          // len + i runs on is the len-th successor of the header i runs on), so log2(runs) rounds list exactly the
          // headers the one-at-a-time walk would visit, in order.  They are then recorded 64 per step with prefix
          // sums, cut at the run that completes the count.
          while (q < lim && o32 < ucnt) {
            const uint32_t q0 = q;
            const uint32_t pl = q0 + 32u * (uint32_t)lane;  // the first of this lane's positions
            uint32_t okw = 0;                               // bit k: a run the table covers starts at pl + k
            uint32_t jp[16];                                // successors of this lane's positions, two per dword
#pragma unroll
            for (int k = 0; k < 16; k++) {
              const uint32_t p = 32u * (uint32_t)lane + 2u * (uint32_t)k;
              jp[k] = p | ((p + 1u) << 16);
            }
            if (pl < lim) {
              const uint32_t dj = (pl - wq) >> 2, sj = (pl & 3u) * 8u;
              uint32_t d[10], b[9];
#pragma unroll
              for (int i = 0; i < 10; i++) d[i] = win[dj + i];
#pragma unroll
              for (int i = 0; i < 9; i++) b[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], sj);
#pragma unroll
              for (int k = 0; k < 32; k++) {
                const uint32_t x = ((k & 3) == 3 ? ((b[k >> 2] >> 24) | (b[(k >> 2) + 1] << 8)) : (b[k >> 2] >> (8 * (k & 3)))) & 0xFFFFu;
                const uint32_t two = (x >> 7) & 1u;
                const uint32_t header = (x & 0x7Fu) | (two ? ((x >> 1) & 0x3F80u) : 0u);
                const uint32_t isbp = header & 1u, cg = header >> 1;
                const uint32_t adv = 1u + two + (isbp ? cg * (uint32_t)w : nb_rle);
                const uint32_t pq = pl + (uint32_t)k;
                const bool ok = (x & 0x8080u) != 0x8080u && cg != 0u && pq < lim && pq + adv <= qend;
                if (ok) {
                  const uint32_t p = 32u * (uint32_t)lane + (uint32_t)k;
                  const uint32_t j = p + adv < (uint32_t)kSpan ? p + adv : (uint32_t)kSpan;
                  okw |= 1u << k;
                  jp[k >> 1] = (k & 1) ? ((jp[k >> 1] & 0xFFFFu) | (j << 16)) : ((jp[k >> 1] & 0xFFFF0000u) | j);
                }
              }
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
              reinterpret_cast<uint4 *>(jt)[4 * lane + i] = make_uint4(jp[4 * i], jp[4 * i + 1], jp[4 * i + 2], jp[4 * i + 3]);
            mk[lane] = okw;
            if (lane == 0) { jt[kSpan] = (uint16_t)kSpan; lst[0] = 0; lst[kSpan / 2] = (uint16_t)kSpan; }
            __syncthreads();
            // For the squaring every lane keeps the pairs of positions {128 k + 2 lane, + 1}: neighbouring lanes then
            // look up neighbouring table entries (with 32 consecutive positions per lane, as above, every lookup of a
            // wave would fall into four LDS banks).
            uint32_t *jt32 = reinterpret_cast<uint32_t *>(jt);
#pragma unroll
            for (int k = 0; k < 16; k++) jp[k] = jt32[64 * k + lane];
            // lst[d] = the header d runs after the one at q0.  With the table holding the len-th successors, the
            // entries [len, 2 len) follow from the entries [0, len); then the table is squared.  A span holds at
            // most kSpan / 2 runs; past the end of the chain the entries repeat its last position or kSpan.
            int len = 1;
            for (;;) {
              if (len <= 64) {
                if (lane < len) lst[len + lane] = jt[lst[lane]];
              } else if (len == 128) {
                const uint32_t p0 = lst[lane], p1 = lst[lane + 64];
                const uint16_t j0 = jt[p0], j1 = jt[p1];
                lst[128 + lane] = j0; lst[192 + lane] = j1;
              } else {
                for (int i = lane; i < len; i += 256) {   // (len is a multiple of 256 here)
                  const uint32_t p0 = lst[i], p1 = lst[i + 64], p2 = lst[i + 128], p3 = lst[i + 192];
                  const uint16_t j0 = jt[p0], j1 = jt[p1], j2 = jt[p2], j3 = jt[p3];
                  lst[len + i] = j0; lst[len + i + 64] = j1; lst[len + i + 128] = j2; lst[len + i + 192] = j3;
                }
              }
              __syncthreads();
              len *= 2;
              const uint32_t tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)lst[len - 1]);
              const bool ended = tail >= (uint32_t)kSpan || ((mk[tail >> 5] >> (tail & 31u)) & 1u) == 0u;
              if (__builtin_amdgcn_readfirstlane((int)ended) || len >= kSpan / 2) break;
              uint32_t jn[16];
#pragma unroll
              for (int k = 0; k < 32; k++) {
                const uint32_t jj = jt[(jp[k >> 1] >> (16 * (k & 1))) & 0xFFFFu];
                jn[k >> 1] = (k & 1) ? (jn[k >> 1] | (jj << 16)) : jj;
              }
              __syncthreads();
#pragma unroll
              for (int k = 0; k < 16; k++) { jt32[64 * k + lane] = jn[k]; jp[k] = jn[k]; }
              __syncthreads();
            }
            int taken = 0;
            while (o32 < ucnt) {
              const int slot0 = nr & 63;                 // pending entries sit in lanes [0, slot0)
              const int i = lane - slot0;
              const uint32_t p = (i >= 0 && taken + i < len) ? (uint32_t)lst[taken + i] : (uint32_t)kSpan;
              const bool valid = p < (uint32_t)kSpan && ((mk[p < (uint32_t)kSpan ? p >> 5 : 0u] >> (p & 31u)) & 1u) != 0u;
              const uint64_t vm = __ballot(valid) >> slot0;
              const int n_av = vm == ~0ull ? 64 : (int)__builtin_ctzll(~vm);   // the runs of the chain not yet taken, up to the free lanes
              if (n_av == 0) break;
              const bool in = i >= 0 && i < n_av;
              const uint32_t qi = q0 + (in ? p : 0u);
              const uint32_t dj = (qi - wq) >> 2, sj = (qi & 3u) * 8u;
              const uint32_t d0 = win[dj], d1 = win[dj + 1], d2 = win[dj + 2];
              const uint32_t x0 = __builtin_amdgcn_alignbit(d1, d0, sj), x1 = __builtin_amdgcn_alignbit(d2, d1, sj);
              const uint32_t two = (x0 >> 7) & 1u;
              const uint32_t header = (x0 & 0x7Fu) | (two ? ((x0 >> 1) & 0x3F80u) : 0u);
              const uint32_t hlen = 1u + two, isbp = header & 1u, cg = header >> 1;
              const uint32_t adv = hlen + (isbp ? cg * (uint32_t)w : nb_rle);
              const uint32_t val = (uint32_t)((((uint64_t)x1 << 32) | x0) >> (8u * hlen)) & rle_mask;
              const uint32_t e_y = isbp ? qi + hlen - a : val;
              const uint32_t nvi = in ? cg << (isbp * 3u) : 0u;
              const uint32_t incl_v = wave_incl_sum32(nvi);
              const uint64_t full = __ballot(in && incl_v >= ucnt - o32);   // the run that completes the count
              const int n_take = full ? (int)__builtin_ctzll(full) - slot0 + 1 : n_av;
              const bool mine = i >= 0 && i < n_take;
              const uint32_t iti = mine ? (nvi + 7u) >> 3 : 0u;
              const uint32_t incl_i = wave_incl_sum32(iti);
              if (mine) {
                const uint32_t start = o32 + incl_v - nvi;
                ex = (isbp << 31) | start; ey = e_y; ez = items + incl_i - iti;
                const uint32_t endv = start + nvi, endc = endv < ucnt ? endv : ucnt;
                for (uint32_t kt = (start + kTile - 1) / kTile; kt * (uint32_t)kTile < endc; kt++)
                  tile_run[kt] = make_int2(nr + i, (int)(isbp ? e_y + (uint32_t)(((uint64_t)(kt * (uint32_t)kTile - start) * (uint32_t)w) >> 3) : qi - a));
              }
              const int last = slot0 + n_take - 1;
              o32 += (uint32_t)__builtin_amdgcn_readlane((int)incl_v, last);
              items += (uint32_t)__builtin_amdgcn_readlane((int)incl_i, last);
              q = (uint32_t)__builtin_amdgcn_readlane((int)(qi + adv), last);
              nr += n_take;
              taken += n_take;
              if ((nr & 63) == 0) runs[nr - 64 + lane] = RunEntry{ex, ey, ez, 0u};
              emit_end = o32;
              emit_q = q;
            }
            __syncthreads();  // the list is rewritten by the next step
            // a header of another kind at q, or runs that have grown long: back to one run at a time
            if (taken == 0 || q - q0 > (uint32_t)taken * kJumpStayBytes) { jumpy = false; break; }
          }
          if (jumpy) break;   // the window or the count ended the stretch: the mode carries over
          flush_q = q;
          continue;
        }
        // Dense stretch (the reference's own writer emits nothing else, encodingwrite.go:149-214: one RLE run per
        // change of value, a one-byte header and the value): as long as every run is of that shape the headers
        // sit at a fixed stride, so every lane checks the header one stride further on and the leading lanes
        // whose headers all have the shape are taken at once - up to 64 runs per step instead of one.
        const uint32_t stride = 1u + nb_rle;
        while (q < lim && o32 < ucnt) {
          const int slot0 = nr & 63;                 // pending entries sit in lanes [0, slot0)
          const int i = lane - slot0;
          const uint32_t qi = q + (uint32_t)(i < 0 ? 0 : i) * stride;
          bool ok = i >= 0 && qi < lim;
          uint32_t hb = 0, vi = 0;
          if (ok) {
            const uint32_t dj = (qi - wq) >> 2, sj = (qi & 3u) * 8u;
            const uint64_t x = ((((uint64_t)win[dj + 1]) << 32) | win[dj]) >> sj;   // header and up to four value bytes
            hb = (uint32_t)x & 0xFFu;
            vi = (uint32_t)(x >> 8) & rle_mask;
            ok = (hb & 0x81u) == 0u && hb != 0u;     // one byte, RLE, not empty
          }
          const uint64_t okm = __ballot(ok) >> slot0;
          int n_ok = okm == ~0ull ? 64 : (int)__builtin_ctzll(~okm);
          if (n_ok > 64 - slot0) n_ok = 64 - slot0;
          if (n_ok < 2) { dense = false; break; }
          const uint32_t nvi = (i >= 0 && i < n_ok) ? hb >> 1 : 0u;
          const uint32_t incl_v = wave_incl_sum32(nvi);
          const uint64_t full = __ballot(i >= 0 && i < n_ok && incl_v >= ucnt - o32);   // the run that completes the count
          const int n_take = full ? (int)__builtin_ctzll(full) - slot0 + 1 : n_ok;
          const bool mine = i >= 0 && i < n_take;
          const uint32_t iti = mine ? (nvi + 7u) >> 3 : 0u;
          const uint32_t incl_i = wave_incl_sum32(iti);
          if (mine) {
            const uint32_t start = o32 + incl_v - nvi;
            ex = start; ey = vi; ez = items + incl_i - iti;
            const uint32_t endv = start + nvi, endc = endv < ucnt ? endv : ucnt;
            const uint32_t kt = (start + kTile - 1) & ~(uint32_t)(kTile - 1);
            if (kt < endc) tile_run[kt / kTile] = make_int2(nr + i, (int)(qi - a));   // a run of < 64 values meets at most one tile start
          }
          const int last = slot0 + n_take - 1;
          o32 += (uint32_t)__builtin_amdgcn_readlane((int)incl_v, last);
          items += (uint32_t)__builtin_amdgcn_readlane((int)incl_i, last);
          nr += n_take;
          q += (uint32_t)n_take * stride;
          if ((nr & 63) == 0) runs[nr - 64 + lane] = RunEntry{ex, ey, ez, 0u};
          emit_end = o32;
          emit_q = q;
          if (n_take < 8) { dense = false; break; }
        }
        if (dense) break;   // the window or the count ended the stretch: the mode carries over
      }
      if (out < (uint64_t)ucnt) out = o32;
    }

    while (q - wq < (uint32_t)kWin && q < qend) {
      if (q >= q_avail_end) {
        // Everything from here to the end of the stream is the zero fill of a short Read (a length prefix larger
        // than the page): headers of value 0, i.e. empty RLE runs of 1 + nb_rle bytes each.  Nothing more is
        // decoded; the walk ends cleanly unless the last header sits on the last byte, where ReadRLE's Read of the
        // value finds the buffer exhausted (encodingread.go:152-156).  Closed form instead of up to 2^30 steps.
        if (nb_rle > 0 && (qend - q) % (1u + nb_rle) == 1u) status = PQD_ERR_EOF;
        q = qend;
        break;
      }
      // 8 bytes at q, wave-uniform, on the scalar unit
      const uint32_t run_q = q;
      uint32_t di = (q - wq) >> 2;
      const uint32_t sh = (q & 3u) * 8u;
      uint32_t d0 = __builtin_amdgcn_readfirstlane(win[di]);
      uint32_t d1 = __builtin_amdgcn_readfirstlane(win[di + 1]);
      uint32_t d2 = __builtin_amdgcn_readfirstlane(win[di + 2]);
      uint32_t x0 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh);
      uint32_t x1 = (uint32_t)((((uint64_t)d2 << 32) | d1) >> sh);
      // varint header (encodingread.go:127-143).  Bytes past the end of the stream read as zero, which ends
      // the varint with the partial value the reference returns at EOF.
      uint32_t header, hlen;
      bool huge = false;  // a header of 2^31 or more: a run of at least 2^30 values
      if ((x0 & 0x80u) == 0) { header = x0 & 0x7Fu; hlen = 1; }
      else if ((x0 & 0x8000u) == 0) { header = (x0 & 0x7Fu) | ((x0 >> 1) & 0x3F80u); hlen = 2; }
      else {
        // longer header: byte by byte inside the 64-byte look-ahead
        uint64_t h64 = 0;
        uint32_t n = 0, shift = 0;
        for (;;) {
          const uint32_t qq = q + n;
          if (qq >= qend || n >= 56) break;
          const uint32_t b = (__builtin_amdgcn_readfirstlane(win[(qq - wq) >> 2]) >> ((qq & 3u) * 8u)) & 0xFFu;
          n++;
          if (shift < 64) h64 |= (uint64_t)(b & 0x7Fu) << shift;
          if ((b & 0x80u) == 0) break;
          shift += 7;
        }
        huge = (h64 >> 31) != 0;  // the reference runs out of memory on it, after ReadRLE has read the value (encodingread.go:152-165)
        header = (uint32_t)h64;
        hlen = n;
        // the bytes after the header, for an RLE value
        const uint32_t qv = q + n;
        di = (qv - wq) >> 2;
        const uint32_t shv = (qv & 3u) * 8u;
        d0 = __builtin_amdgcn_readfirstlane(win[di]);
        d1 = __builtin_amdgcn_readfirstlane(win[di + 1]);
        x0 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> shv) ;
        x1 = 0;
        // shift so that the common code below finds the value at byte `hlen`
        const uint64_t v64 = (uint64_t)x0;
        x0 = 0; x1 = 0;
        // store value bytes at byte offset hlen of (x1:x0) only when hlen <= 4; longer headers keep it aside
        if (n <= 4) { const uint64_t t = v64 << (8u * n); x0 = (uint32_t)t; x1 = (uint32_t)(t >> 32); }
        else { x0 = (uint32_t)v64; hlen = 0; q += n; if (q > qend) q = qend; }
      }
      const uint32_t qh = (q + hlen < qend) ? q + hlen : qend;  // position after the header
      const uint32_t c_or_g = header >> 1;
      uint32_t e_x, e_y, nv;
      if ((header & 1u) == 0) {  // RLE run, encodingread.go:146-167
        nv = c_or_g;
        uint32_t val = 0;
        q = qh;
        if (nb_rle > 0) {
          if (qh >= qend) { status = PQD_ERR_EOF; break; }   // Read at the end of the buffer
          uint32_t have = qend - qh < nb_rle ? qend - qh : nb_rle;  // a short read is zero filled
          val = (uint32_t)((((uint64_t)x1 << 32) | x0) >> (8u * hlen));
          if (have < 4) val &= (1u << (8u * have)) - 1u;
          q = qh + have;
        }
        e_x = 0; e_y = val;
        if (huge || out + nv > (1ull << 30)) { status = PQD_ERR_MALFORMED; break; }  // as for bit-packed runs below
      } else {  // bit-packed run, encodingread.go:170-228
        // more than 2^30 values held at once stands for the reference running out of memory (16 bytes per boxed value)
        if (huge || c_or_g > (1u << 27) || out + 8ull * c_or_g > (1ull << 30)) { status = PQD_ERR_MALFORMED; break; }
        nv = c_or_g * 8u;
        q = qh;
        if (nv == 0) continue;
        if (w > 0) {
          if (qh >= qend) { status = PQD_ERR_EOF; break; }   // Read at the end of the buffer
          e_x = 0x80000000u; e_y = qh - a;
          const uint64_t nbytes = (uint64_t)c_or_g * (uint32_t)w;
          const uint32_t left = qend - qh;
          q = qh + (nbytes < (uint64_t)left ? (uint32_t)nbytes : left);  // a payload cut short is zero filled
        } else {
          e_x = 0; e_y = 0;  // width 0: zeros, nothing consumed
        }
      }
      if (nv == 0) continue;
      if (out < (uint64_t)ucnt) {
        if (nr >= run_cap - 1) { status = PQD_ERR_NOMEM; break; }
        const uint32_t start = (uint32_t)out;
        const uint64_t e64 = out + nv;
        const uint32_t endc = e64 < (uint64_t)ucnt ? (uint32_t)e64 : ucnt;
        // tiles whose first value lies in this run
        const uint32_t kt = (start + kTile - 1) & ~(uint32_t)(kTile - 1);
        if (kt < endc)
          for (uint32_t k = kt / kTile + lane; k * (uint32_t)kTile < endc; k += 64)
            tile_run[k] = make_int2(nr, (int)((e_x >> 31) ? e_y + (uint32_t)(((uint64_t)(k * (uint32_t)kTile - start) * (uint32_t)w) >> 3) : run_q - a));
        if (lane == (nr & 63)) { ex = e_x | start; ey = e_y; ez = items; }
        nr++;
        items += (nv + 7u) >> 3;
        if ((nr & 63) == 0) runs[nr - 64 + lane] = RunEntry{ex, ey, ez, 0u};
        emit_end = e64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)e64;
        emit_q = q;
      }
      out += nv;
    }
  }
  // end sentinel, then flush the partial group
  const int n_runs = nr;
  if (lane == (nr & 63)) { ex = emit_end; ey = 0; ez = items; }
  nr++;
  if ((nr & 63) == 0) runs[nr - 64 + lane] = RunEntry{ex, ey, ez, 0u};
  {
    int pending = nr & 63;
    if (lane < pending) runs[nr - pending + lane] = RunEntry{ex, ey, ez, 0u};
  }
  if (status == PQD_OK && out < (uint64_t)cnt) status = PQD_ERR_MALFORMED;  // buf[:cnt] past the decoded values
  if (lane == 0) {
    sp->n_runs = n_runs;
    sp->pad = (int32_t)(emit_q - a);  // end of the bytes the emitted runs occupy
    sp->decoded = (int64_t)out;
    if (status != PQD_OK) sp->status = status;
  }
}

// ---------------------------------------------------------------------------------------------
// After the levels are expanded: per-page null / row / value counts (layout/page.go:883-888, :925-927), running
// bases, and the descriptor of the value section (dictionary pages: first byte is the index width, the rest is the
// hybrid stream, layout/page.go:644-655).  Three steps, so that a chunk of tens of thousands of small pages (the
// reference's writer cuts pages at 8 KiB) is not walked by a single workgroup: counts per page (one thread per
// page) summed per block of 256 pages, one workgroup scanning the block sums, then bases and value streams per page.
// totals: [0] slots [1] values [2] nulls [3] rows
// ---------------------------------------------------------------------------------------------
// block_sums: four totals (slots, values, nulls, rows) per block of 256 pages
__global__ __launch_bounds__(256) void k_pages_count(ChunkCtx c, int64_t *__restrict__ block_sums) {
  __shared__ int64_t s_part[4][4];
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t sum[4] = {0, 0, 0, 0};
  if (p < c.n_pages) {
    const PageIn pi = c.pages[p];
    if (pi.data_idx >= 0) {
      PageState *ps = &c.state[p];
      // a failed level stream fails the page
      int st = ps->status;
      if (st == 0 && c.max_rep > 0) st = c.streams[pi.data_idx].status;
      if (st == 0 && c.max_def > 0) st = c.streams[c.nd + pi.data_idx].status;
      const int64_t nv = pi.num_values > 0 ? pi.num_values : 0;
      const int64_t nulls = c.max_def > 0 ? ps->n_nulls : 0;
      const int64_t rows = c.max_rep > 0 ? ps->n_rows : nv;
      const int64_t cnt = st != 0 ? 0 : nv - nulls;
      ps->status = st;
      ps->n_nulls = (int32_t)nulls;
      ps->n_rows = (int32_t)rows;
      ps->cnt = (int32_t)cnt;
      sum[0] = nv; sum[1] = cnt; sum[2] = nulls; sum[3] = rows;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    int64_t t = sum[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
    if (lane == 0) s_part[q][wave] = t;
  }
  __syncthreads();
  if (threadIdx.x < 4) block_sums[4 * (int64_t)blockIdx.x + threadIdx.x] = s_part[threadIdx.x][0] + s_part[threadIdx.x][1] + s_part[threadIdx.x][2] + s_part[threadIdx.x][3];
}

// One workgroup: the totals of the call, and in place of every block's value count the values in front of the block.
// (Four wavefronts: a workgroup of sixteen waits for room on a GPU that other chunks keep busy.)
constexpr int kFinThreads = 256;
__global__ __launch_bounds__(kFinThreads) void k_pages_finalize(int n_blocks, int64_t *__restrict__ block_sums, int64_t *__restrict__ totals) {
  __shared__ int64_t s_wave[kFinThreads / 64];
  __shared__ unsigned long long s_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 4) s_tot[tid] = 0;
  __syncthreads();
  for (int b0 = 0; b0 < n_blocks; b0 += kFinThreads) {
    const int64_t carry = (int64_t)s_tot[1];
    const int b = b0 + tid;
    int64_t sum[4] = {0, 0, 0, 0};
    if (b < n_blocks) {
#pragma unroll
      for (int q = 0; q < 4; q++) sum[q] = block_sums[4 * (int64_t)b + q];
    }
    int64_t x = sum[1];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int64_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();   // every thread has read its carry from s_tot by now
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int64_t t = sum[q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
      if (lane == 0 && t != 0) atomicAdd(&s_tot[q], (unsigned long long)t);
    }
    int64_t base = carry + x - sum[1];
    for (int k = 0; k < wave; k++) base += s_wave[k];
    if (b < n_blocks) block_sums[4 * (int64_t)b + 1] = base;
    __syncthreads();   // s_wave and the read of s_tot[1] at the top of the next round
  }
  if (tid < 4) totals[tid] = (int64_t)s_tot[tid];
  // the mark the run-header walks leave when a stream needs the tile expansion (chunk_status[kWorkFlag]) starts clear
  if (tid == 0) reinterpret_cast<int32_t *>(totals)[kStatusWord + kWorkFlag] = 0;
}

__global__ __launch_bounds__(256) void k_pages_values(ChunkCtx c, const int64_t *__restrict__ block_sums) {
  __shared__ int64_t s_wave[4];
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool is_data = p < c.n_pages && c.pages[p].data_idx >= 0;
  // the page's place among the values of the call: the values in front of its block of pages, and an exclusive
  // scan of the counts inside the block
  const int64_t my_cnt = is_data ? c.state[p].cnt : 0;
  int64_t x = my_cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int64_t y = __shfl_up(x, o); if (lane >= o) x += y; }
  if (lane == 63) s_wave[wave] = x;
  __syncthreads();
  int64_t dense_base = block_sums[4 * (int64_t)blockIdx.x + 1] + x - my_cnt;
  for (int k = 0; k < wave; k++) dense_base += s_wave[k];
  if (!is_data) return;
  const PageIn pi = c.pages[p];
  PageState *ps = &c.state[p];
  ps->dense_base = dense_base;
  Stream *vs = &c.streams[2 * c.nd + pi.data_idx];
  const int64_t cnt = ps->cnt;
  vs->out_base = dense_base;
  vs->cnt = 0;
  vs->status = ps->status;
  const bool dict_enc = pi.encoding == PQD_ENC_PLAIN_DICTIONARY || pi.encoding == PQD_ENC_RLE_DICTIONARY;
  vs->dict_off = pi.dict_off; vs->dict_n = pi.dict_n; vs->dict_ok = dict_enc && pi.dict_page >= 0;
  if (ps->status == 0 && cnt > 0 && dict_enc) {
    const int64_t off = ps->val_off, len = ps->val_len;
    int st = 0;
    if (len <= 0) st = PQD_ERR_EOF;        // ReadByte at the end
    else if (len == 1) st = PQD_ERR_EOF;   // no run bytes: the reference then looks for a length prefix and hits EOF
    if (st == 0) {
      vs->w = c.in[off];
      vs->off = off + 1;
      vs->len = (int32_t)(len - 1);
      vs->avail = (int32_t)(len - 1);
      vs->cnt = (int32_t)cnt;
    } else {
      ps->status = st;
      vs->status = st;
    }
  }
  // Encoding_RLE values (layout/page.go:657-667): ReadRLEBitPackedHybrid with the column's type length as the bit
  // width and a 4-byte length prefix in front of the runs; the numbers are the values (int32(value) for INT32).
  // Typed output exists for BOOLEAN (low byte), INT32 and INT64; the other types are refused on the host.
  if (ps->status == 0 && cnt > 0 && pi.encoding == PQD_ENC_RLE &&
      (c.physical_type == PQD_BOOLEAN || c.physical_type == PQD_INT32 || c.physical_type == PQD_INT64)) {
    int64_t pos = ps->val_off;
    int st = c.type_length < 0 ? PQD_ERR_MALFORMED : take_prefixed_v1(c.in, pos, ps->val_off + ps->val_len, *vs);
    if (st == 0) {
      vs->w = c.type_length;
      vs->cnt = (int32_t)cnt;
    } else {
      ps->status = st;
      vs->status = st;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_rle_only_pages: dictionary-index streams that consist of nothing but RLE runs with a one-byte header - what the
// reference's own writer produces for every dictionary page (layout/dictpage.go:155-158 -> encodingwrite.go:149-214:
// width 32, one run per change of value, so 5 bytes for every value or two).  A run table of 16 bytes per run and an
// expansion by work items would move several times the bytes of the page; instead one workgroup per page checks
// that every header, at its fixed stride, has that shape (and that the runs hold enough values), and then writes the
// looked-up values straight to the output: run k starts at the sum of the counts before it.  Anything else - another
// header anywhere, a stream that does not end on a run boundary, zero fill, too few values - leaves the stream to the
// general path untouched, so the reference's behaviour on odd input stays in one place.
// ---------------------------------------------------------------------------------------------
constexpr int kDirectMaxLen = 64 * 1024;

// kLevels: the same for repetition / definition level streams (the reference writes those with WriteRLE too,
// layout/page.go:234-238); the values are the levels themselves, and the nulls (definition level below the maximum)
// or row starts (repetition level 0) are counted into the page state as the tile expansion would.
template <typename T, bool kLevels>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 8 : 6) void k_rle_only_pages(ChunkCtx c, const T *__restrict__ dict_table, T *__restrict__ values_out,
                                                        int32_t *__restrict__ rep_out, int32_t *__restrict__ def_out) {
  // One round takes 1536 runs, six consecutive runs per thread (a page of the reference's writer - 8 KiB, about 1400
  // runs - is one round).  The bytes of the stream come into LDS in 16-byte
  // pieces, a window of 7808 bytes at a time (a round's bytes always fit it); every thread parses its runs and asks for
  // their dictionary entries together; a scan of the counts gives the runs their places, and the values go out by
  // position - four consecutive values per thread, the run that covers the first of them found by bisection over
  // the places - so that the stores are as wide and as dense as the output allows.  Nothing is checked ahead of time:
  // a header of another shape anywhere, or too few values, and the stream is left to the general path, which writes
  // the same places again.
  constexpr int kWin = 7808;               // (bytes of a round and 15 of misalignment fit; with the other arrays eight workgroups of INT32 fit a CU's LDS)
  constexpr int kPer = 6;                  // runs per thread and round
  constexpr int kRound = 256 * kPer;
  __shared__ uint4 s_in4[kWin / 16 + 1];
  __shared__ uint32_t s_start[kRound + 1];
  __shared__ T s_val[kRound];
  __shared__ uint32_t s_sum[4];
  const uint32_t *s_in = reinterpret_cast<const uint32_t *>(s_in4);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int si = kLevels ? (int)blockIdx.x : 2 * c.nd + (int)blockIdx.x;
  const bool is_def = kLevels && si >= c.nd;
  Stream *sp = &c.streams[si];
  const int w = sp->w, cnt = sp->cnt, len = sp->len;
  if (sp->status != 0 || cnt <= 0 || w <= 0 || w > 32 || len != sp->avail || len <= 0 || len > kDirectMaxLen) return;
  const T *dict = nullptr;
  int dict_n = 0;
  if (!kLevels) {
    // dictionary-encoded pages only, looked up in the dictionary in effect for the page
    // (what the page entry says about it was copied into the stream: one dependent load less before the bytes are asked for)
    if (!sp->dict_ok) return;
    dict = dict_table + sp->dict_off;
    dict_n = sp->dict_n;
  }
  const uint32_t nb = (uint32_t)(w + 7) / 8, stride = 1u + nb;
  if ((uint32_t)len % stride) return;
  const int n_runs = (int)((uint32_t)len / stride);
  const uint32_t vmask = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8u * nb)) - 1u);
  const uint8_t *src = c.in + sp->off;
  const uint8_t *in_end = c.in + c.in_len;
  T *out = kLevels ? reinterpret_cast<T *>(is_def ? def_out : rep_out) + sp->out_base : values_out + sp->out_base;
  const uint32_t out_al = (uint32_t)((reinterpret_cast<uintptr_t>(out) / sizeof(T)) & 3u);   // elements past a 4-element boundary
  const uint32_t flag_val = is_def ? (uint32_t)c.max_def : 0u;
  uint32_t carry = 0;
  int bad = 0, bad_index = 0, n_flag = 0;
  int win_first = 0, win_runs = 0;         // runs [win_first, win_first + win_runs) have their bytes in LDS
  uint32_t win_al = 0;
  for (int k0 = 0; k0 < n_runs; k0 += kRound) {
    const int in_round = n_runs - k0 < kRound ? n_runs - k0 : kRound;
    __syncthreads();                       // the previous round's readers of s_in, s_start and s_val are done
    if (k0 + in_round > win_first + win_runs) {
      // ---- the next window of bytes, from this round on ----
      const uint8_t *r0 = src + (uint32_t)k0 * stride;
      win_al = (uint32_t)(reinterpret_cast<uintptr_t>(r0) & 15u);
      const int fit = (int)(((uint32_t)kWin - win_al) / stride);
      win_first = k0;
      win_runs = n_runs - k0 < fit ? n_runs - k0 : fit;      // (fit >= kRound: a round always fits)
      const uint32_t need = win_al + (uint32_t)win_runs * stride;
#pragma unroll
      for (int j = 0; j < (kWin / 16 + 255) / 256; j++) {
        const int ci = tid + 256 * j;
        if ((uint32_t)ci * 16u < need) {
          const uint8_t *q = r0 - win_al + 16 * ci;
          s_in4[ci] = q < in_end ? *reinterpret_cast<const uint4 *>(q) : make_uint4(0u, 0u, 0u, 0u);   // (its granule holds bytes of the buffer)
        }
      }
      __syncthreads();
    }
    // ---- kPer runs per thread ----
    uint32_t n[kPer], val[kPer], mine = 0;
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int kr = kPer * tid + j;
      n[j] = 0; val[j] = 0;
      if (kr < in_round) {
        const uint32_t bo = win_al + (uint32_t)(k0 - win_first + kr) * stride;
        const uint64_t x = (((uint64_t)s_in[(bo >> 2) + 1] << 32) | s_in[bo >> 2]) >> (8u * (bo & 3u));
        const uint32_t hb = (uint32_t)x & 0xFFu;
        bad |= (hb & 0x81u) != 0u || hb == 0u;   // one byte, RLE, not empty
        n[j] = hb >> 1;
        val[j] = (uint32_t)(x >> 8) & vmask;
      }
      mine += n[j];
    }
    T v[kPer];
    uint32_t oob = 0;       // runs whose index the dictionary does not hold
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      if (kLevels) {
        v[j] = (T)val[j];
      } else {
        uint32_t idx = val[j];
        if (n[j] > 0 && idx >= (uint32_t)dict_n) { oob |= 1u << j; idx = 0; }
        v[j] = (n[j] > 0 && dict_n > 0) ? dict[idx] : T(0);
      }
    }
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) s_sum[wv] = incl;
    __syncthreads();
    uint32_t before = 0, round = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { if (q < wv) before += s_sum[q]; round += s_sum[q]; }
    if (carry < (uint32_t)cnt) {
      uint32_t start = before + incl - mine;           // place of the thread's first run inside the round
#pragma unroll
      for (int j = 0; j < kPer; j++) {
        if (kLevels && n[j] > 0 && carry + start < (uint32_t)cnt) {
          const uint32_t end = carry + start + n[j] < (uint32_t)cnt ? start + n[j] : (uint32_t)cnt - carry;
          if (is_def ? val[j] != flag_val : val[j] == flag_val) n_flag += (int)(end - start);
        }
        // dict.Values[index] out of range (layout/page.go:139-164) - for a run the page's count reaches: the reference
        // stops reading the stream once it has the values it needs
        if (((oob >> j) & 1u) && carry + start < (uint32_t)cnt) bad_index = 1;
        s_start[kPer * tid + j] = start;
        s_val[kPer * tid + j] = v[j];
        start += n[j];
      }
      __syncthreads();
      // ---- the round's values, by position ----
      const uint32_t limit = carry + round < (uint32_t)cnt ? round : (uint32_t)cnt - carry;   // values of the round that are wanted
      const uint32_t shift = (out_al + carry) & 3u;       // the round's first value sits this far into a group of four
      for (uint32_t g0 = 0; g0 * 4u < limit + shift; g0 += 256u) {
        const uint32_t g = g0 + (uint32_t)tid;
        const int64_t p0 = (int64_t)g * 4 - (int64_t)shift;            // position of the group's first element in the round
        if (p0 < (int64_t)limit && p0 + 4 > 0) {
          // the run that covers the first wanted position of the group
          const uint32_t first = p0 < 0 ? 0u : (uint32_t)p0;
          int lo = 0, hi = in_round - 1;
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_start[mid] <= first) lo = mid; else hi = mid - 1;
          }
          int k = lo;
          T vv[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int64_t pos = p0 + j;
            vv[j] = T(0);
            if (pos >= 0 && pos < (int64_t)limit) {
              while (k + 1 < in_round && s_start[k + 1] <= (uint32_t)pos) k++;
              vv[j] = s_val[k];
            }
          }
          T *dst = out + (int64_t)carry + p0;
          if (p0 >= 0 && p0 + 4 <= (int64_t)limit) {
            if (sizeof(T) == 4) {
              *reinterpret_cast<uint4 *>(dst) = make_uint4((uint32_t)vv[0], (uint32_t)vv[1], (uint32_t)vv[2], (uint32_t)vv[3]);
            } else {
              reinterpret_cast<uint4 *>(dst)[0] = make_uint4((uint32_t)vv[0], (uint32_t)((uint64_t)vv[0] >> 32), (uint32_t)vv[1], (uint32_t)((uint64_t)vv[1] >> 32));
              reinterpret_cast<uint4 *>(dst)[1] = make_uint4((uint32_t)vv[2], (uint32_t)((uint64_t)vv[2] >> 32), (uint32_t)vv[3], (uint32_t)((uint64_t)vv[3] >> 32));
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (p0 + j >= 0 && p0 + j < (int64_t)limit) dst[j] = vv[j];
          }
        }
      }
    }
    carry += round;
  }
  // ---- the verdict ----
  for (int o = 32; o > 0; o >>= 1) bad |= __shfl_down(bad, o);
  __syncthreads();
  if (lane == 0) s_sum[wv] = (uint32_t)bad;
  __syncthreads();
  bad = (int)(s_sum[0] | s_sum[1] | s_sum[2] | s_sum[3]);
  if (bad || carry < (uint32_t)cnt) return;     // (uniform) not this shape, or too few values: the general path
  if (kLevels) {
    for (int o = 32; o > 0; o >>= 1) n_flag += __shfl_down(n_flag, o);
    if (lane == 0 && n_flag) atomicAdd(is_def ? &c.state[sp->page].n_nulls : &c.state[sp->page].n_rows, n_flag);
  }
  if (__syncthreads_or(bad_index)) { if (tid == 0) set_status(&c.state[sp->page].status, PQD_ERR_MALFORMED); }
  if (tid == 0) { sp->direct = 1; sp->n_runs = 0; sp->decoded = (int64_t)carry; }
}

hipError_t launch_rle_only_pages(const ChunkCtx &c, const uint8_t *dict_table, uint8_t *values_out, hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  if (c.elem_size == 4)
    hipLaunchKernelGGL((k_rle_only_pages<uint32_t, false>), dim3(c.nd), dim3(256), 0, stream, c, reinterpret_cast<const uint32_t *>(dict_table),
                       reinterpret_cast<uint32_t *>(values_out), (int32_t *)nullptr, (int32_t *)nullptr);
  else if (c.elem_size == 8)
    hipLaunchKernelGGL((k_rle_only_pages<uint64_t, false>), dim3(c.nd), dim3(256), 0, stream, c, reinterpret_cast<const uint64_t *>(dict_table),
                       reinterpret_cast<uint64_t *>(values_out), (int32_t *)nullptr, (int32_t *)nullptr);
  return hipGetLastError();
}

hipError_t launch_rle_only_levels(const ChunkCtx &c, int32_t *rep_out, int32_t *def_out, hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_rle_only_pages<int32_t, true>), dim3(2 * c.nd), dim3(256), 0, stream, c, (const int32_t *)nullptr, (int32_t *)nullptr,
                     rep_out, def_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_collect_status: first failing page of the chunk, in page order (the reference stops at the first
// page whose ReadPage returns an error).  out[0] = status, out[1] = page index or -1.
// ---------------------------------------------------------------------------------------------
// out[kDirectCount]: data pages whose index stream was expanded straight from the page (k_rle_only_pages, k_page_fused).
__global__ __launch_bounds__(256) void k_collect_status(ChunkCtx c, int32_t *out) {
  __shared__ int s_first, s_direct;
  if (threadIdx.x == 0) { s_first = 0x7FFFFFFF; s_direct = 0; }
  __syncthreads();
  int n_direct = 0;
  for (int p = threadIdx.x; p < c.n_pages; p += 256) {
    int st = c.state[p].status;
    const int d = c.pages[p].data_idx;
    if (d >= 0) n_direct += c.streams[2 * c.nd + d].direct != 0;
    if (st == 0 && d >= 0) st = c.streams[2 * c.nd + d].status;
    if (st != 0) { atomicMin(&s_first, p); c.state[p].status = st; }
  }
  if (n_direct) atomicAdd(&s_direct, n_direct);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_first == 0x7FFFFFFF) { out[0] = 0; out[1] = -1; }
    else { out[0] = c.state[s_first].status; out[1] = s_first; }
    out[kDirectCount] = s_direct;
  }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_page_setup(const ChunkCtx &c, hipStream_t stream) {
  if (c.n_pages <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_page_setup, dim3((c.n_pages + 255) / 256), dim3(256), 0, stream, c);
  return hipGetLastError();
}

size_t stream_tables_work_bytes(int n_pages) { return 32 * ((size_t)(n_pages > 0 ? n_pages : 0) / kTablesPages + 2); }

hipError_t launch_stream_tables(const ChunkCtx &c, const TableBases &b, void *work, StreamSeed *seeds, int32_t *tp_levels, int32_t *tp_dict, int32_t *tp_plain,
                                int32_t *ts_levels, int32_t *ts_dict, hipStream_t stream) {
  if (c.n_pages <= 0) return hipSuccess;
  const int nb = (c.n_pages + kTablesPages - 1) / kTablesPages;
  int64_t *sums = static_cast<int64_t *>(work);
  hipLaunchKernelGGL(k_stream_table_sums, dim3(nb), dim3(kTablesThreads), 0, stream, c, b, sums);
  hipLaunchKernelGGL(k_stream_tables, dim3(nb), dim3(kTablesThreads), 0, stream, c, b, sums, seeds, tp_levels, tp_dict, tp_plain, ts_levels, ts_dict);
  return hipGetLastError();
}

hipError_t launch_hybrid_prescan(const ChunkCtx &c, int s0, int n, bool wide, bool jump, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (wide) return launch_hybrid_prescan_wide(c, s0, n, stream);
  if (jump) hipLaunchKernelGGL(k_hybrid_prescan<true>, dim3(n), dim3(64), 0, stream, c, s0);
  else hipLaunchKernelGGL(k_hybrid_prescan<false>, dim3(n), dim3(64), 0, stream, c, s0);
  return hipGetLastError();
}

// the same over many workgroups, for chunks of many pages: out[1] starts as a large sentinel (0x7F7F7F7F)
__global__ __launch_bounds__(256) void k_collect_status_wide(ChunkCtx c, int32_t *out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= c.n_pages) return;
  int st = c.state[p].status;
  const int d = c.pages[p].data_idx;
  if (d >= 0 && c.streams[2 * c.nd + d].direct != 0) atomicAdd(&out[kDirectCount], 1);
  if (st == 0 && d >= 0) st = c.streams[2 * c.nd + d].status;
  if (st != 0) { atomicMin(&out[1], p); c.state[p].status = st; }
}
__global__ void k_collect_status_done(ChunkCtx c, int32_t *out) {
  if (out[1] == 0x7F7F7F7F) { out[0] = 0; out[1] = -1; }
  else out[0] = c.state[out[1]].status;
}

hipError_t launch_collect_status(const ChunkCtx &c, int32_t *out, hipStream_t stream) {
  if (c.n_pages <= 4096) {
    hipLaunchKernelGGL(k_collect_status, dim3(1), dim3(256), 0, stream, c, out);   // (few wavefronts: placed at once on a busy GPU)
  } else {
    hipError_t e = hipMemsetAsync(out + 1, 0x7F, 4, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(out + kDirectCount, 0, 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_collect_status_wide, dim3((c.n_pages + 255) / 256), dim3(256), 0, stream, c, out);
    hipLaunchKernelGGL(k_collect_status_done, dim3(1), dim3(1), 0, stream, c, out);
  }
  return hipGetLastError();
}

hipError_t launch_pages_finalize(const ChunkCtx &c, int64_t *totals, int64_t *block_sums, hipStream_t stream) {
  const int n_blocks = (c.n_pages + 255) / 256;
  if (n_blocks > 0) hipLaunchKernelGGL(k_pages_count, dim3(n_blocks), dim3(256), 0, stream, c, block_sums);
  hipLaunchKernelGGL(k_pages_finalize, dim3(1), dim3(kFinThreads), 0, stream, n_blocks, block_sums, totals);
  if (n_blocks > 0) hipLaunchKernelGGL(k_pages_values, dim3(n_blocks), dim3(256), 0, stream, c, block_sums);
  return hipGetLastError();
}

}  // namespace pqd
