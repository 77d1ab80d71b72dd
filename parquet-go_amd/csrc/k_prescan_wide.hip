// Run-header prescan of RLE / bit-packed hybrid streams, one workgroup per stream (few, long streams).
//
// The walk over the run headers is a dependent chain: the position of a header is known only once the header
// before it has been read (encoding/encodingread.go:231-268).  What can be taken off that chain is the decoding:
// builder waves compute, for EVERY byte position of a window, where the next header would be and how many
// values the run would hold if a header started there; the follower wave then only hops through that table
// (one LDS read and two additions per run) and notes the positions it visited.  The notes are turned into
// run-table entries 64 at a time by all lanes of the follower (prefix sums for the output positions and the
// work-item numbering, tile table entries where a run crosses a tile boundary).  Headers the table does not
// cover (three or more bytes, empty runs, runs reaching the end of the stream, width 0) take the same
// one-at-a-time path as in k_hybrid_prescan, with the reference's end-of-buffer semantics.
//
// Window k of a stream covers positions [k * kWin, (k + 1) * kWin) counted from the 16-byte aligned address at
// or below the stream start.  While the follower hops through window k, the builders prepare window k + 1
// (raw bytes and table are double buffered) and their loads for windows k + 2 .. k + 4 are in flight.
#include <hip/hip_runtime.h>
#include "pqd_kernels.h"

namespace pqd {

namespace {

constexpr int kWin = 4096;                  // bytes per window
constexpr int kUnit = 8;                    // bytes a builder thread turns into table entries per window
constexpr int kBuilders = kWin / kUnit;     // 512 builder threads
constexpr int kLook = 64;                   // bytes of the next window kept behind each window
constexpr int kThreads = 64 + kBuilders;    // wave 0 follows
constexpr int kRawDwords = (kWin + kLook) / 4;
constexpr int kAhead = 3;                   // windows whose loads are in flight

struct Shared {
  uint2 tab[2][kWin];                 // per position: x = bytes to the next header (0: not covered), y = values | bit_packed << 31
  uint32_t raw[2][kRawDwords];
  uint32_t q;                         // follower position after the last barrier
  int32_t done;
};

// 8 bytes at q-space position qp (a multiple of 8), or the first 8 bytes when qp lies at or beyond q_avail_end.  The bytes of a
// partly valid load are masked when the value is used (mask8), not here: masking at once would make every load
// wait for its data and defeat the loads in flight.
__device__ __forceinline__ uint2 load8(const uint8_t *base16, uint32_t qp, uint32_t q_avail_end) {
  // (an unconditional load from a clamped address: with branches around the loads the compiler can no longer count
  //  the ones in flight and waits for all of them)
  return *reinterpret_cast<const uint2 *>(base16 + (qp < q_avail_end ? qp : 0u));
}
__device__ __forceinline__ uint2 mask8(uint2 x, uint32_t qp, uint32_t q_avail_end) {
  if (qp >= q_avail_end) return make_uint2(0, 0);
  const uint32_t left = q_avail_end - qp;
  if (left < 8) {
    const uint64_t m = (1ull << (8 * left)) - 1ull;
    x.x &= (uint32_t)m;
    x.y &= (uint32_t)(m >> 32);
  }
  return x;
}

struct Pre {     // one window's worth of a builder thread's loads
  uint2 own;     // its 8 bytes
  uint2 nxt;     // the 8 bytes after them
  uint2 look;    // threads 0..7: their 8 bytes of the look-ahead
};

__device__ __forceinline__ Pre pre_load(const uint8_t *base16, uint32_t wq, uint32_t q_avail_end, int u) {
  Pre p;
  p.own = load8(base16, wq + 8u * (uint32_t)u, q_avail_end);
  p.nxt = load8(base16, wq + 8u * (uint32_t)u + 8u, q_avail_end);
  p.look = load8(base16, wq + (uint32_t)kWin + 8u * (uint32_t)(u & (kLook / 8 - 1)), q_avail_end);
  return p;
}
__device__ __forceinline__ Pre pre_mask(Pre p, uint32_t wq, uint32_t q_avail_end, int u) {
  p.own = mask8(p.own, wq + 8u * (uint32_t)u, q_avail_end);
  p.nxt = mask8(p.nxt, wq + 8u * (uint32_t)u + 8u, q_avail_end);
  if (u < kLook / 8) p.look = mask8(p.look, wq + (uint32_t)kWin + 8u * (uint32_t)u, q_avail_end);
  return p;
}

__device__ __forceinline__ uint32_t dpp_shr(uint32_t x, int n) {
  switch (n) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
  }
}
// lane `sel` of `old` replaced by the wave-uniform value `val`
__device__ __forceinline__ uint32_t write_lane(uint32_t old, uint32_t val, int sel) {
  asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(val), "s"(sel));  // (m0 is not otherwise used in this kernel)
  return old;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
#pragma unroll
  for (int n = 1; n <= 8; n <<= 1) x += dpp_shr(x, n);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31
  return x;
}

}  // namespace

__global__ __launch_bounds__(kThreads) void k_hybrid_prescan_wide(ChunkCtx c, int s0) {
  extern __shared__ uint4 smem_wide[];
  Shared &sh = *reinterpret_cast<Shared *>(smem_wide);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const bool follower = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;  // wave-uniform, and known to be: the follower's state stays scalar
  const int u = tid - 64;  // builder unit
  Stream *sp = &c.streams[s0 + blockIdx.x];
  const int status_in = sp->status;
  const int cnt = sp->cnt;
  if (sp->direct) return;  // already expanded (k_rle_only_pages)
  if (status_in != 0 || cnt <= 0) {
    if (tid == 0) { sp->n_runs = 0; sp->decoded = 0; }
    return;
  }
  if (tid == 0) c.chunk_status[kWorkFlag] = 1;   // a stream for the tile expansion
  const int w = sp->w;
  if (w > 32) {
    if (tid == 0) { sp->status = PQD_ERR_BIT_WIDTH; sp->n_runs = 0; }
    return;
  }
  const int64_t off = sp->off;
  const uint32_t L = (uint32_t)sp->len, avail = (uint32_t)sp->avail;
  RunEntry *runs = c.runs + sp->run_base;
  const int run_cap = sp->run_cap;
  int2 *tile_run = reinterpret_cast<int2 *>(c.tile_run) + sp->tile_base;  // per tile: {run index, stream byte position}

  const uint8_t *abs0 = c.in + off;
  const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(abs0) & 15u);
  const uint8_t *base16 = abs0 - a;
  const uint32_t q_avail_end = a + avail;
  const uint32_t qend = a + L;  // L < 2^31
  const uint32_t nb_rle = (uint32_t)(w + 7) / 8;
  const uint32_t rle_mask = nb_rle >= 4 ? 0xFFFFFFFFu : ((1u << (8u * nb_rle)) - 1u);
  const uint32_t ucnt = (uint32_t)cnt;
  const uint32_t lim = qend >= 12u ? qend - 12u : 0u;  // the table covers headers that start before this position

  if (tid == 0) { sh.q = a; sh.done = 0; }

  // ---- follower state (wave 0; every lane holds the same scalars) ----
  uint32_t q = a;
  uint64_t out = 0;                // values of all runs walked so far (saturates at 2^40)
  uint32_t emit_q = a;             // position after the last recorded run
  int nr = 0;                      // runs recorded
  int nr_flushed = 0;
  uint32_t base_out = 0, base_items = 0;
  uint32_t rq = 0, rx = 0, ry = 0; // this lane's record: header position, values | bit_packed << 31, RLE value or payload offset
  uint64_t dec_mask = 0;           // bit i: pending record i already holds rx / ry (else they are read from the table
                                   // and the raw bytes before the window goes away)
  int status = PQD_OK;
  bool dense = false;              // most of the last 64 runs were one-byte-header RLE runs: try whole strides of them at once
  const uint32_t stride = 1u + nb_rle;

  // ---- builder state ----
  Pre pf[kAhead];
  int kb = 0;  // window being built
  if (!follower) {
#pragma unroll
    for (int i = 0; i < kAhead; i++) pf[i] = pre_load(base16, (uint32_t)(kb + i) * kWin, q_avail_end, u);
  }

  // Turns the records not yet decoded (header positions inside window `wq`, buffer b) into rx / ry.
  auto finalize = [&](uint32_t wq, int b) {
    const int pending = nr - nr_flushed;
    if (lane < pending && !((dec_mask >> lane) & 1ull)) {
      const uint32_t o = rq - wq;
      rx = sh.tab[b][o].y;
      const uint32_t d0 = sh.raw[b][o >> 2], d1 = sh.raw[b][(o >> 2) + 1], d2 = sh.raw[b][(o >> 2) + 2];
      const uint32_t s8 = (o & 3u) * 8u;
      const uint32_t x0 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> s8);
      const uint32_t x1 = (uint32_t)((((uint64_t)d2 << 32) | d1) >> s8);
      const uint32_t hlen = 1u + ((x0 >> 7) & 1u);
      const uint32_t val = (uint32_t)((((uint64_t)x1 << 32) | x0) >> (8u * hlen)) & rle_mask;
      ry = (rx >> 31) ? rq + hlen - a : val;
    }
    dec_mask |= pending >= 64 ? ~0ull : ((1ull << pending) - 1ull);  // only the records that exist: lanes beyond them are free
  };

  // Writes the pending records (at most 64, all decoded) to the run table.
  auto flush = [&]() {
    const int pending = nr - nr_flushed;
    if (pending <= 0) return;
    if (nr + 1 > run_cap) { status = PQD_ERR_NOMEM; return; }
    const bool mine = lane < pending;
    const uint32_t nv = mine ? (rx & 0x7FFFFFFFu) : 0u;
    const uint32_t items = (nv + 7u) >> 3;
    const uint32_t incl_v = wave_incl_sum(nv), incl_i = wave_incl_sum(items);
    const uint32_t start = base_out + incl_v - nv;
    if (mine) {
      runs[nr_flushed + lane] = RunEntry{(rx & 0x80000000u) | start, ry, base_items + incl_i - items, 0u};
      // tiles whose first value lies in this run
      const uint32_t endv = start + nv;
      const uint32_t endc = endv < ucnt ? endv : ucnt;
      for (uint32_t kt = (start + kTile - 1) / kTile; kt * (uint32_t)kTile < endc; kt++) {
        const int pos = (rx >> 31) ? (int)(ry + (uint32_t)(((uint64_t)(kt * (uint32_t)kTile - start) * (uint32_t)w) >> 3))
                                   : (int)(rq - a);
        tile_run[kt] = make_int2(nr_flushed + lane, pos);
      }
    }
    base_out += (uint32_t)__builtin_amdgcn_readlane((int)incl_v, 63);
    base_items += (uint32_t)__builtin_amdgcn_readlane((int)incl_i, 63);
    // an RLE run of fewer than 64 values has a one-byte header
    if (pending == 64) dense = __builtin_popcountll(__ballot((rx >> 31) == 0u && nv < 64u)) >= 48;
    nr_flushed = nr;
    dec_mask = 0;
  };

  auto record_decoded = [&](uint32_t run_q, uint32_t e_x, uint32_t e_y) {
    const int np = nr - nr_flushed;
    rq = write_lane(rq, run_q, np);
    rx = write_lane(rx, e_x, np);
    ry = write_lane(ry, e_y, np);
    dec_mask |= 1ull << np;
    nr++;
  };

  // ---- pipeline: iteration i prepares window kb while the follower walks window kb - 1 ----
  bool all_done = false;
  while (!all_done) {
    const int kf = kb - 1;  // window the follower walks in this iteration (none in the first)
    if (!follower) {
      // raw bytes and table of window kb
      const int b = kb & 1;
      const Pre p = pre_mask(pf[0], (uint32_t)kb * kWin, q_avail_end, u);
#pragma unroll
      for (int i = 0; i + 1 < kAhead; i++) pf[i] = pf[i + 1];
      pf[kAhead - 1] = pre_load(base16, (uint32_t)(kb + kAhead) * kWin, q_avail_end, u);
      reinterpret_cast<uint2 *>(sh.raw[b])[u] = p.own;
      if (u < kLook / 8) reinterpret_cast<uint2 *>(sh.raw[b])[kWin / 8 + u] = p.look;
      const uint32_t p0 = (uint32_t)kb * kWin + 8u * (uint32_t)u;
      const uint64_t v64 = ((uint64_t)p.own.y << 32) | p.own.x;
      uint2 e[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t x = k < 7 ? (uint32_t)(v64 >> (8 * k)) & 0xFFFFu : ((p.own.y >> 24) | ((p.nxt.x & 0xFFu) << 8));
        const uint32_t two = (x >> 7) & 1u;
        const uint32_t header = (x & 0x7Fu) | (two ? ((x >> 1) & 0x3F80u) : 0u);
        const uint32_t hlen = 1u + two;
        const uint32_t isbp = header & 1u;
        const uint32_t cg = header >> 1;
        const uint32_t nv = cg << (isbp * 3u);
        const uint32_t adv = hlen + (isbp ? cg * (uint32_t)w : nb_rle);
        const uint32_t p = p0 + (uint32_t)k;
        const bool ok = (x & 0x8080u) != 0x8080u && nv != 0u && w > 0 && p < lim && p + adv <= qend;
        e[k] = make_uint2(ok ? adv : 0u, nv | (isbp << 31));
      }
      uint4 *dst = reinterpret_cast<uint4 *>(&sh.tab[b][8 * u]);
#pragma unroll
      for (int k = 0; k < 4; k++) dst[k] = make_uint4(e[2 * k].x, e[2 * k].y, e[2 * k + 1].x, e[2 * k + 1].y);
    } else if (kf >= 0) {
      const int b = kf & 1;
      const uint32_t wq = (uint32_t)kf * kWin;
      const uint32_t *win = sh.raw[b];
      while (q - wq < (uint32_t)kWin && q < qend && status == PQD_OK) {
        if (out < (uint64_t)ucnt) {
          // Hops through the table while values are still wanted: one LDS read, the position noted in the next
          // free lane, three additions.  Everything else about these runs is worked out later, 64 runs at a time.
          uint32_t o32 = (uint32_t)out;
          int np = nr - nr_flushed;
          const int np0 = np;
          if (dense) {
            // Dense stretch of one-byte-header RLE runs (all the reference's own writer emits,
            // encodingwrite.go:149-214): their headers sit at a fixed stride, so every free lane looks up the
            // position one stride further on and the leading lanes that all find such a run are noted at once.
            // (Kept apart from the hop loop below, whose few instructions per run are the kernel's critical path.)
            for (;;) {
              const uint32_t o = q - wq;
              if (o >= (uint32_t)kWin) break;
              const int i = lane - np;
              const uint32_t oi = o + (uint32_t)(i < 0 ? 0 : i) * stride;
              bool ok = i >= 0 && oi < (uint32_t)kWin;
              uint2 ti = make_uint2(0u, 0u);
              if (ok) ti = sh.tab[b][oi];
              ok = ok && ti.x == stride && (ti.y >> 31) == 0u;
              const uint64_t okm = __ballot(ok) >> np;
              int n_ok = okm == ~0ull ? 64 : (int)__builtin_ctzll(~okm);
              if (n_ok > 64 - np) n_ok = 64 - np;
              if (n_ok < 2) { dense = false; break; }
              const uint32_t nvi = (i >= 0 && i < n_ok) ? ti.y : 0u;
              const uint32_t incl = wave_incl_sum(nvi);
              const uint64_t full = __ballot(i >= 0 && i < n_ok && incl >= ucnt - o32);  // the run that completes the count
              const int n_take = full ? (int)__builtin_ctzll(full) - np + 1 : n_ok;
              if (i >= 0 && i < n_take) rq = wq + oi;
              o32 += (uint32_t)__builtin_amdgcn_readlane((int)incl, np + n_take - 1);
              np += n_take;
              q += (uint32_t)n_take * stride;
              if (n_take < 8) { dense = false; break; }
              if (np == 64 || o32 >= ucnt) break;
            }
          } else {
            for (;;) {
              const uint32_t o = q - wq;
              if (o >= (uint32_t)kWin) break;
              const uint2 te = sh.tab[b][o];
              const uint32_t adv = __builtin_amdgcn_readfirstlane(te.x);
              const uint32_t nvf = __builtin_amdgcn_readfirstlane(te.y);
              if (adv == 0u) break;
              rq = write_lane(rq, q, np);
              np++;
              o32 += nvf & 0x7FFFFFFFu;
              q += adv;
              if (np == 64 || o32 >= ucnt) break;
            }
          }
          if (np != np0) {
            nr = nr_flushed + np;
            out = o32;
            emit_q = q;
            if (np == 64) { finalize(wq, b); flush(); }
            continue;
          }
        }
        if (q >= q_avail_end) {
          // the rest of the stream is the zero fill of a short Read: empty RLE runs of 1 + nb_rle bytes; the walk ends
          // cleanly unless the last header sits on the last byte (encodingread.go:152-156: Read at the end -> io.EOF)
          if (nb_rle > 0 && (qend - q) % (1u + nb_rle) == 1u) status = PQD_ERR_EOF;
          q = qend;
          break;
        }
        // ---- one run the long way (encodingread.go:127-228) ----
        const uint32_t run_q = q;
        uint32_t di = (q - wq) >> 2;
        const uint32_t s8 = (q & 3u) * 8u;
        uint32_t d0 = __builtin_amdgcn_readfirstlane(win[di]);
        uint32_t d1 = __builtin_amdgcn_readfirstlane(win[di + 1]);
        uint32_t d2 = __builtin_amdgcn_readfirstlane(win[di + 2]);
        uint32_t x0 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> s8);
        uint32_t x1 = (uint32_t)((((uint64_t)d2 << 32) | d1) >> s8);
        // varint header.  Bytes past the end of the stream read as zero, which ends the varint with the partial
        // value the reference returns at EOF.
        uint32_t header, hlen;
        bool huge = false;  // a header of 2^31 or more: a run of at least 2^30 values
        if ((x0 & 0x80u) == 0) { header = x0 & 0x7Fu; hlen = 1; }
        else if ((x0 & 0x8000u) == 0) { header = (x0 & 0x7Fu) | ((x0 >> 1) & 0x3F80u); hlen = 2; }
        else {
          // longer header: byte by byte inside the look-ahead
          uint64_t h64 = 0;
          uint32_t n = 0, shift = 0;
          for (;;) {
            const uint32_t qq = q + n;
            if (qq >= qend || n >= 56) break;
            const uint32_t bb = (__builtin_amdgcn_readfirstlane(win[(qq - wq) >> 2]) >> ((qq & 3u) * 8u)) & 0xFFu;
            n++;
            if (shift < 64) h64 |= (uint64_t)(bb & 0x7Fu) << shift;
            if ((bb & 0x80u) == 0) break;
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
          const uint64_t v64 = (uint64_t)(uint32_t)((((uint64_t)d1 << 32) | d0) >> shv);
          // place the value bytes at byte offset hlen of (x1:x0) when hlen <= 4; longer headers keep them at offset 0
          if (n <= 4) { const uint64_t t = v64 << (8u * n); x0 = (uint32_t)t; x1 = (uint32_t)(t >> 32); }
          else { x0 = (uint32_t)v64; x1 = 0; hlen = 0; q += n; if (q > qend) q = qend; }
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
            const uint32_t have = qend - qh < nb_rle ? qend - qh : nb_rle;  // a short read is zero filled
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
          record_decoded(run_q, e_x | nv, e_y);
          emit_q = q;
          if (nr - nr_flushed == 64) { finalize(wq, b); flush(); }
        }
        out += nv;
        if (out > (1ull << 40)) out = 1ull << 40;
      }
      finalize(wq, b);  // the window is about to be replaced
      if (lane == 0) {
        sh.q = q;
        sh.done = (q >= qend || status != PQD_OK) ? 1 : 0;
      }
    }
    __syncthreads();
    // next window: the one after the one just built, or further on when the follower has jumped past it
    const uint32_t q_now = sh.q;
    all_done = sh.done != 0;
    const int k_q = (int)(q_now / (uint32_t)kWin);
    int kb_next = kb + 1;
    if (kf >= 0 && k_q > kb) kb_next = k_q;  // the window just built holds nothing the follower needs
    if (!follower && kb_next != kb + 1 && !all_done) {
      // a jump: the loads in flight are for the wrong windows
#pragma unroll
      for (int i = 0; i < kAhead; i++) pf[i] = pre_load(base16, (uint32_t)(kb_next + i) * kWin, q_avail_end, u);
    }
    // (after a jump the window kb_next - 1 is not in the buffers: the follower's loop condition q - wq < kWin is
    //  false for it, so in the coming iteration it only republishes q)
    kb = kb_next;
    __syncthreads();  // sh.q / sh.done are rewritten by the follower in the next iteration
  }

  if (!follower) return;
  // ---- end of the stream: last records, end sentinel, totals ----
  flush();
  if (lane == 0 && status != PQD_ERR_NOMEM) runs[nr] = RunEntry{base_out > 0x7FFFFFFFu ? 0x7FFFFFFFu : base_out, 0u, base_items, 0u};
  if (status == PQD_OK && out < (uint64_t)cnt) status = PQD_ERR_MALFORMED;  // buf[:cnt] past the decoded values
  if (lane == 0) {
    sp->n_runs = nr;
    sp->pad = (int32_t)(emit_q - a);  // end of the bytes the recorded runs occupy
    sp->decoded = (int64_t)out;
    if (status != PQD_OK) sp->status = status;
  }
}

hipError_t launch_hybrid_prescan_wide(const ChunkCtx &c, int s0, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  static_assert(sizeof(Shared) <= 80 * 1024, "two workgroups per CU");
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_hybrid_prescan_wide), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(Shared));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_hybrid_prescan_wide, dim3(n), dim3(kThreads), sizeof(Shared), stream, c, s0);
  return hipGetLastError();
}

}  // namespace pqd
