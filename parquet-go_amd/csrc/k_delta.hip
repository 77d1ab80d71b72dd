// DELTA_BINARY_PACKED on gfx950: one wavefront per page.
//
// Reference behaviour reproduced here: encoding/encodingread.go:270-326 (INT32) and :329-383 (INT64):
// header {block size, miniblocks per block, value count, zigzag first value}; per block a zigzag minimum
// delta, one width byte per miniblock and the bit-packed miniblocks; v[i] = v[i-1] + d + minDelta with
// two's-complement wrap; miniblocks after the one that reaches the value count are not read; the result
// length is the stream's own count.
//
// The wave stages the page through an 8 KiB LDS ring (two 4 KiB segments loaded with 128-bit accesses, the
// next segment prefetched into registers), parses each block header on wave-uniform scalars, unpacks the
// standard 128-value block (4 miniblocks of 32) with two values per lane and reconstructs the values with
// a DPP prefix scan; 16-byte stores cover 1 KiB of contiguous output per block (INT64).
#include <hip/hip_runtime.h>
#include "pqd_kernels.h"

namespace pqd {

constexpr int kSeg = 4096;            // ring segment bytes
constexpr int kRingDwords = 2 * kSeg / 4;

__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t x, int n) {
  switch (n) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
  }
}

// inclusive prefix sum over the 64 lanes, wrapping
__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t x, int lane) {
#pragma unroll
  for (int n = 1; n <= 8; n <<= 1) {
    uint32_t lo = dpp_row_shr((uint32_t)x, n), hi = dpp_row_shr((uint32_t)(x >> 32), n);
    x += ((uint64_t)hi << 32) | lo;
  }
  // row totals sit in lanes 15, 31, 47
  uint64_t t0 = ((uint64_t)__builtin_amdgcn_readlane((int)(x >> 32), 15) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)x, 15);
  uint64_t t1 = ((uint64_t)__builtin_amdgcn_readlane((int)(x >> 32), 31) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)x, 31);
  uint64_t t2 = ((uint64_t)__builtin_amdgcn_readlane((int)(x >> 32), 47) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)x, 47);
  const int row = lane >> 4;
  uint64_t add = row == 0 ? 0 : row == 1 ? t0 : row == 2 ? t0 + t1 : t0 + t1 + t2;
  return x + add;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, int lane) {
#pragma unroll
  for (int n = 1; n <= 8; n <<= 1) x += dpp_row_shr(x, n);
  uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 15);
  uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)x, 31);
  uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)x, 47);
  const int row = lane >> 4;
  uint32_t add = row == 0 ? 0 : row == 1 ? t0 : row == 2 ? t0 + t1 : t0 + t1 + t2;
  return x + add;
}

__device__ __forceinline__ uint64_t bcast_lane(uint64_t x, int src) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(x >> 32), src) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)x, src);
}
__device__ __forceinline__ uint32_t bcast_lane(uint32_t x, int src) {
  return (uint32_t)__builtin_amdgcn_readlane((int)x, src);
}

template <class T> struct DeltaTraits;
template <> struct DeltaTraits<uint64_t> {
  // encodingread.go:351, :362
  static __device__ __forceinline__ uint64_t zigzag(uint64_t z) { return (z >> 1) ^ (uint64_t)(-(int64_t)(z & 1)); }
};
template <> struct DeltaTraits<uint32_t> {
  // encodingread.go:293-294, :305-306: truncate to int32 first
  static __device__ __forceinline__ uint32_t zigzag(uint64_t z) {
    uint32_t t = (uint32_t)z;
    return (t >> 1) ^ (uint32_t)(-(int32_t)(t & 1));
  }
};

struct SegRegs { uint4 v[4]; };

__device__ __forceinline__ void seg_load(SegRegs &r, const uint8_t *base16, uint32_t seg, uint32_t q_end, int lane) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t qp = seg * (uint32_t)kSeg + 16u * (uint32_t)(lane + 64 * j);
    uint4 x = make_uint4(0, 0, 0, 0);
    if (qp < q_end) {
      x = *reinterpret_cast<const uint4 *>(base16 + qp);
      uint32_t left = q_end - qp;
      if (left < 16) {
        uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          int lb = (int)left - 4 * k;
          if (lb <= 0) w[k] = 0;
          else if (lb < 4) w[k] &= (1u << (8 * lb)) - 1u;
        }
        x = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    r.v[j] = x;
  }
}
__device__ __forceinline__ void seg_store(const SegRegs &r, uint4 *ring4, uint32_t seg, int lane) {
#pragma unroll
  for (int j = 0; j < 4; j++) ring4[(seg & 1u) * (kSeg / 16) + lane + 64 * j] = r.v[j];
}

// w (<= 64) bits starting `bit` bits after byte position q of the ring
__device__ __forceinline__ uint64_t ring_extract(const uint32_t *ring, uint32_t q, uint32_t bit, uint32_t w) {
  uint32_t bytepos = q + (bit >> 3);
  uint32_t sh = (bytepos & 3u) * 8u + (bit & 7u);
  uint32_t di = bytepos >> 2;
  uint64_t d0 = ring[di & (kRingDwords - 1)], d1 = ring[(di + 1) & (kRingDwords - 1)], d2 = ring[(di + 2) & (kRingDwords - 1)];
  uint64_t v = (d0 | (d1 << 32)) >> sh;
  if (sh) v |= d2 << (64 - sh);
  if (w < 64) v &= ((uint64_t)1 << w) - 1;
  return v;
}

// byte at stream position (q-space) straight from global memory, zero past the end
__device__ __forceinline__ uint32_t gbyte(const uint8_t *base16, uint32_t q, uint32_t q_end) {
  return q < q_end ? base16[q] : 0u;
}
__device__ __forceinline__ uint64_t gextract(const uint8_t *base16, uint32_t q, uint64_t bit, uint32_t w, uint32_t q_end) {
  uint64_t bytepos = (uint64_t)q + (bit >> 3);
  uint32_t sh = (uint32_t)bit & 7u;
  uint64_t lo = 0, hi = 0;
  for (int k = 0; k < 8; k++) if (bytepos + k < q_end) lo |= (uint64_t)base16[bytepos + k] << (8 * k);
  if (bytepos + 8 < q_end) hi = base16[bytepos + 8];
  uint64_t v = lo >> sh;
  if (sh) v |= hi << (64 - sh);
  if (w < 64) v &= ((uint64_t)1 << w) - 1;
  return v;
}

// T = uint32_t (INT32 pages) or uint64_t (INT64 pages, and the length streams of the byte-array encodings).
// MODE 0: values -> out[dense_base + i]
// encoding: only pages with this value encoding are decoded.  src_sel 0: the stream starts at the page's value
// section; 1: it starts where the previous delta stream of the page ended (PageState.aux0).
template <class T>
__global__ __launch_bounds__(64) void k_delta_decode(ChunkCtx c, T *values_out, int encoding, int src_sel, int narrow) {
  __shared__ uint4 ring4[2 * kSeg / 16];
  const uint32_t *ring = reinterpret_cast<const uint32_t *>(ring4);
  const int lane = threadIdx.x;
  const int d = blockIdx.x;
  const Stream *vs = &c.streams[2 * c.nd + d];
  const int p = vs->page;
  PageState *ps = &c.state[p];
  if (c.pages[p].encoding != encoding) return;
  const int64_t cnt = ps->cnt;
  if (ps->status != 0 || cnt <= 0) return;  // cnt == 0: ReadDataPageValues returns before reading
  if (ps->delta_fast == 1) return;          // decoded block-parallel
  if (narrow && lane == 0) c.chunk_status[kLayoutMismatch] = 1;     // (a page the block-parallel kernels left: nothing has checked its blocks)

  const int64_t src_off = src_sel ? ps->aux0 : ps->val_off;
  const uint8_t *abs0 = c.in + src_off;
  const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(abs0) & 15u);
  const uint8_t *base16 = abs0 - a;
  const uint32_t q_end = a + (uint32_t)(ps->val_off + ps->val_len - src_off);
  T *out = values_out + ps->dense_base;

  int status = PQD_OK;
  uint32_t q = a;

  // first two ring segments; the stream header is parsed from LDS
  uint32_t hi_seg = 0;  // segments [.., hi_seg) are resident (at most the last two)
  SegRegs pf;
  seg_load(pf, base16, 0, q_end, lane);
  seg_store(pf, ring4, 0, lane);
  hi_seg = 1;
  seg_load(pf, base16, 1, q_end, lane);
  __syncthreads();
  auto ring_byte = [&](uint32_t qq) -> uint32_t {
    return (__builtin_amdgcn_readfirstlane(ring[(qq >> 2) & (kRingDwords - 1)]) >> ((qq & 3u) * 8u)) & 0xFFu;
  };
  // ---- stream header: four varints (at most 40 bytes, inside segment 0) ----
  uint64_t hdr[4];
  bool overlong = false;      // a varint of more than ten bytes: the reference reads on and drops the bits beyond 64; such a stream is refused
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint64_t v = 0;
    uint32_t shift = 0;
    for (int n = 0; n < 10; n++) {
      if (q >= q_end) break;  // EOF: partial value, no error (encodingread.go:132)
      uint32_t b = ring_byte(q);
      q++;
      if (shift < 64) v |= (uint64_t)(b & 0x7Fu) << shift;
      if ((b & 0x80u) == 0) break;
      shift += 7;
      if (n == 9) overlong = true;
    }
    hdr[i] = v;
  }
  const uint64_t block_size = hdr[0], n_mb = hdr[1], n_values = hdr[2];
  const T first = DeltaTraits<T>::zigzag(hdr[3]);
  if (overlong) status = PQD_ERR_REFUSED;
  else if (n_mb == 0) status = PQD_ERR_MALFORMED;                  // division by zero
  else if (n_values > (1ull << 30)) status = PQD_ERR_MALFORMED;    // stands for the reference running out of memory
  else if (n_mb > 65535) status = PQD_ERR_MALFORMED;               // likewise (one uint64 per miniblock width)
  if (status != PQD_OK) {
    if (lane == 0) atomicCAS(&ps->status, 0, status);
    return;
  }
  const uint64_t vals_in_mb = block_size / n_mb;
  const uint64_t groups = vals_in_mb / 8;
  if (lane == 0) out[0] = first;
  T carry = first;
  uint64_t produced = 1;

  if (block_size == 128 && n_mb == 4) {
    // ---- standard geometry: 4 miniblocks of 32 values, staged through the LDS ring ----
    const uint32_t last_seg = (q_end + kSeg - 1) / kSeg;  // segments at or beyond hold nothing
    while (produced < n_values) {
      // residency: header (<= 14 bytes) + 4 miniblocks (<= 1024 bytes) + slack
      while (q + 1100 > hi_seg * (uint32_t)kSeg && hi_seg < last_seg + 1) {
        __syncthreads();
        seg_store(pf, ring4, hi_seg, lane);
        hi_seg++;
        seg_load(pf, base16, hi_seg, q_end, lane);
        __syncthreads();
      }
      // bytes at q, wave-uniform
      const uint32_t di = q >> 2, sh = (q & 3u) * 8u;
      uint32_t d0 = __builtin_amdgcn_readfirstlane(ring[di & (kRingDwords - 1)]);
      uint32_t d1 = __builtin_amdgcn_readfirstlane(ring[(di + 1) & (kRingDwords - 1)]);
      uint32_t d2 = __builtin_amdgcn_readfirstlane(ring[(di + 2) & (kRingDwords - 1)]);
      const uint64_t xlo = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32) | __builtin_amdgcn_alignbit(d1, d0, sh);
      const uint64_t remaining = n_values - produced;

      T min_delta;
      uint32_t wb, qp;
      uint32_t st0, st1, st2, st3;
      int mb_needed;
      // Common case, kept short on the scalar unit (one scalar unit serves all waves of a CU, and this loop runs
      // once per 128 values): a minimum delta of at most four varint bytes, a full block, every miniblock inside
      // the stream, no width above 64.
      const uint32_t x0 = (uint32_t)xlo;
      const uint32_t stop = ~x0 & 0x80808080u;
      bool hot = false;
      if (__builtin_expect(stop != 0u && remaining >= 128u, 1)) {
        const uint32_t hl = ((uint32_t)__builtin_ctz(stop) >> 3) + 1u;  // 1..4 varint bytes
        uint32_t md32 = (x0 & 0x7Fu) | ((x0 >> 1) & 0x3F80u) | ((x0 >> 2) & 0x1FC000u) | ((x0 >> 3) & 0xFE00000u);
        md32 &= 0xFFFFFFFFu >> (32u - 7u * hl);
        wb = (uint32_t)(xlo >> (8u * hl));
        st0 = q + hl + 4u;
        st1 = st0 + 4u * (wb & 0xFFu);
        st2 = st1 + 4u * ((wb >> 8) & 0xFFu);
        st3 = st2 + 4u * ((wb >> 16) & 0xFFu);
        qp = st3 + 4u * (wb >> 24);
        const uint32_t over64 = (((wb & 0x7F7F7F7Fu) + 0x3F3F3F3Fu) | wb) & 0x80808080u;  // some width byte >= 65
        if (__builtin_expect(qp <= q_end && over64 == 0u, 1)) {
          hot = true;
          min_delta = DeltaTraits<T>::zigzag((uint64_t)md32);
          mb_needed = 4;
        }
      }
      if (!hot) {
        uint32_t d3 = __builtin_amdgcn_readfirstlane(ring[(di + 3) & (kRingDwords - 1)]);
        uint32_t d4 = __builtin_amdgcn_readfirstlane(ring[(di + 4) & (kRingDwords - 1)]);
        const uint64_t xhi = ((uint64_t)__builtin_amdgcn_alignbit(d4, d3, sh) << 32) | __builtin_amdgcn_alignbit(d3, d2, sh);
        // minDelta varint (at most 10 bytes; a longer one - the reference reads on and drops the bits beyond 64 - is refused)
        uint64_t md = 0;
        uint32_t hl = 0;
        {
          uint32_t shift = 0;
          for (;;) {
            if (q + hl >= q_end || hl >= 11) break;
            uint32_t b = (uint32_t)((hl < 8 ? xlo >> (8u * hl) : xhi >> (8u * (hl - 8u))) & 0xFFu);
            hl++;
            if (shift < 64) md |= (uint64_t)(b & 0x7Fu) << shift;
            if ((b & 0x80u) == 0) break;
            shift += 7;
          }
        }
        if (hl >= 11) { status = PQD_ERR_REFUSED; break; }
        min_delta = DeltaTraits<T>::zigzag(md);
        qp = q + hl;
        // the four width bytes: ReadByte at the end is io.EOF (encodingread.go:309-312)
        if (qp + 4 > q_end) { status = PQD_ERR_EOF; break; }
        if (hl == 0) wb = (uint32_t)xlo;
        else if (hl < 8) wb = (uint32_t)((xlo >> (8u * hl)) | (xhi << (64u - 8u * hl)));
        else wb = (uint32_t)(xhi >> (8u * (hl - 8u)));
        qp += 4;
        mb_needed = remaining >= 128 ? 4 : (int)((remaining + 31) / 32);
        uint32_t start[4];
        uint32_t wmax = 0;
        bool eof = false;
        for (int m = 0; m < 4; m++) {
          uint32_t wm = (wb >> (8 * m)) & 0xFFu;
          start[m] = qp;
          if (m < mb_needed) {
            if (wm > wmax) wmax = wm;
            if (wm > 0) {
              if (qp >= q_end) { eof = true; break; }       // ReadBitPacked with nothing left
              uint32_t nbytes = 4u * wm;
              qp += nbytes < q_end - qp ? nbytes : q_end - qp;  // a short miniblock is zero filled
            }
          }
        }
        if (eof) { status = PQD_ERR_EOF; break; }
        if (wmax > 64) { status = PQD_ERR_BIT_WIDTH; break; }
        st0 = start[0]; st1 = start[1]; st2 = start[2]; st3 = start[3];
      }
      // unpack: lane -> miniblock lane/16, values 2*(lane%16) and +1
      const int m = lane >> 4;
      const uint32_t wm = (wb >> (8 * m)) & 0xFFu;
      const uint32_t j0 = (uint32_t)(lane & 15) * 2u;
      T t0 = 0, t1 = 0;
      if (m < mb_needed) {
        const uint32_t sm = m == 0 ? st0 : m == 1 ? st1 : m == 2 ? st2 : st3;
        uint64_t e0 = 0, e1 = 0;
        if (wm > 0) {
          e0 = ring_extract(ring, sm, j0 * wm, wm);
          e1 = ring_extract(ring, sm, (j0 + 1) * wm, wm);
        }
        t0 = (T)e0 + min_delta;
        t1 = (T)e1 + min_delta;
      }
      const T incl = wave_incl_scan((T)(t0 + t1), lane);
      const T v1 = carry + incl;
      const T v0 = v1 - t1;
      const uint64_t idx = produced + 2u * (uint32_t)lane;
      if (m < mb_needed) {
        if (idx + 1 < (uint64_t)cnt) {
          struct Pair { T a, b; } __attribute__((packed));
          Pair pr = {v0, v1};
          *reinterpret_cast<Pair *>(out + idx) = pr;
        } else if (idx < (uint64_t)cnt) {
          out[idx] = v0;
        }
      }
      carry = bcast_lane(v1, 16 * mb_needed - 1);
      produced += 32u * (uint32_t)mb_needed;
      q = qp;
    }
  } else {
    // ---- any other geometry: straight from memory, one miniblock at a time ----
    // Miniblocks of fewer than 8 values hold nothing (ReadBitPacked is asked for zero groups): the reference reads
    // block header after block header without ever gaining a value, until ReadByte meets the end of the stream.
    if (groups == 0 && n_values > 1) status = PQD_ERR_EOF;
    const uint64_t vpm = groups * 8;
    while (status == PQD_OK && produced < n_values) {
      uint64_t md = 0;
      uint32_t shift = 0;
      for (;;) {
        if (q >= q_end) break;
        uint32_t b = base16[q];
        q++;
        if (shift < 64) md |= (uint64_t)(b & 0x7Fu) << shift;
        if ((b & 0x80u) == 0) break;
        shift += 7;
      }
      const T min_delta = DeltaTraits<T>::zigzag(md);
      if ((uint64_t)q + n_mb > (uint64_t)q_end) { status = PQD_ERR_EOF; break; }
      // a miniblock of more than 2^30 values stands for the reference running out of memory in ReadBitPacked
      if (groups > (1ull << 27)) { status = PQD_ERR_MALFORMED; break; }
      const uint32_t wq = q;
      q += (uint32_t)n_mb;
      for (uint64_t m = 0; m < n_mb && produced < n_values; m++) {
        const uint32_t wm = base16[wq + m];
        if (wm > 64) { status = PQD_ERR_BIT_WIDTH; break; }
        uint32_t sm = q;
        if (wm > 0) {
          if (q >= q_end) { status = PQD_ERR_EOF; break; }
          uint64_t nbytes = groups * wm;
          q += (uint32_t)(nbytes < (uint64_t)(q_end - q) ? nbytes : (uint64_t)(q_end - q));
        }
        // (values beyond the stream's count are decoded by the reference and then dropped: only the bytes count)
        const uint64_t want = n_values - produced < vpm ? n_values - produced : vpm;
        for (uint64_t i0 = 0; i0 < want; i0 += 64) {
          const uint64_t i = i0 + lane;
          T t = 0;
          if (i < vpm) t = (T)(wm ? gextract(base16, sm, i * wm, wm, q_end) : 0) + min_delta;
          const T incl = wave_incl_scan(t, lane);
          const T v = carry + incl;
          const uint64_t idx = produced + i;
          if (i < vpm && idx < (uint64_t)cnt) out[idx] = v;
          const int last = (int)(vpm - i0 < 64 ? vpm - i0 : 64) - 1;
          carry = bcast_lane(v, last);
        }
        produced += vpm;
      }
    }
  }
  // the caller slices the decoded values to the page's count (layout/page.go:917 values[j]): too few is an index out
  // of range there, after the stream itself has been read without error
  if (status == PQD_OK && n_values < (uint64_t)cnt) status = PQD_ERR_MALFORMED;
  // Length and prefix streams of byte-array pages that announce more values than the page holds: the reference
  // builds (and may fail on) every one of those strings (encodingread.go:394-406, :425-433); the strings beyond
  // the page's count are not looked at here, so such a page is refused rather than decoded differently.
  if (status == PQD_OK && encoding != PQD_ENC_DELTA_BINARY_PACKED && n_values > (uint64_t)cnt) status = PQD_ERR_REFUSED;
  if (lane == 0) {
    if (status != PQD_OK) atomicCAS(&ps->status, 0, status);
    ps->aux0 = src_off + (int64_t)(q - a);  // where the delta stream ended (start of a byte-array payload)
    ps->aux1 = (int64_t)n_values;
  }
}

// ---------------------------------------------------------------------------------------------
// Block-parallel path for streams of the standard geometry (128-value blocks, 4 miniblocks) that are intact.
//
// k_delta_blocks (one wavefront per page) walks the block headers and leaves one DeltaBlock per block.  The chain -
// a block's length follows from its own widths - is followed 64 blocks at a time: every lane parses the header that
// sits where block j would start if all blocks were as long as the last one seen; the leading lanes whose
// predecessors all have that length are at true block starts (exact: the first block of another length ends the
// step, and its length is the next guess).  Streams of steady width - timestamps, counters, lengths - take one step
// per 64 blocks.  Anything unusual (another geometry, a header cut by the end of the stream, a miniblock that runs
// past it, a count that contradicts the page) leaves the page to k_delta_decode, which reproduces the reference's
// behaviour on such input; delta_fast tells the two apart.
//
// The blocks of a page are then cut into units of up to 64 (8192 values), one wavefront each (k_delta_units).  The only
// thing that ties a block to the ones before it is the running value.  A unit stages its packed bytes in LDS (one
// coalesced sweep; the loop that follows has no global loads, so its stores are never waited for), adds up the
// deltas of every block - one block per lane, fields peeled off a bit buffer - and publishes the unit's total; it
// then collects the totals of the units before it in the page (they were started no later and publish before they
// wait, so the wait is short; it is bounded all the same, and a page whose wait ran out is decoded again by
// k_delta_decode) and decodes two blocks per step with nothing carried from step to step: four consecutive values
// per lane from four LDS dwords, a lane-local running sum, one scan over the 32 lanes of a block, 16 or 32 bytes of
// output per lane.  The packed bytes are read from memory once.
//
// Work space: DeltaBlock[nb] | unit totals 2 x uint64[nu] | DeltaUnit[nu], with
// nb = slots / 128 + 3 * pages + 8 and nu = slots / 1024 + pages + 8 (room for the smallest units).  Per page the block table holds n_blocks
// entries, then {first value (min_delta field)} and {stream start (min_delta field)}.
// ---------------------------------------------------------------------------------------------
struct DeltaWork {
  DeltaBlock *blocks;
  uint64_t *gsum;       // two words per unit: the unit's total, published by the unit itself
  DeltaUnit *units;
  int64_t n_units;
  int32_t ushift;       // a unit is 2^ushift blocks (3 .. 6), chosen per call from the bytes the pages spend per value
  int32_t win_bytes;    // LDS window over the packed bytes of a unit (the launch's dynamic LDS)
  int32_t spec;         // pages whose blocks all have one length skip the walk (0: every page is walked)
  int32_t narrow;       // the values are lengths of strings, kept as 32-bit words (see delta_narrow_ok)
};

// Lengths of strings are decoded with 32-bit arithmetic and kept as 32-bit words where that is provably the same thing: when
// the first value and every step from one value to the next are below 2^31 in magnitude (a minimum delta below 2^30, packed
// deltas of at most 30 bits) and all the 32-bit results lie in the range of a string's length - which their consumers
// check value by value -, the 64-bit results the reference computes are those very numbers.  A block that does not keep to
// that sets chunk_status[kLayoutMismatch], and so does a page left to k_delta_decode: the call is then decoded again with
// 64-bit words by the kernels that decide what such a page is.
__device__ __forceinline__ bool delta_narrow_ok(uint64_t md_zigzag, uint32_t widths) {
  return md_zigzag < (1ull << 31) && ((widths | (widths + 0x61616161u)) & 0x80808080u) == 0u;      // every width <= 30
}

// A unit is 2^ushift blocks, as many as fit the LDS window (at most 64: one block per lane).  What a unit costs beyond
// its values - its record, its table entries, the wait for its bytes, the totals of the units before it - is the same
// for a small unit and a large one, so units are as large as the window allows (512 Mi INT64 values of 12-bit deltas:
// 1.92 ms with 16 blocks per unit, 1.53 with 32, 1.32 with 64), and the window no larger than the unit needs (LDS per
// wavefront sets how many are in flight per CU: 1.38 ms with a window a quarter larger).  A unit whose bytes outgrow
// the window is decoded through a window that moves over its blocks (half again as slow).
constexpr int kMaxUnitShift = 6, kMinUnitShift = 3;
constexpr uint32_t kMaxUnitBlocks = 1u << kMaxUnitShift;
constexpr int kMaxWinBytes = 13312;
constexpr int kMinWinBytes = 2560;      // two blocks of the widest kind and the window's alignment

__device__ __forceinline__ int64_t delta_table_base(const PageIn &pi) { return (pi.slot_base >> 7) + 3 * (int64_t)pi.data_idx; }
__device__ __forceinline__ int64_t delta_unit_base(const PageIn &pi, int ushift) { return (pi.slot_base >> (7 + ushift)) + (int64_t)pi.data_idx; }

// One block header out of the 16 bytes that start at it (x[0..3], little endian): the minimum delta (a varint of at most ten
// bytes), the four miniblock widths, and from them where the block's data starts and how long the block is.  pj: position
// of the header in the stream, end: length of the stream, blk: index of the block.  good: the header and the miniblocks the
// stream still needs at this block (encodingread.go:372: reading stops with the count) lie inside the stream and no width
// exceeds 64.
struct DeltaHdr {
  bool good;
  uint32_t size, data_off, widths, dbytes;
  uint64_t md;
};
__device__ __forceinline__ DeltaHdr delta_parse_header(const uint32_t x[4], uint32_t pj, uint32_t end, uint32_t blk, uint64_t n_values) {
  DeltaHdr h{false, 0u, 0u, 0u, 0u, 0ull};
  uint32_t hl = 0, shift = 0;
  bool closed = false;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    if (!closed) {
      const uint32_t b = (x[i >> 2] >> (8 * (i & 3))) & 0xFFu;
      hl++;
      if (shift < 64) h.md |= (uint64_t)(b & 0x7Fu) << shift;
      shift += 7;
      closed = (b & 0x80u) == 0;
    }
  }
  if (closed && (uint64_t)pj + hl + 4 <= end) {
    const uint64_t x64lo = ((uint64_t)x[1] << 32) | x[0], x64hi = ((uint64_t)x[3] << 32) | x[2];
    const uint32_t wb = hl < 8 ? (uint32_t)((x64lo >> (8u * hl)) | (hl ? x64hi << (64u - 8u * hl) : 0ull)) : (uint32_t)(x64hi >> (8u * (hl - 8u)));
    const uint64_t remaining = n_values - 1 - (uint64_t)blk * 128u;
    const int need = remaining >= 128 ? 4 : (int)((remaining + 31) / 32);
    uint32_t bytes = 0, wmax = 0;
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const uint32_t wm = (wb >> (8 * m)) & 0xFFu;
      if (m < need) { bytes += 4u * wm; h.widths |= wm << (8 * m); wmax = wm > wmax ? wm : wmax; }
    }
    h.data_off = pj + hl + 4u;
    h.dbytes = bytes;
    h.size = hl + 4u + bytes;
    h.good = wmax <= 64u && (uint64_t)h.data_off + bytes <= end;
  }
  return h;
}

// What a page whose blocks all have one length needs instead of a walk (k_delta_units then reads the headers itself).
struct DeltaSpec {
  uint32_t L;         // the length of every block but the last; 0: the page was walked
  uint32_t q0;        // position of the first block header
  uint32_t end_pos;   // where the last block ends
  uint32_t slen;      // length of the stream
  uint64_t n_values;
};

constexpr int kWalkWinVec = 1024 + 2;     // 16-byte vectors of the walk's LDS window (16 KiB and the reach of a header)

// -> the number of blocks, or -1 when the page is left to k_delta_decode
__device__ __forceinline__ int64_t delta_walk(const ChunkCtx &c, const PageIn &pi, PageState *ps, int src_sel, DeltaBlock *table, DeltaUnit *pu, int ushift, int lane,
                                              int64_t *src_off_out, uint64_t *first_out, int spec_win, DeltaSpec *spec, uint4 *s_win, bool narrow) {
  spec->L = 0;
  if (lane == 0) ps->delta_fast = 0;
  const int64_t cnt = ps->cnt;
  if (ps->status != 0 || cnt <= 0) return -1;
  const int64_t src_off = src_sel ? ps->aux0 : ps->val_off;
  const int64_t slen = ps->val_off + ps->val_len - src_off;
  if (slen <= 0 || slen > 0x7FFFFF00ll) return -1;
  const uint8_t *src = c.in + src_off;
  const uint32_t end = (uint32_t)slen;
  // last dword that may be read: the one holding the stream's last byte (a call may span several chunk buffers; what
  // lies between them is not the caller's to give)
  const uint8_t *in_last = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(src + end - 1u) & ~(uintptr_t)3);

  // A window of the stream in LDS, from the 16-byte boundary at or below position p (never below the chunk buffer, which is
  // aligned): all its vectors are asked for before the first is waited for - one trip to memory per window.  Vectors that
  // reach past the stream's last dword are not read: that dword's vector comes in dword by dword, the rest is zero.
  uint32_t *win32 = reinterpret_cast<uint32_t *>(s_win);
  uint32_t wpos = 0, mis = 0;
  int wvec = 0;          // vectors of the window that are in place (the first window is a short one: a page whose blocks turn out one
                         // length long needs the stream's header and its first block header, nothing else)
  auto stage = [&](uint32_t p, int n_vec) {
    wvec = n_vec;
    mis = (uint32_t)(reinterpret_cast<uintptr_t>(src + p) & 15u);
    wpos = p;
    const uint8_t *wsrc = src + p - mis;
    constexpr int kPer = (kWalkWinVec + 63) / 64;
    uint4 v[kPer];
    __syncthreads();       // (one wavefront: the reads of the window before are done)
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      if (k > 0 && n_vec <= 64) break;       // (the short first window: one vector per lane)
      const uint8_t *a = wsrc + 16 * (lane + 64 * k);
      const bool ok = lane + 64 * k < n_vec && a + 12 <= in_last;
      v[k] = *reinterpret_cast<const uint4 *>(ok ? a : c.in);     // (an unconditional load: with branches around the loads each one is waited for)
    }
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const uint8_t *a = wsrc + 16 * (lane + 64 * k);
      const bool ok = a + 12 <= in_last;
      if (k > 0 && n_vec <= 64) break;
      if (lane + 64 * k < n_vec) s_win[lane + 64 * k] = ok ? v[k] : make_uint4(0u, 0u, 0u, 0u);
    }
    const uint8_t *tail = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(in_last) & ~(uintptr_t)15);   // the vector of the last dword
    if (tail >= wsrc && tail < wsrc + 16 * n_vec && tail + 12 > in_last && lane < 4 && tail + 4 * lane <= in_last)
      win32[(tail - wsrc) / 4 + lane] = *reinterpret_cast<const uint32_t *>(tail + 4 * lane);
    __syncthreads();
  };
  auto win_byte = [&](uint32_t p) -> uint32_t {      // the byte at stream position p (inside the window)
    const uint32_t r = p - wpos + mis;
    return (win32[r >> 2] >> (8u * (r & 3u))) & 0xFFu;
  };
  stage(0, 64);

  // stream header: four varints, parsed by every lane alike
  uint32_t q = 0;
  uint64_t hdr[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint64_t v = 0;
    uint32_t shift = 0;
    for (int n = 0; n < 10; n++) {
      if (q >= end) return -1;                   // a header cut short: the general path
      const uint32_t b = win_byte(q++);
      if (shift < 64) v |= (uint64_t)(b & 0x7Fu) << shift;
      if ((b & 0x80u) == 0) break;
      shift += 7;
      if (n == 9) return -1;
    }
    hdr[i] = v;
  }
  if (hdr[0] != 128 || hdr[1] != 4) return -1;
  if (narrow && hdr[3] >= (1ull << 31) && lane == 0) c.chunk_status[kLayoutMismatch] = 1;
  const uint64_t n_values = hdr[2];
  const int64_t nv_page = pi.num_values > 0 ? pi.num_values : 0;
  if (n_values < (uint64_t)cnt || n_values > (uint64_t)nv_page) return -1;       // (short: an error; long: no room in the table)
  if (pi.encoding != PQD_ENC_DELTA_BINARY_PACKED && n_values != (uint64_t)cnt) return -1;
  const uint32_t n_blocks = (uint32_t)((n_values - 1 + 127) / 128);
  DeltaBlock *tb = table + delta_table_base(pi);

  auto header_at = [&](uint64_t pj, uint32_t x[4]) {     // the 16 bytes at stream position pj (pj + 5 <= end)
    const uint8_t *a0 = src + (uint32_t)pj;
    const uint32_t al = (uint32_t)(reinterpret_cast<uintptr_t>(a0) & 3u);
    const uint32_t *w0 = reinterpret_cast<const uint32_t *>(a0 - al);
    const uint32_t *wl = reinterpret_cast<const uint32_t *>(in_last);
    uint32_t dw[5];
#pragma unroll
    for (int i = 0; i < 5; i++) dw[i] = *(w0 + i <= wl ? w0 + i : wl);
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = __builtin_amdgcn_alignbit(dw[i + 1], dw[i], al * 8u);
  };

  // Streams of steady width - timestamps, counters - consist of blocks of one length L.  Then block k starts at q + k L and
  // nothing has to be walked: the units read their own headers out of the bytes they stage anyway, check that every block
  // is L long and hand the verdict down the page with their totals (k_delta_units); a page where that fails is decoded
  // again by k_delta_decode.  Tried here on 64 blocks spread over the page (the last one among them, whose length may
  // differ); anything else is walked as before.
  if (spec_win > 0 && n_blocks >= 2u && (uint64_t)q + 5 <= end) {
    uint32_t x[4];
    {     // (the first block header lies in the window)
      const uint32_t r = q - wpos + mis, al = (r & 3u) * 8u;
      uint32_t dw[5];
#pragma unroll
      for (int i = 0; i < 5; i++) dw[i] = win32[(r >> 2) + i];
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = __builtin_amdgcn_alignbit(dw[i + 1], dw[i], al);
    }
    const DeltaHdr h0 = delta_parse_header(x, q, end, 0u, n_values);
    const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane((int)h0.size);
    const bool fits = h0.good && L != 0u && ((uint64_t)L << ushift) + 96u <= (uint64_t)spec_win;
    if (__builtin_amdgcn_readfirstlane((int)fits)) {
      const uint32_t k = (uint32_t)(((uint64_t)(n_blocks - 1u) * (uint32_t)lane) / 63u);
      const uint64_t pj = (uint64_t)q + (uint64_t)k * L;
      bool ok = pj + 5 <= end;
      DeltaHdr hk{false, 0u, 0u, 0u, 0u, 0ull};
      if (ok) {
        header_at(pj, x);
        hk = delta_parse_header(x, (uint32_t)pj, end, k, n_values);
        ok = hk.good && (hk.size == L || k == n_blocks - 1u);
      }
      if (__ballot(ok) == ~0ull) {
        const uint32_t size_last = (uint32_t)__builtin_amdgcn_readlane((int)hk.size, 63);
        const uint32_t end_pos = q + (n_blocks - 1u) * L + size_last;      // (<= end: the last block is good)
        // (the last unit holds the last block, which may be longer than the others: it must fit the window too)
        const uint32_t n_last = n_blocks - (((n_blocks - 1u) >> ushift) << ushift);
        if ((uint64_t)(n_last - 1u) * L + size_last + 96u > (uint64_t)spec_win) goto walk;
        if (lane == 0) {
          tb[n_blocks] = DeltaBlock{0u, 0u, hdr[3]};
          tb[n_blocks + 1] = DeltaBlock{0u, 0u, (uint64_t)src_off};
          ps->aux0 = src_off + (int64_t)end_pos;
          ps->aux1 = (int64_t)n_values;
          ps->delta_fast = 1;
        }
        spec->L = L; spec->q0 = q; spec->end_pos = end_pos; spec->slen = end; spec->n_values = n_values;
        *src_off_out = src_off;
        *first_out = hdr[3];
        return (int64_t)n_blocks;
      }
    }
  }

walk:
  // The chain is followed inside a window of the stream kept in LDS: a step costs a few LDS reads and the parse, not a
  // round trip to memory, so a stream whose blocks keep changing length (every other block, for lengths of strings) moves
  // at the pace of the arithmetic.  The window is loaded whenever the chain leaves it.
  uint32_t pos = q, guess = 0;
  uint32_t b0 = 0;
  while (b0 < n_blocks) {
    if (wvec < kWalkWinVec || (uint64_t)pos - wpos + mis + 20 > (uint64_t)kWalkWinVec * 16u) stage(pos, kWalkWinVec);
    for (;;) {
      const uint32_t blk = b0 + (uint32_t)lane;
      const uint64_t pj64 = (uint64_t)pos + (uint64_t)(uint32_t)lane * guess;
      const uint64_t rel = pj64 - wpos + mis;                    // offset of the header in the window
      const bool inwin = rel + 20 <= (uint64_t)kWalkWinVec * 16u;
      const bool mine = inwin && blk < n_blocks && (lane == 0 || guess != 0) && pj64 + 5 <= end;
      uint32_t size = 0, data_off = 0, widths = 0, dbytes = 0;
      uint64_t md = 0;
      bool good = false;
      uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;
      if (mine) {
        const uint32_t di = (uint32_t)rel >> 2, al = ((uint32_t)rel & 3u) * 8u;
        uint32_t dw[5], x[4];
#pragma unroll
        for (int i = 0; i < 5; i++) dw[i] = win32[di + i];
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = __builtin_amdgcn_alignbit(dw[i + 1], dw[i], al);
        // The common header in a few instructions: a minimum delta of at most three bytes, a full block (all four
        // miniblocks needed), the block inside the stream.  Anything else is parsed the long way.
        const uint32_t c0 = x[0] & 0x80u, c1 = x[0] & 0x8000u, c2 = x[0] & 0x800000u;
        const uint32_t hl = c0 == 0u ? 1u : (c1 == 0u ? 2u : 3u);
        const uint32_t wb = __builtin_amdgcn_alignbit(x[1], x[0], 8u * hl);          // the four widths
        const uint32_t sumw = __builtin_amdgcn_sad_u8(wb, 0u, 0u);
        const bool small_w = ((wb | (wb + 0x3F3F3F3Fu)) & 0x80808080u) == 0u;       // every width <= 64
        const uint64_t remaining = n_values - 1 - (uint64_t)blk * 128u;
        if ((c0 == 0u || c1 == 0u || c2 == 0u) && small_w && remaining >= 128u && pj64 + hl + 4u + 4u * sumw <= end) {
          md = (x[0] & 0x7Fu) | (hl > 1u ? ((x[0] >> 1) & 0x3F80u) : 0u) | (hl > 2u ? ((x[0] >> 2) & 0x1FC000u) : 0u);
          widths = wb; dbytes = 4u * sumw; data_off = (uint32_t)pj64 + hl + 4u; size = hl + 4u + dbytes; good = true;
        }
        x0 = x[0]; x1 = x[1]; x2 = x[2]; x3 = x[3];
      }
      // lane j sits at a true block start when every lane before it is good and as long as the guess
      uint64_t okm = __ballot(good && size == guess);
      // The long way for the one lane that needs it: the first lane that does not continue the chain sits at a true start
      // too, and when the short parse did not take its header (the stream's last block, a long minimum delta, a width
      // above 64) it is parsed in full.  The lanes behind it sit wherever the guess put them and mostly look at packed
      // bytes: what they hold is never used, and parsing them in full every step is what made a step slow.
      {
        const int nxt = guess == 0u ? 0 : (okm == ~0ull ? 64 : (int)__builtin_ctzll(~okm));
        if (lane == nxt && mine && !good) {
          const uint32_t x[4] = {x0, x1, x2, x3};
          const DeltaHdr h = delta_parse_header(x, (uint32_t)pj64, end, blk, n_values);
          size = h.size; data_off = h.data_off; widths = h.widths; dbytes = h.dbytes; md = h.md; good = h.good;
        }
        okm = __ballot(good && size == guess);
      }
      const uint64_t goodm = __ballot(good);
      const uint64_t inm = __ballot(inwin);
      const int n_chain = okm == ~0ull ? 64 : (int)__builtin_ctzll(~okm);   // lanes [0, n_chain) continue the chain at the guessed length
      // the lane after them is at a true start too (its own length may differ)
      int n_true = n_chain < 64 ? n_chain + 1 : 64;
      if (guess == 0) n_true = 1;
      if (b0 + (uint32_t)n_true > n_blocks) n_true = (int)(n_blocks - b0);
      // ... as far as the window reaches (its lanes are the leading ones: positions grow with the lane)
      const int n_in = inm == ~0ull ? 64 : (int)__builtin_ctzll(~inm);
      if (n_true > n_in) n_true = n_in;
      if (n_true == 0) break;                      // (the next header lies at the window's end: load again from there)
      const uint64_t want = n_true >= 64 ? ~0ull : ((1ull << n_true) - 1);
      if ((goodm & want) != want) return -1;       // a block the fast path does not take
      if (lane < n_true) {
        if (narrow && !delta_narrow_ok(md, widths)) c.chunk_status[kLayoutMismatch] = 1;
        tb[blk] = DeltaBlock{data_off, widths, md};
        // the byte range of the unit the block belongs to
        const uint32_t ublocks = 1u << ushift;
        if ((blk & (ublocks - 1u)) == 0u) pu[blk >> ushift].win_start = data_off;
        if ((blk & (ublocks - 1u)) == ublocks - 1u || blk + 1u == n_blocks) pu[blk >> ushift].win_end = data_off + dbytes;
      }
      const int last = n_true - 1;
      const uint32_t pos_last = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)pj64), last);
      const uint32_t size_last = (uint32_t)__builtin_amdgcn_readlane((int)size, last);
      pos = pos_last + size_last;
      guess = size_last;
      b0 += (uint32_t)n_true;
      if (b0 >= n_blocks) break;
      if ((uint64_t)pos - wpos + mis + 20 > (uint64_t)kWalkWinVec * 16u) break;
    }
  }
  if (lane == 0) {
    tb[n_blocks] = DeltaBlock{0u, 0u, hdr[3]};
    tb[n_blocks + 1] = DeltaBlock{0u, 0u, (uint64_t)src_off};
    ps->aux0 = src_off + (int64_t)pos;   // where the delta stream ended (start of a byte-array payload)
    ps->aux1 = (int64_t)n_values;
    ps->delta_fast = 1;
  }
  *src_off_out = src_off;
  *first_out = hdr[3];
  return (int64_t)n_blocks;
}

__global__ __launch_bounds__(64) void k_delta_blocks(ChunkCtx c, int encoding, int src_sel, DeltaWork w) {
  __shared__ uint4 s_win[kWalkWinVec];
  const int lane = threadIdx.x;
  const int d = blockIdx.x;
  const int p = c.streams[2 * c.nd + d].page;
  const PageIn &pi = c.pages[p];
  const int64_t nvp = pi.num_values > 0 ? pi.num_values : 0;
  const int64_t ub = delta_unit_base(pi, w.ushift);
  const int64_t range = ((pi.slot_base + nvp) >> (7 + w.ushift)) - (pi.slot_base >> (7 + w.ushift)) + 1;
  int64_t src_off = 0;
  uint64_t first_raw = 0;
  DeltaSpec spec;
  spec.L = 0;
  const int64_t nb = pi.encoding == encoding ? delta_walk(c, pi, &c.state[p], src_sel, w.blocks, w.units + ub, w.ushift, lane, &src_off, &first_raw,
                                                          w.spec ? w.win_bytes : 0, &spec, s_win, w.narrow != 0) : -1;
  // the page's units: the ones its blocks fill get their records, the rest of its range holds nothing
  const int64_t ng = nb < 0 ? 0 : (nb == 0 ? 1 : (nb + ((int64_t)1 << w.ushift) - 1) >> w.ushift);     // (a stream of one value still has its first value to write)
  for (int64_t j = lane; j < range; j += 64) {
    if (ub + j >= w.n_units) continue;
    DeltaUnit *r = w.units + ub + j;
    if (j < ng) {
      r->g = (uint32_t)j; r->n_blocks = (uint32_t)nb; r->spare = spec.L;
      r->tbase = delta_table_base(pi); r->src_off = src_off; r->out_base = c.state[p].dense_base; r->cnt = c.state[p].cnt;
      if (spec.L != 0u) {
        // blocks of one length: the unit's bytes follow from its index; in place of the table's base the record carries
        // the stream's length and value count, which the unit needs to read the headers as the walk would
        const uint64_t b_lo = (uint64_t)j << w.ushift, b_hi = ((uint64_t)j + 1) << w.ushift;
        r->win_start = spec.q0 + (uint32_t)b_lo * spec.L;
        r->win_end = b_hi >= (uint64_t)nb ? spec.end_pos : spec.q0 + (uint32_t)b_hi * spec.L;
        r->tbase = (int64_t)((uint64_t)spec.slen | (spec.n_values << 32));
      }
      r->first_raw = first_raw;
      r->page = p;
    } else {
      r->page = -1;
    }
    w.gsum[2 * (ub + j)] = 0;
    w.gsum[2 * (ub + j) + 1] = 0;
  }
}

__device__ __forceinline__ int64_t uniform_i64(int64_t v) {  // a value every lane holds alike, moved to scalar registers
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

constexpr int kWinSlack = 64;           // bytes past the window that a lane's last reads may touch (never used)
constexpr uint32_t kSpinLimit = 1u << 22;

// Totals are published as self-validating 64-bit words (bit 32 set, 32 bits of the total below it), written and
// polled with device-scope atomics: no fence is needed and a word is either absent or whole.
__device__ __forceinline__ void publish_word(uint64_t *slot, uint32_t v) {
  __hip_atomic_store(slot, (1ull << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// both words of a unit (one for 32-bit values), asked for back to back
template <bool TWO>
__device__ __forceinline__ void poll_raw(const uint64_t *slot, uint64_t *a, uint64_t *b) {
  *a = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *b = TWO ? __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1ull << 32);
}
__device__ __forceinline__ bool poll_word(const uint64_t *slot, uint32_t *v) {
  const uint64_t x = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *v = (uint32_t)x;
  return (x >> 32) != 0;
}

// inclusive prefix sum over each half of the wavefront (the 32 lanes of one block), wrapping
__device__ __forceinline__ uint32_t half_incl_scan(uint32_t x) {
#pragma unroll
  for (int n = 1; n <= 8; n <<= 1) x += dpp_row_shr(x, n);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // lane 15 of a row to the row after it, rows 1 and 3
  return x;
}
__device__ __forceinline__ uint64_t half_incl_scan(uint64_t x) {
#pragma unroll
  for (int n = 1; n <= 8; n <<= 1) {
    const uint32_t lo = dpp_row_shr((uint32_t)x, n), hi = dpp_row_shr((uint32_t)(x >> 32), n);
    x += ((uint64_t)hi << 32) | lo;
  }
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, 0x142, 0xA, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), 0x142, 0xA, 0xF, false);
  return x + (((uint64_t)hi << 32) | lo);
}

// Sum of the 128 packed deltas of a full block (without the minimum delta), by one lane, from the LDS window.
// r: byte offset of the block's first miniblock in the window.  A miniblock of width w is exactly w dwords.
template <class T>
__device__ __forceinline__ T delta_block_sum(const uint32_t *win, uint32_t r, uint32_t widths) {
  T sum = 0;
  const uint32_t al = (r & 3u) * 8u;
  uint32_t di = r >> 2;
  uint32_t cur = win[di];
#pragma unroll 1
  for (int m = 0; m < 4; m++) {
    const uint32_t w = (widths >> (8 * m)) & 0xFFu;
    if (w == 0) continue;
    const uint64_t mask = w >= 64u ? ~0ull : ((1ull << w) - 1ull);
    // a bit buffer of up to 127 bits in (hi:lo); one dword in, values out while enough bits are there
    uint64_t lo = 0, hi = 0;
    uint32_t nbits = 0;
#pragma unroll 1
    for (uint32_t j = 0; j < w; j++) {
      const uint32_t nxt = win[di + 1];
      const uint64_t dw = __builtin_amdgcn_alignbit(nxt, cur, al);
      cur = nxt;
      di++;
      if (nbits < 64u) { lo |= dw << nbits; if (nbits > 32u) hi |= dw >> (64u - nbits); }
      else hi |= dw << (nbits - 64u);
      nbits += 32u;
      while (nbits >= w) {
        sum += (T)(lo & mask);
        if (w >= 64u) { lo = hi; hi = 0; }
        else { lo = (lo >> w) | (hi << (64u - w)); hi >>= w; }
        nbits -= w;
      }
    }
  }
  return sum;
}

// One wavefront per unit of 32 blocks: stage the packed bytes in LDS, add up every block's deltas, publish the unit's
// total, collect the totals of the units before it in the page, decode two blocks per step.
// Several wavefronts share a unit: its window, its table entries and its block sums sit in LDS once, and each wavefront
// takes its share of the steps of both sweeps.  What bounds the kernel is how many wavefronts the LDS windows leave room
// for (a unit's first trip to memory, the wait for the totals in front of it and its stores are covered by other
// wavefronts only), and this way a window keeps several in flight.  The record, the entries and the scan of the block sums
// are done by all alike (they read the same words); the first wavefront alone publishes the unit's total and looks back
// for the totals in front of it (with every wavefront polling, two per unit were no gain for INT64 and 10 % for INT32).
// Measured on 512 Mi values of w = 12, kernels (wavefronts per unit 1 / 2 / 4 / 8): INT32 0.78 / 0.67 / 0.61 / 0.82 ms,
// INT64 1.16 / 1.12 / 1.05 ms.
template <class T, int kUnitThreads>
__global__ __launch_bounds__(kUnitThreads) void k_delta_units(ChunkCtx c, T *__restrict__ values_out, DeltaWork w) {
  extern __shared__ uint4 win4[];           // w.win_bytes + kWinSlack
  __shared__ DeltaBlock s_ent[kMaxUnitBlocks];
  __shared__ uint64_t s_prev[kMaxUnitBlocks];
  __shared__ uint32_t s_tot[kMaxUnitBlocks];
  __shared__ int s_late, s_fail;
  const uint32_t kUnitBlocks = 1u << w.ushift;
  const uint32_t kWinBytes = (uint32_t)w.win_bytes;
  __shared__ uint4 s_stage_all[sizeof(T) == 8 ? 2 * kUnitThreads : 1];
  uint32_t *win = reinterpret_cast<uint32_t *>(win4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = kUnitThreads > 64 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  uint4 *s_stage = s_stage_all + (sizeof(T) == 8 ? 128 * wv : 0);
  const int64_t u = blockIdx.x;
  // the unit's record: sixteen dwords, one load
  const uint32_t rv = reinterpret_cast<const uint32_t *>(w.units + u)[lane & 15];
  auto r32 = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)rv, i); };
  auto r64 = [&](int i) { return (int64_t)(((uint64_t)r32(i + 1) << 32) | r32(i)); };
  const int p = (int)r32(0);
  if (p < 0) return;
  const uint32_t g = r32(1), n_blocks = r32(2), win_start = r32(3), win_end = r32(4), spec_len = r32(5);
  const int64_t tbase = r64(6), cnt = r64(12);
  const DeltaBlock *tb = w.blocks + (spec_len ? 0 : tbase);
  const T first = DeltaTraits<T>::zigzag((uint64_t)r64(14));
  const uint8_t *src = c.in + r64(8);
  const int64_t lim = (int64_t)((c.in + c.in_len) - src);          // bytes of the buffer from src on
  const uint32_t src_al = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u);
  T *out = values_out + r64(10);
  if (g == 0 && tid == 0) out[0] = first;
  const uint32_t b0 = kUnitBlocks * g;
  const uint32_t nb_here = b0 >= n_blocks ? 0u : (n_blocks - b0 < kUnitBlocks ? n_blocks - b0 : kUnitBlocks);

  // one block per lane (of the first 32)
  DeltaBlock E{0u, 0u, 0ull};
  if (wv == 0 && (uint32_t)lane < nb_here && spec_len == 0u) E = tb[b0 + (uint32_t)lane];      // (the first wavefront reads or works out the entries; the others take them from LDS)
  // The unit's bytes, when the window holds them all, are asked for at once with the table entries (the record
  // names the range); otherwise the window moves over the blocks.
  uint32_t win_a0 = win_start - ((src_al + win_start) & 15u);
  const bool one_window = nb_here != 0u && win_end - win_a0 <= kWinBytes;
  auto fill_window = [&](uint32_t a0, uint32_t end) {
    // (a0 may lie up to 15 bytes in front of the stream: signed from here on)  All loads first, then the LDS writes.
    constexpr int kSweeps = (kMaxWinBytes + 16 * kUnitThreads - 1) / (16 * kUnitThreads);   // (sweeps past `end` load and write nothing)
    const int64_t o0 = (int64_t)(int32_t)a0 + 16 * tid;
    uint4 x[kSweeps];
#pragma unroll
    for (int i = 0; i < kSweeps; i++) {
      const int64_t o = o0 + 16 * kUnitThreads * i;
      x[i] = make_uint4(0u, 0u, 0u, 0u);
      if (o < (int64_t)end && o < lim) x[i] = *reinterpret_cast<const uint4 *>(src + o);   // (16-byte aligned; its granule holds bytes of the buffer)
    }
#pragma unroll
    for (int i = 0; i < kSweeps; i++)
      if (o0 + 16 * kUnitThreads * i < (int64_t)end) win4[(uint32_t)(16 * tid + 16 * kUnitThreads * i) >> 4] = x[i];
  };
  if (one_window) fill_window(win_a0, win_end);
  if (spec_len != 0u) {
    // A page taken to consist of blocks of one length (k_delta_blocks): no table; block j of the unit starts spec_len bytes
    // after block j - 1, and every lane reads its block's header out of the window.  That holds for this unit when each of
    // its blocks (the page's last one aside) turns out spec_len long, and for the page when it holds for every unit: a unit
    // that finds otherwise says so in the word it publishes, the units behind it stop when they read that, and the page is
    // decoded again by k_delta_decode.
    const uint32_t slen = (uint32_t)tbase;
    const uint64_t n_values = (uint64_t)tbase >> 32;
    bool ok = one_window;
    __syncthreads();
    if (wv == 0 && ok && (uint32_t)lane < nb_here) {
      const uint32_t pj = win_start + (uint32_t)lane * spec_len;
      const uint32_t r = pj - win_a0;
      const uint32_t *q = win + (r >> 2);
      const uint32_t sh = (r & 3u) * 8u;
      const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
      const uint32_t x[4] = {__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh), __builtin_amdgcn_alignbit(d3, d2, sh),
                             __builtin_amdgcn_alignbit(d4, d3, sh)};
      const DeltaHdr h = delta_parse_header(x, pj, slen, b0 + (uint32_t)lane, n_values);
      ok = h.good && (h.size == spec_len || b0 + (uint32_t)lane + 1u == n_blocks);
      if (w.narrow && ok && !delta_narrow_ok(h.md, h.widths)) c.chunk_status[kLayoutMismatch] = 1;
      E = DeltaBlock{h.data_off, h.widths, h.md};
    }
    if (wv == 0 && lane == 0) s_fail = 0;
    if (wv == 0 && __ballot(!ok) != 0ull) {
      if (lane == 0) {
        __hip_atomic_store(&w.gsum[2 * u], 3ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&w.gsum[2 * u + 1], 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.state[p].delta_fast = 2;
        s_fail = 1;
      }
    }
  }
  if (wv == 0 && (uint32_t)lane < kUnitBlocks) s_ent[lane] = E;
  __syncthreads();
  if (spec_len != 0u && s_fail) return;      // (uniform) a block of another length: the page is decoded again
  if (wv != 0 && (uint32_t)lane < kUnitBlocks) E = s_ent[lane];
  const uint32_t wsum = (E.widths & 0xFFu) + ((E.widths >> 8) & 0xFFu) + ((E.widths >> 16) & 0xFFu) + (E.widths >> 24);
  const uint32_t data_end = E.data_off + 4u * wsum;

  // Stage the bytes of blocks [bs, be) - as many as the window holds - and return be.  The window starts at the
  // 16-byte boundary at or below the first block; every block has its bytes inside the stream (k_delta_blocks).
  auto stage = [&](uint32_t bs) -> uint32_t {
    const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)E.data_off, (int)bs);
    const uint32_t a0 = start - ((src_al + start) & 15u);
    const bool fits = (uint32_t)lane >= bs && (uint32_t)lane < nb_here && data_end - a0 <= kWinBytes;
    const uint64_t fm = __ballot(fits) >> bs;
    uint32_t n = fm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fm);
    if (bs + n < nb_here && n > 1u) n &= ~1u;       // steps take blocks in pairs
    const uint32_t be = bs + n;                     // (n >= 1: a block of the widest kind - ten header bytes, 64-bit deltas - is 1038 bytes, and the window never has fewer than kMinWinBytes)
    const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)data_end, (int)(be - 1u));
    __syncthreads();                                // the window's previous readers are done
    fill_window(a0, end);
    __syncthreads();
    win_a0 = a0;
    return be;
  };

  // ---- the running value in front of every block of the unit ----
  uint32_t be = one_window ? nb_here : (nb_here ? stage(0u) : 0u);
  const bool wide_mine = ((((E.widths & 0x7F7F7F7Fu) + 0x67676767u) | E.widths) & 0x80808080u) != 0u;   // a width byte above 24
  // The quick unit: one window holds it and no width exceeds 24 bits.  Then the sum of the packed deltas in front of
  // any value of a block fits 32 bits.  A first sweep over the window - two blocks per step, four values per lane, as
  // the values will be written - leaves every block's total in LDS and keeps nothing else, so that the unit's total
  // goes out as early as possible; the second sweep, once the unit's starting value is known, unpacks again and
  // stores.
  const bool quick = nb_here != 0u && one_window && __ballot(wide_mine) == 0ull;
  const int half = lane >> 5, m = (lane >> 3) & 3, k = lane & 7;
  // step i of the quick sweeps: the running sums c[0..3] of this lane's four packed deltas
  auto unpack4 = [&](int i, uint32_t (&cs)[4], uint32_t &blk_l, bool &live) {
    blk_l = 2u * (uint32_t)i + (uint32_t)half;
    live = blk_l < nb_here;
    const uint32_t ei = live ? blk_l : 0u;
    const uint32_t data_off = s_ent[ei].data_off, widths = s_ent[ei].widths;
    const uint32_t wm = (widths >> (8 * m)) & 0xFFu;
    const uint32_t pre = m ? ((widths * 0x01010101u) >> (8 * (m - 1))) & 0xFFu : 0u;   // widths of the miniblocks before this one
    const uint32_t bit = 32u * pre + 4u * (uint32_t)k * wm;
    const uint32_t r = data_off - win_a0 + (bit >> 3);
    const uint32_t shift = (r & 3u) * 8u + (bit & 7u);
    const uint32_t *q = win + (r >> 2);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
    const uint32_t g0 = __builtin_amdgcn_alignbit(d1, d0, shift), g1 = __builtin_amdgcn_alignbit(d2, d1, shift),
                   g2 = __builtin_amdgcn_alignbit(d3, d2, shift);
    const uint64_t lo = ((uint64_t)g1 << 32) | g0;
    const uint32_t mask = (1u << wm) - 1u;
    const uint32_t t2 = 2u * wm;
    const uint64_t lo2 = (lo >> t2) | ((uint64_t)g2 << ((64u - t2) & 63u));
    cs[0] = (uint32_t)lo & mask;
    cs[1] = cs[0] + ((uint32_t)(lo >> wm) & mask);
    cs[2] = cs[1] + ((uint32_t)lo2 & mask);
    cs[3] = cs[2] + ((uint32_t)(lo2 >> wm) & mask);
  };
  T s = 0;
  // the steps of this wavefront in the two sweeps: half of them each (units of eight blocks - four steps - are one wavefront's)
  const int n_steps = (int)kUnitBlocks / 2;
  constexpr int kNW = kUnitThreads / 64;
  const bool shared = kNW > 1 && n_steps >= 4 * kNW;
  const int step_lo = shared ? wv * (n_steps / kNW) : 0, step_hi = shared ? step_lo + n_steps / kNW : (wv == 0 ? n_steps : 0);
  if (quick) {
    // (four steps at a time without a branch between them, so that the LDS reads of one step overlap the arithmetic
    //  of another; steps past the unit's last block unpack block 0 again and keep nothing)
    for (int i0 = step_lo; i0 < step_hi; i0 += 4) {
      if (2u * (uint32_t)i0 < nb_here) {
#pragma unroll
        for (int i = i0; i < i0 + 4; i++) {
          uint32_t cs[4], blk_l;
          bool live;
          unpack4(i, cs, blk_l, live);
          const uint32_t sc = half_incl_scan(cs[3]);
          if ((lane & 31) == 31 && live) s_tot[blk_l] = sc;
        }
      }
    }
    __syncthreads();
    if (wv == 0 && (uint32_t)lane < nb_here && b0 + (uint32_t)lane + 1u < n_blocks)   // (the last block's sum is nobody's start)
      s = (T)s_tot[lane] + (T)128 * DeltaTraits<T>::zigzag(E.min_delta);
  } else {
    for (uint32_t bs = 0; bs < nb_here;) {
      if (wv == 0 && (uint32_t)lane >= bs && (uint32_t)lane < be && b0 + (uint32_t)lane + 1u < n_blocks)
        s = delta_block_sum<T>(win, E.data_off - win_a0, E.widths) + (T)128 * DeltaTraits<T>::zigzag(E.min_delta);
      bs = be;
      if (bs < nb_here) be = stage(bs);
    }
  }
  // (the block sums, their scan, the unit's total and the look-back are the first wavefront's alone)
  T incl = 0;
  if (wv == 0) {
    incl = wave_incl_scan(s, lane);
    const T total = bcast_lane(incl, 63);
    if (lane == 0) {
      publish_word(&w.gsum[2 * u], (uint32_t)total);
      if (sizeof(T) == 8) publish_word(&w.gsum[2 * u + 1], (uint32_t)((uint64_t)total >> 32));
    }
  }
  if (nb_here == 0u) return;
  // the units before this one in the page: they started no later than this one and publish before they wait
  T before = 0;
  bool late = false;
  for (uint32_t j0 = 0; j0 < g && wv == 0; j0 += 64u) {      // (the first wavefront looks back for both)
    const uint32_t j = j0 + (uint32_t)lane;
    T v = 0;
    if (j < g) {
      const uint64_t *slot = &w.gsum[2 * (u - (int64_t)g + (int64_t)j)];
      uint64_t a = 0, b = 0;
      uint32_t spins = 0;
      while (((a >> 32) == 0 || (b >> 32) == 0) && ++spins < kSpinLimit) {
        if (spins > 1u) __builtin_amdgcn_s_sleep(2);
        poll_raw<sizeof(T) == 8>(slot, &a, &b);
      }
      late = late || (a >> 32) != 1 || (b >> 32) == 0;     // (never came, or: a unit in front found a block of another length)
      v = (T)((b << 32) | (uint32_t)a);
    }
    before += bcast_lane(wave_incl_scan(v, lane), 63);
  }
  if (wv == 0) {
    const bool any_late = __ballot(late) != 0ull;
    if (lane == 0) s_late = any_late ? 1 : 0;
    if ((uint32_t)lane < kUnitBlocks) s_prev[lane] = (uint64_t)(T)(first + before + (T)(incl - s));
  }
  __syncthreads();
  if (s_late) {
    // a total never came (the units of a page did not start in order): the page is decoded again by the
    // wavefront-per-page kernel that runs after this one
    if (tid == 0) c.state[p].delta_fast = 2;
    return;
  }

  // ---- values ----
  if (quick) {
    for (int i0 = step_lo; i0 < step_hi; i0 += 4) {
      if (2u * (uint32_t)i0 >= nb_here) break;
#pragma unroll
      for (int i = i0; i < i0 + 4; i++) {
        uint32_t e4[4], blk_l;
        bool live;
        unpack4(i, e4, blk_l, live);
        const uint32_t ex = half_incl_scan(e4[3]) - e4[3];
        e4[0] += ex; e4[1] += ex; e4[2] += ex; e4[3] += ex;
        const uint32_t ei = live ? blk_l : 0u;
        const T prev = (T)s_prev[ei];
        const T md = DeltaTraits<T>::zigzag(s_ent[ei].min_delta);
        const uint32_t l4 = (uint32_t)(lane & 31) * 4u;
        const T a1 = prev + md * (T)(l4 + 1u), a2 = a1 + md, a3 = a2 + md, a4 = a3 + md;
        const T v0 = a1 + (T)e4[0], v1 = a2 + (T)e4[1], v2 = a3 + (T)e4[2], v3 = a4 + (T)e4[3];
        const uint64_t idx = 1ull + (uint64_t)(b0 + blk_l) * 128u + l4;
        if (sizeof(T) == 8 && __ballot(live && idx + 3 < (uint64_t)cnt) == ~0ull) {
          // 64-bit values: a lane's 32 bytes go through LDS so that each store instruction writes 1 KiB without gaps
          // (lane-major in LDS = value order in memory, the two blocks of a step being neighbours)
          s_stage[2 * lane] = make_uint4((uint32_t)v0, (uint32_t)((uint64_t)v0 >> 32), (uint32_t)v1, (uint32_t)((uint64_t)v1 >> 32));
          s_stage[2 * lane + 1] = make_uint4((uint32_t)v2, (uint32_t)((uint64_t)v2 >> 32), (uint32_t)v3, (uint32_t)((uint64_t)v3 >> 32));
          const uint4 qa = s_stage[lane], qb = s_stage[64 + lane];
          struct Q { uint32_t a, b, c, d; } __attribute__((packed));
          uint8_t *region = reinterpret_cast<uint8_t *>(out + 1ull + (uint64_t)(b0 + 2u * (uint32_t)i) * 128u);
          *reinterpret_cast<Q *>(region + 16 * lane) = Q{qa.x, qa.y, qa.z, qa.w};
          *reinterpret_cast<Q *>(region + 1024 + 16 * lane) = Q{qb.x, qb.y, qb.z, qb.w};
        } else if (live) {
          if (idx + 3 < (uint64_t)cnt) {
            if (sizeof(T) == 8) {
              struct Q { uint64_t a, b; } __attribute__((packed));
              *reinterpret_cast<Q *>(out + idx) = Q{(uint64_t)v0, (uint64_t)v1};
              *reinterpret_cast<Q *>(out + idx + 2) = Q{(uint64_t)v2, (uint64_t)v3};
            } else {
              struct Q { uint32_t a, b, c, d; } __attribute__((packed));
              *reinterpret_cast<Q *>(out + idx) = Q{(uint32_t)v0, (uint32_t)v1, (uint32_t)v2, (uint32_t)v3};
            }
          } else {
            if (idx < (uint64_t)cnt) out[idx] = v0;
            if (idx + 1 < (uint64_t)cnt) out[idx + 1] = v1;
            if (idx + 2 < (uint64_t)cnt) out[idx + 2] = v2;
          }
        }
      }
    }
    return;
  }
  // anything else: widths up to 64, or more bytes than the window holds
  if (!one_window) be = stage(0u);
  for (uint32_t bs = 0; bs < nb_here;) {
#pragma unroll 1
    for (uint32_t i = bs; i < be && wv == 0; i += 2u) {      // (the slow kinds of unit: one wavefront's work, the other one helps to stage)
      const uint32_t blk_l = i + (uint32_t)half;
      const bool live = blk_l < be;
      const DeltaBlock e = s_ent[live ? blk_l : i];
      const T prev = (T)s_prev[live ? blk_l : i];
      const uint32_t wm = (e.widths >> (8 * m)) & 0xFFu;
      const uint32_t pre = m ? ((e.widths * 0x01010101u) >> (8 * (m - 1))) & 0xFFu : 0u;
      const uint32_t bit0 = 32u * pre + 4u * (uint32_t)k * wm;
      const uint32_t r = e.data_off - win_a0 + (bit0 >> 3);
      const uint32_t shift = (r & 3u) * 8u + (bit0 & 7u);
      const uint32_t *q = win + (r >> 2);
      const uint64_t mask = wm >= 64u ? ~0ull : ((1ull << wm) - 1ull);
      const T md = DeltaTraits<T>::zigzag(e.min_delta);
      T t[4];
#pragma unroll
      for (int v = 0; v < 4; v++) {
        // every value from the three dwords that hold it
        const uint32_t bit = shift + (uint32_t)v * wm;
        const uint32_t *wp = q + (bit >> 5);
        const uint32_t sh = bit & 31u;
        const uint64_t a0 = wp[0], a1 = wp[1], a2 = wp[2];
        uint64_t v64 = (a0 | (a1 << 32)) >> sh;
        if (sh) v64 |= a2 << (64u - sh);
        t[v] = (T)(wm ? (v64 & mask) : 0ull);
      }
      const T p0 = t[0] + md, p1 = p0 + t[1] + md, p2 = p1 + t[2] + md, p3 = p2 + t[3] + md;
      const T base = prev + (T)(half_incl_scan(p3) - p3);
      if (live) {
        const uint64_t idx = 1ull + (uint64_t)(b0 + blk_l) * 128u + (uint32_t)(lane & 31) * 4u;
        const T v0 = base + p0, v1 = base + p1, v2 = base + p2, v3 = base + p3;
        if (idx + 3 < (uint64_t)cnt) {
          if (sizeof(T) == 8) {
            struct Q { uint64_t a, b; } __attribute__((packed));
            *reinterpret_cast<Q *>(out + idx) = Q{(uint64_t)v0, (uint64_t)v1};
            *reinterpret_cast<Q *>(out + idx + 2) = Q{(uint64_t)v2, (uint64_t)v3};
          } else {
            struct Q { uint32_t a, b, c, d; } __attribute__((packed));
            *reinterpret_cast<Q *>(out + idx) = Q{(uint32_t)v0, (uint32_t)v1, (uint32_t)v2, (uint32_t)v3};
          }
        } else {
          if (idx < (uint64_t)cnt) out[idx] = v0;
          if (idx + 1 < (uint64_t)cnt) out[idx + 1] = v1;
          if (idx + 2 < (uint64_t)cnt) out[idx + 2] = v2;
        }
      }
    }
    bs = be;
    if (bs < nb_here) be = stage(bs);
  }
}

// (sized for the smallest units)
size_t delta_work_bytes(int64_t slots, int nd) {
  const size_t nb = (size_t)slots / 128 + 3 * (size_t)nd + 8, nu = ((size_t)slots >> (7 + kMinUnitShift)) + (size_t)nd + 8;
  return 16 * nb + (16 + sizeof(DeltaUnit)) * nu + 128;
}

hipError_t launch_delta_decode(const ChunkCtx &c, uint8_t *values_out, int elem_size, int encoding, int src_sel, void *work, int64_t slots,
                               double bytes_per_value, bool spec, bool narrow, hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  if (narrow && (!work || elem_size != 4)) return hipErrorInvalidValue;
  if (work) {
    // Units as large as the window allows, the window as small as the units need: from the bytes the pages spend per
    // value (0: not known - the lengths of byte arrays, which share their pages with the bytes: 32 blocks, 6.5 KiB).
    int ushift = 5, win_bytes = 6656;
    if (bytes_per_value > 0.0) {
      const double block_bytes = 128.0 * bytes_per_value;     // (block headers included: it is what the page spends)
      ushift = kMaxUnitShift;
      while (ushift > kMinUnitShift && block_bytes * (double)(1 << ushift) + 32.0 > (double)kMaxWinBytes) ushift--;
      win_bytes = ((int)(block_bytes * (double)(1 << ushift) * 1.01) + 64 + 511) / 512 * 512;
      if (win_bytes < kMinWinBytes) win_bytes = kMinWinBytes;
      if (win_bytes > kMaxWinBytes) win_bytes = kMaxWinBytes;
    }
    const size_t nb = (size_t)slots / 128 + 3 * (size_t)c.nd + 8, nu = ((size_t)slots >> (7 + kMinUnitShift)) + (size_t)c.nd + 8;
    DeltaWork w;
    w.blocks = static_cast<DeltaBlock *>(work);
    w.gsum = reinterpret_cast<uint64_t *>(w.blocks + nb);
    w.units = reinterpret_cast<DeltaUnit *>((reinterpret_cast<uintptr_t>(w.gsum + 2 * nu) + 63) & ~(uintptr_t)63);
    w.n_units = (int64_t)(((size_t)slots >> (7 + ushift)) + (size_t)c.nd);
    w.ushift = ushift;
    w.win_bytes = win_bytes;
    w.spec = spec ? 1 : 0;
    w.narrow = narrow ? 1 : 0;
    hipLaunchKernelGGL(k_delta_blocks, dim3(c.nd), dim3(64), 0, stream, c, encoding, src_sel, w);
    const dim3 grid((unsigned)w.n_units);
    const size_t shmem = (size_t)win_bytes + kWinSlack;
    if (elem_size == 4) hipLaunchKernelGGL((k_delta_units<uint32_t, 256>), grid, dim3(256), shmem, stream, c, reinterpret_cast<uint32_t *>(values_out), w);
    else hipLaunchKernelGGL((k_delta_units<uint64_t, 256>), grid, dim3(256), shmem, stream, c, reinterpret_cast<uint64_t *>(values_out), w);
  }
  if (elem_size == 4)
    hipLaunchKernelGGL(k_delta_decode<uint32_t>, dim3(c.nd), dim3(64), 0, stream, c, reinterpret_cast<uint32_t *>(values_out), encoding, src_sel, narrow ? 1 : 0);
  else
    hipLaunchKernelGGL(k_delta_decode<uint64_t>, dim3(c.nd), dim3(64), 0, stream, c, reinterpret_cast<uint64_t *>(values_out), encoding, src_sel, 0);
  return hipGetLastError();
}

}  // namespace pqd
