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
__global__ __launch_bounds__(64) void k_delta_decode(ChunkCtx c, T *values_out, int encoding, int src_sel) {
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
    }
    hdr[i] = v;
  }
  const uint64_t block_size = hdr[0], n_mb = hdr[1], n_values = hdr[2];
  const T first = DeltaTraits<T>::zigzag(hdr[3]);
  if (n_mb == 0) status = PQD_ERR_MALFORMED;                       // division by zero
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
        // minDelta varint (at most 10 bytes; a longer one is rejected)
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
        if (hl >= 11) { status = PQD_ERR_MALFORMED; break; }
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

hipError_t launch_delta_decode(const ChunkCtx &c, uint8_t *values_out, int elem_size, int encoding, int src_sel, hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  if (elem_size == 4)
    hipLaunchKernelGGL(k_delta_decode<uint32_t>, dim3(c.nd), dim3(64), 0, stream, c, reinterpret_cast<uint32_t *>(values_out), encoding, src_sel);
  else
    hipLaunchKernelGGL(k_delta_decode<uint64_t>, dim3(c.nd), dim3(64), 0, stream, c, reinterpret_cast<uint64_t *>(values_out), encoding, src_sel);
  return hipGetLastError();
}

}  // namespace pqd
