// BYTE_ARRAY columns on gfx950: PLAIN length chains, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY (prefix
// chains), output offsets and the byte copies.
//
// Reference behaviour reproduced here: encoding/encodingread.go:98-112 (PLAIN BYTE_ARRAY), :385-409
// (DELTA_LENGTH_BYTE_ARRAY), :411-435 (DELTA_BYTE_ARRAY), layout/page.go:684-706 (values[:cnt]).
//
// Every non-null value is first described by a 64-bit "pair": bits 0..39 = chunk offset of its bytes,
// bits 40..63 = length (strings of 16 MiB or more are rejected).  A chunk-wide exclusive scan of the lengths
// gives the output offsets; the copies follow.
#include <hip/hip_runtime.h>
#include "pqd_kernels.h"

namespace pqd {

constexpr uint64_t kSrcMask = (1ull << 40) - 1;
constexpr int kMaxStrLen = (1 << 24) - 1;

__device__ __forceinline__ void set_status_b(int32_t *p, int code) { atomicCAS(p, 0, code); }
__device__ __forceinline__ uint64_t make_pair(int64_t src, int64_t len) { return ((uint64_t)len << 40) | ((uint64_t)src & kSrcMask); }

// ---------------------------------------------------------------------------------------------
// PLAIN BYTE_ARRAY: [u32 length][bytes] ... — a serial chain per page; one thread walks one page.
// dict_pages == nullptr: every PLAIN data page (pairs -> pairs_out[dense_base + i]); else the listed dictionary
// pages (pairs -> pairs_out[dict_off + i]).
// Reference semantics (encodingread.go:98-112): the 4-byte length is read leniently (zero filled when cut),
// io.EOF only when a value starts exactly at the end.  A string that runs past the page would be zero
// padded by the reference; here the page is refused (PQD_ERR_REFUSED).
// ---------------------------------------------------------------------------------------------
__global__ void k_plain_ba_walk(ChunkCtx c, const int32_t *__restrict__ dict_pages, int n_dict, uint64_t *pairs_out) {
  int p;
  int64_t cnt, pos, end;
  uint64_t *out;
  if (dict_pages) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_dict) return;
    p = dict_pages[k];
    const PageIn pi = c.pages[p];
    cnt = pi.num_values;
    pos = pi.body_off;
    end = pi.body_off + pi.body_len;
    out = pairs_out + pi.dict_off;
  } else {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= c.nd) return;
    p = c.streams[2 * c.nd + d].page;
    if (c.pages[p].encoding != PQD_ENC_PLAIN) return;
    const PageState ps = c.state[p];
    if (ps.status != 0) return;
    cnt = ps.cnt;
    pos = ps.val_off;
    end = ps.val_off + ps.val_len;
    out = pairs_out + ps.dense_base;
  }
  int status = PQD_OK;
  if (cnt < 0) status = PQD_ERR_MALFORMED;
  for (int64_t i = 0; i < cnt && status == PQD_OK; i++) {
    if (pos >= end) { status = PQD_ERR_EOF; break; }
    uint32_t ln = 0;
    for (int k = 0; k < 4; k++)
      if (pos + k < end) ln |= (uint32_t)c.in[pos + k] << (8 * k);
    pos = pos + 4 < end ? pos + 4 : end;
    if ((int64_t)ln > end - pos || ln > (uint32_t)kMaxStrLen) { status = PQD_ERR_REFUSED; break; }
    out[i] = make_pair(pos, ln);
    pos += ln;
  }
  if (status != PQD_OK) set_status_b(&c.state[p].status, status);
}

// ---------------------------------------------------------------------------------------------
// DELTA_LENGTH_BYTE_ARRAY / the suffix half of DELTA_BYTE_ARRAY: lengths (already delta-decoded into
// lens[dense_base + i]) -> pairs; the bytes of value i start where value i-1 ended, the first at the end of
// the length stream (PageState.aux0).  One block per data page, running carry.
// ---------------------------------------------------------------------------------------------
// With page_base / doff given (a chunk of DELTA_LENGTH_BYTE_ARRAY pages only) no pairs are written: the strings of a page
// follow each other in the output as they do in the page, so value i goes to page_base[d] + (its position - the payload's
// start), page_base being the payload sizes of the pages in front added up (k_dlba_page_bases).  A page that holds bytes
// behind its last string makes those sums wrong for every page after it: noted in chunk_status[kLayoutMismatch].
constexpr int kPairsThreads = 1024;   // (a page is one workgroup's: the more threads, the fewer rounds and barriers)
__global__ __launch_bounds__(kPairsThreads) void k_dlba_pairs(ChunkCtx c, int encoding, const int64_t *__restrict__ lens, uint64_t *__restrict__ pairs_out,
                                                    const int64_t *__restrict__ page_base, int64_t *__restrict__ doff) {
  constexpr int kPer = 4;  // consecutive values per thread and round
  __shared__ int64_t s_wave[kPairsThreads / 64];
  __shared__ int64_t s_carry;
  __shared__ int s_bad, s_refused;
  const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = c.streams[2 * c.nd + d].page;
  if (c.pages[p].encoding != encoding) return;
  PageState *ps = &c.state[p];
  if (ps->status != 0 || ps->cnt <= 0) return;
  const int64_t cnt = ps->cnt, base = ps->dense_base;
  const int64_t end = ps->val_off + ps->val_len;
  const int64_t pay0 = ps->aux0;
  const int64_t out0 = doff ? page_base[d] - pay0 : 0;
  if (tid == 0) { s_carry = ps->aux0; s_bad = 0; s_refused = 0; }
  __syncthreads();
  for (int64_t i0 = 0; i0 < cnt; i0 += kPairsThreads * kPer) {
    const int64_t carry = s_carry;
    int64_t ln[kPer], sum = 0;
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int64_t i = i0 + (int64_t)tid * kPer + j;
      ln[j] = i < cnt ? (int64_t)lens[base + i] : 0;
      if (ln[j] < 0) { s_bad = 1; ln[j] = 0; }  // make([]byte, huge) panics in the reference
      if (ln[j] > kMaxStrLen) { s_refused = 1; ln[j] = 0; }
      sum += ln[j];
    }
    // exclusive scan of the threads' sums: shuffles inside the wave, LDS across the waves
    int64_t x = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int64_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();   // also: every thread has read s_carry
    int64_t src = carry + x - sum;
    for (int k = 0; k < wave; k++) src += s_wave[k];
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int64_t i = i0 + (int64_t)tid * kPer + j;
      if (i < cnt) {
        if (ln[j] > 0 && src + ln[j] > end) s_refused = 1;  // the reference would zero pad a cut string (or hit EOF at the next one)
        if (doff) doff[base + i] = out0 + src;
        else pairs_out[base + i] = make_pair(src, ln[j]);
      }
      src += ln[j];
    }
    if (tid == kPairsThreads - 1) s_carry = src;   // the last thread ends where the round ends
    __syncthreads();
  }
  if (tid == 0) {
    if (s_refused) set_status_b(&ps->status, PQD_ERR_REFUSED);
    else if (s_bad) set_status_b(&ps->status, PQD_ERR_MALFORMED);
    ps->aux1 = s_carry;  // end of the payload
    if (doff) {
      doff[base + cnt] = out0 + s_carry;     // (the next page's first offset, or the total)
      if (s_carry != end && !s_refused && !s_bad) c.chunk_status[kLayoutMismatch] = 1;
    }
  }
}

// inclusive prefix sum over the 64 lanes (DPP row shifts and broadcasts)
__device__ __forceinline__ uint32_t dba_wave_incl_sum(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31
  return x;
}

// The offsets of a chunk of DELTA_LENGTH_BYTE_ARRAY pages from 32-bit lengths (launch_delta_decode, narrow): what k_dlba_pairs
// does with page_base / doff given, four lengths per thread as one vector, the sums of a wavefront by a DPP scan (a
// wavefront's 256 lengths of at most kMaxStrLen add up to less than 2^32), two 16-byte stores of offsets per thread.
__global__ __launch_bounds__(kPairsThreads) void k_dlba_offsets32(ChunkCtx c, const int32_t *__restrict__ lens, const int64_t *__restrict__ page_base,
                                                                int64_t *__restrict__ doff) {
  __shared__ int64_t s_wave[kPairsThreads / 64];
  __shared__ int64_t s_carry;
  __shared__ int s_bad, s_refused;
  const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = c.streams[2 * c.nd + d].page;
  if (c.pages[p].encoding != PQD_ENC_DELTA_LENGTH_BYTE_ARRAY) return;
  // (32-bit lengths that may not be the reference's 64-bit ones - the delta decode has said so - decide nothing: the call is
  //  decoded again from 64-bit lengths)
  if (c.chunk_status[kLayoutMismatch] != 0) return;
  PageState *ps = &c.state[p];
  if (ps->status != 0 || ps->cnt <= 0) return;
  const int64_t cnt = ps->cnt, base = ps->dense_base;
  const int64_t end = ps->val_off + ps->val_len;
  const int64_t pay0 = ps->aux0;
  const int64_t out0 = page_base[d] - pay0;
  if (tid == 0) { s_carry = pay0; s_bad = 0; s_refused = 0; }
  __syncthreads();
  struct I4 { int32_t v[4]; } __attribute__((packed));
  struct L2 { int64_t v[2]; } __attribute__((packed));
  for (int64_t i0 = 0; i0 < cnt; i0 += kPairsThreads * 4) {
    const int64_t carry = s_carry;
    const int64_t i = i0 + 4 * (int64_t)tid;
    int32_t ln[4] = {0, 0, 0, 0};
    if (i + 3 < cnt) {
      const I4 x = *reinterpret_cast<const I4 *>(lens + base + i);
#pragma unroll
      for (int j = 0; j < 4; j++) ln[j] = x.v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (i + j < cnt) ln[j] = lens[base + i + j];
    }
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (ln[j] < 0) { s_bad = 1; ln[j] = 0; }  // make([]byte, huge) panics in the reference
      if (ln[j] > kMaxStrLen) { s_refused = 1; ln[j] = 0; }
      sum += (uint32_t)ln[j];
    }
    const uint32_t incl = dba_wave_incl_sum(sum);
    if (lane == 63) s_wave[wave] = (int64_t)incl;
    __syncthreads();   // also: every thread has read s_carry
    int64_t src = carry + (int64_t)(incl - sum);
    for (int k = 0; k < wave; k++) src += s_wave[k];
    int64_t o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (i + j < cnt && ln[j] > 0 && src + ln[j] > end) s_refused = 1;  // the reference would zero pad a cut string (or hit EOF at the next one)
      o[j] = out0 + src;
      src += ln[j];
    }
    if (i + 3 < cnt) {
      *reinterpret_cast<L2 *>(doff + base + i) = L2{{o[0], o[1]}};
      *reinterpret_cast<L2 *>(doff + base + i + 2) = L2{{o[2], o[3]}};
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (i + j < cnt) doff[base + i + j] = o[j];
    }
    if (tid == kPairsThreads - 1) s_carry = src;   // the last thread ends where the round ends
    __syncthreads();
  }
  if (tid == 0) {
    if (s_refused) set_status_b(&ps->status, PQD_ERR_REFUSED);
    else if (s_bad) set_status_b(&ps->status, PQD_ERR_MALFORMED);
    ps->aux1 = s_carry;  // end of the payload
    doff[base + cnt] = out0 + s_carry;     // (the next page's first offset, or the total)
    if (s_carry != end && !s_refused && !s_bad) c.chunk_status[kLayoutMismatch] = 1;
  }
}

// page_base[d] = payload bytes of the DELTA_LENGTH_BYTE_ARRAY pages in front of data page d (what lies between the end of a
// page's length stream and the end of the page), page_base[nd] = their total; one workgroup.
__global__ __launch_bounds__(1024) void k_dlba_page_bases(ChunkCtx c, int64_t *page_base) {
  __shared__ int64_t s_w[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  auto payload = [&](int d) -> int64_t {
    const int p = c.streams[2 * c.nd + d].page;
    const PageState &ps = c.state[p];
    if (c.pages[p].encoding != PQD_ENC_DELTA_LENGTH_BYTE_ARRAY || ps.status != 0 || ps.cnt <= 0) return 0;
    const int64_t n = ps.val_off + ps.val_len - ps.aux0;
    return n > 0 ? n : 0;
  };
  const int per = (c.nd + 1023) / 1024;
  const int lo = tid * per, hi = lo + per < c.nd ? lo + per : c.nd;
  int64_t sum = 0;
  for (int d = lo; d < hi; d++) sum += payload(d);
  int64_t incl = sum;
  for (int o = 1; o < 64; o <<= 1) {
    const int64_t y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  int64_t before = 0, total = 0;
#pragma unroll
  for (int q = 0; q < 16; q++) { if (q < wv) before += s_w[q]; total += s_w[q]; }
  int64_t run = before + incl - sum;
  for (int d = lo; d < hi; d++) { page_base[d] = run; run += payload(d); }
  if (tid == 0) page_base[c.nd] = total;
}

// the slots behind the last value repeat the byte total (nullable columns: their dense offsets cover all slots)
__global__ __launch_bounds__(256) void k_fill_offsets_tail(int64_t *doff, int64_t from, int64_t n_slots, const int64_t *n_dense, const int64_t *total) {
  const int64_t i = from + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= *n_dense && i <= n_slots) doff[i] = *total;
}

// ---------------------------------------------------------------------------------------------
// Chunk-wide exclusive scan of the value lengths -> dense output offsets (n + 1 entries).
// len(i) = pair length (+ prefix length for DELTA_BYTE_ARRAY pages; the first value of a page has none,
// encodingread.go:427).
// ---------------------------------------------------------------------------------------------
constexpr int kScanTile = 2048;

struct LenSrc {
  const uint64_t *pairs;
  const int64_t *prefix;     // DELTA_BYTE_ARRAY prefix lengths (dense), or nullptr
  const int32_t *page_of;    // unused
};

__device__ __forceinline__ int64_t value_len(const uint64_t *pairs, const int64_t *prefix, int64_t i) {
  int64_t ln = (int64_t)(pairs[i] >> 40);
  if (prefix) ln += prefix[i];
  return ln;
}

__global__ __launch_bounds__(256) void k_len_tile_sums(const uint64_t *pairs, const int64_t *prefix, int64_t n, int64_t *partial) {
  __shared__ int64_t s_w[4];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  int64_t sum = 0;
  for (int k = threadIdx.x; k < kScanTile; k += 256) {
    const int64_t i = base + k;
    if (i < n) sum += value_len(pairs, prefix, i);
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan of the tile sums in place; total -> partial[n_tiles]
__global__ __launch_bounds__(1024) void k_scan_partials(int64_t *partial, int n_tiles) {
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
  if (threadIdx.x == 0) partial[n_tiles] = carry;
}

// Offsets of one tile: every wave takes a quarter of the tile, two neighbouring values per lane and step (128-bit
// loads and stores that cover contiguous memory across the wave), a wave scan per step, the carry in a scalar.
__global__ __launch_bounds__(256) void k_len_offsets(const uint64_t *__restrict__ pairs, const int64_t *__restrict__ prefix, int64_t n,
                                                     const int64_t *__restrict__ partial, int n_tiles, int64_t *__restrict__ doff) {
  __shared__ int64_t s_wave[4];
  constexpr int kPerWave = kScanTile / 4, kSteps = kPerWave / 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + wave * kPerWave;
  int64_t l0[kSteps], l1[kSteps];
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    const int64_t i = base + s * 128 + lane * 2;
    l0[s] = 0; l1[s] = 0;
    if (i + 1 < n) {
      const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(pairs + i);
      l0[s] = (int64_t)(a.x >> 40); l1[s] = (int64_t)(a.y >> 40);
      if (prefix) { const longlong2 q = *reinterpret_cast<const longlong2 *>(prefix + i); l0[s] += q.x; l1[s] += q.y; }
    } else if (i < n) {
      l0[s] = value_len(pairs, prefix, i);
    }
  }
  // exclusive offsets inside the wave's quarter
  int64_t carry = 0, ex[kSteps];
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    int64_t x = l0[s] + l1[s];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int64_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    ex[s] = carry + x - (l0[s] + l1[s]);
    carry += __shfl(x, 63);
  }
  if (lane == 0) s_wave[wave] = carry;
  __syncthreads();
  int64_t off = partial[blockIdx.x];
  for (int w = 0; w < wave; w++) off += s_wave[w];
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    const int64_t i = base + s * 128 + lane * 2;
    const int64_t o0 = off + ex[s];
    if (i + 1 < n) *reinterpret_cast<longlong2 *>(doff + i) = longlong2{o0, o0 + l0[s]};
    else if (i < n) doff[i] = o0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) doff[n] = partial[n_tiles];
}

// ---------------------------------------------------------------------------------------------
// The same offsets in one pass over the lengths: every workgroup takes the next tile (a ticket, so that tiles start in
// order), adds up its tile, publishes the sum, and finds the sum of all tiles before it by looking back - 64 tiles at a
// time - for the nearest tile that has already published its inclusive prefix, adding the plain sums of the tiles in
// between; it then publishes its own inclusive prefix and writes its offsets.  A tile's word carries a two-bit flag (0
// nothing yet, 1 the tile's sum, 2 the inclusive prefix) above 62 bits of value: written and read whole.
//
// SRC 0: lengths from the pairs (+ prefix lengths); SRC 1: lengths as int64 (DELTA_LENGTH_BYTE_ARRAY straight from the
// decoded length stream: a negative length - make([]byte, n) panics in the reference - or one of 16 MiB or more is noted
// per page in page_flags, bits 0 and 1; n_dense: on the device, the number of values in front; the slots behind read 0).
// ---------------------------------------------------------------------------------------------
constexpr uint64_t kScanValueMask = (1ull << 62) - 1;
constexpr int kLbTile = 4096;             // values per tile of the one-pass scan
constexpr uint32_t kScanSpinLimit = 1u << 24;

struct LenScanArgs {
  const uint64_t *pairs;     // SRC 0
  const int64_t *prefix;     // SRC 0: or nullptr
  const int64_t *lens;       // SRC 1
  const int64_t *n_dense;    // SRC 1: lens[i] counts for i < *n_dense only
  int64_t n;                 // slots: doff gets n + 1 entries
  uint64_t *state;           // one word per tile, zeroed
  uint32_t *ticket;          // zeroed; [1]: set when a look-back gave up (never, with tiles started in ticket order)
  int64_t *doff;
  int64_t *total_out;        // or nullptr: the byte total once more (the handle's block of totals)
  int32_t *page_flags;       // SRC 1: per data page
};

template <int SRC>
__global__ __launch_bounds__(256) void k_len_scan(ChunkCtx c, LenScanArgs a) {
  __shared__ int64_t s_wave[4];
  __shared__ int64_t s_excl;
  __shared__ uint32_t s_tile;
  constexpr int kPerWave = kLbTile / 4, kSteps = kPerWave / 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_tile = atomicAdd(a.ticket, 1u);
  __syncthreads();
  const uint32_t tile = s_tile;
  const int64_t n = a.n;
  const int64_t lim = SRC == 1 ? (*a.n_dense < n ? *a.n_dense : n) : n;
  const int64_t base = (int64_t)tile * kLbTile + wave * kPerWave;
  int64_t l0[kSteps], l1[kSteps];
  int bad = 0;
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    const int64_t i = base + s * 128 + lane * 2;
    l0[s] = 0; l1[s] = 0;
    if (SRC == 0) {
      if (i + 1 < n) {
        const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(a.pairs + i);
        l0[s] = (int64_t)(q.x >> 40); l1[s] = (int64_t)(q.y >> 40);
        if (a.prefix) { const longlong2 r = *reinterpret_cast<const longlong2 *>(a.prefix + i); l0[s] += r.x; l1[s] += r.y; }
      } else if (i < n) {
        l0[s] = value_len(a.pairs, a.prefix, i);
      }
    } else {
      if (i + 1 < lim) { const longlong2 r = *reinterpret_cast<const longlong2 *>(a.lens + i); l0[s] = r.x; l1[s] = r.y; }
      else if (i < lim) l0[s] = a.lens[i];
      if (l0[s] < 0 || l0[s] > kMaxStrLen) { bad |= l0[s] < 0 ? 1 : 2; bad |= 4; l0[s] = 0; }
      if (l1[s] < 0 || l1[s] > kMaxStrLen) { bad |= l1[s] < 0 ? 1 : 2; bad |= 8; l1[s] = 0; }
    }
  }
  if (SRC == 1 && bad) {
    // (rare) the page of the bad length: the last data page whose values start at or before it
#pragma unroll 1
    for (int s = 0; s < kSteps; s++) {
      for (int k = 0; k < 2; k++) {
        const int64_t i = base + s * 128 + lane * 2 + k;
        if (i >= lim) continue;
        const int64_t v = a.lens[i];
        if (v >= 0 && v <= kMaxStrLen) continue;
        int lo = 0, hi = c.nd;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (c.state[c.streams[2 * c.nd + mid].page].dense_base <= i) lo = mid; else hi = mid;
        }
        atomicOr(&a.page_flags[lo], v < 0 ? 1 : 2);
      }
    }
  }
  // exclusive offsets inside the wave's quarter
  int64_t carry = 0, ex[kSteps];
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    int64_t x = l0[s] + l1[s];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int64_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    ex[s] = carry + x - (l0[s] + l1[s]);
    carry += __shfl(x, 63);
  }
  if (lane == 0) s_wave[wave] = carry;
  __syncthreads();
  if (wave == 0) {
    const int64_t sum = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (lane == 0) __hip_atomic_store(&a.state[tile], (1ull << 62) | ((uint64_t)sum & kScanValueMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int64_t excl = 0;
    int64_t at = (int64_t)tile - 1;
    bool gave_up = false;
    while (at >= 0) {
      const int64_t idx = at - lane;
      uint64_t w = 2ull << 62;                 // in front of tile 0: an inclusive prefix of nothing
      uint32_t spins = 0;
      if (idx >= 0) {
        w = __hip_atomic_load(&a.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((w >> 62) == 0 && ++spins < kScanSpinLimit) {
          __builtin_amdgcn_s_sleep(1);
          w = __hip_atomic_load(&a.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (__ballot((w >> 62) == 0) != 0ull) { gave_up = true; break; }
      const uint64_t pm = __ballot((w >> 62) == 2);
      const int first_p = pm ? (int)__builtin_ctzll(pm) : 64;       // the nearest tile that knows everything before it
      int64_t v = lane <= first_p ? (int64_t)(w & kScanValueMask) : 0;
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      excl += v;
      if (pm) break;
      at -= 64;
    }
    if (lane == 0) {
      if (gave_up) {      // (the call fails rather than hand out offsets that may be wrong)
        a.ticket[1] = 1u; excl = 0;
        if (c.nd > 0) set_status_b(&c.state[c.streams[2 * c.nd].page].status, PQD_ERR_HIP);
      }
      __hip_atomic_store(&a.state[tile], (2ull << 62) | ((uint64_t)(excl + sum) & kScanValueMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_excl = excl;
    }
  }
  __syncthreads();
  int64_t off = s_excl;
  for (int w = 0; w < wave; w++) off += s_wave[w];
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    const int64_t i = base + s * 128 + lane * 2;
    const int64_t o0 = off + ex[s];
    if (i + 1 <= n) *reinterpret_cast<longlong2 *>(a.doff + i) = longlong2{o0, o0 + l0[s]};     // (index n is the total: the array has n + 1 entries and slack)
    else if (i <= n) a.doff[i] = o0;
  }
  if (a.total_out && (int64_t)tile == n / kLbTile && threadIdx.x == 0) {
    // the tile that holds index n knows the total: everything before it plus its own values
    *a.total_out = s_excl + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  }
}

// DELTA_LENGTH_BYTE_ARRAY pages whose offsets came straight from the decoded lengths (k_len_scan<1>): what k_dlba_pairs
// checks otherwise - a bad length, strings that run past the page - and where the page's bytes start and end.
__global__ __launch_bounds__(64) void k_dlba_check(ChunkCtx c, const int64_t *__restrict__ doff, const int32_t *__restrict__ page_flags) {
  const int d = blockIdx.x * 64 + threadIdx.x;
  if (d >= c.nd) return;
  const int p = c.streams[2 * c.nd + d].page;
  if (c.pages[p].encoding != PQD_ENC_DELTA_LENGTH_BYTE_ARRAY) return;
  PageState *ps = &c.state[p];
  if (ps->status != 0 || ps->cnt <= 0) return;
  const int64_t total = doff[ps->dense_base + ps->cnt] - doff[ps->dense_base];
  const int f = page_flags[d];
  if ((f & 2) || ps->aux0 + total > ps->val_off + ps->val_len) set_status_b(&ps->status, PQD_ERR_REFUSED);   // the reference would zero pad a cut string (or hit EOF at the next one)
  else if (f & 1) set_status_b(&ps->status, PQD_ERR_MALFORMED);
  ps->aux1 = ps->aux0 + total;  // end of the payload
}

// lengths of failed pages read as 0 (their words may hold anything, and the chunk-wide offsets are sums over all of them)
__global__ __launch_bounds__(256) void k_void_failed_lens(ChunkCtx c, int64_t *lens) {
  const int d = blockIdx.x;
  const int p = c.streams[2 * c.nd + d].page;
  const PageState *ps = &c.state[p];
  if (ps->status == 0 || ps->cnt <= 0) return;
  const int64_t base = ps->dense_base;
  for (int i = threadIdx.x; i < ps->cnt; i += 256) lens[base + i] = 0;
}

// ---------------------------------------------------------------------------------------------
// Byte copies.  General form: one thread per value (PLAIN, dictionary).  DELTA_LENGTH_BYTE_ARRAY pages are
// one contiguous block each: flat 128-bit copy.
// ---------------------------------------------------------------------------------------------
// grid.x = data pages, grid.y = slices of the page; pages copied by the two special kernels are skipped
// (every copy kernel first compares the byte total, doff[n], with the capacity of the output it was given: the
//  output is sized from an estimate when the call is enqueued, and a call whose strings need more is copied again
//  into a larger buffer once the total is known)
__global__ __launch_bounds__(256) void k_copy_values(ChunkCtx c, const uint64_t *pairs, const int64_t *doff, int flat_dlba,
                                                     uint8_t *bytes_out, const int64_t *__restrict__ total, int64_t cap) {
  if (*total > cap) return;
  const int d = blockIdx.x;
  const int p = c.streams[2 * c.nd + d].page;
  const int enc = c.pages[p].encoding;
  if (enc == PQD_ENC_DELTA_BYTE_ARRAY || (flat_dlba && enc == PQD_ENC_DELTA_LENGTH_BYTE_ARRAY)) return;
  const PageState ps = c.state[p];
  if (ps.status != 0) return;
  for (int64_t k = (int64_t)blockIdx.y * 256 + threadIdx.x; k < ps.cnt; k += (int64_t)gridDim.y * 256) {
    const int64_t i = ps.dense_base + k;
    const uint64_t pr = pairs[i];
    const uint8_t *src = c.in + (pr & kSrcMask);
    uint8_t *dst = bytes_out + doff[i];
    const int len = (int)(pr >> 40);
    for (int b = 0; b < len; b++) dst[b] = src[b];
  }
}

struct V16U { uint32_t v[4]; } __attribute__((packed));

// A page's strings are one contiguous piece.  k_copy_plan notes per page where it comes from, where it goes and how long it
// is (nothing for pages that are not copied); the copying workgroups then start with one load instead of a chain of four.
// Workgroups that follow each other take pieces that follow each other in memory (the piece index is the fast grid
// dimension): 100 M strings 1.3 ms against 1.6 with the page as the fast dimension and every workgroup striding over
// its page - the pieces in flight at any moment then lie megabytes apart.
constexpr int kCopyPiece = 8192;   // bytes per workgroup (grid.x * kCopyPiece covers the longest page)
struct CopyPlan { int64_t src, dst, n; };

__global__ __launch_bounds__(256) void k_copy_plan(ChunkCtx c, const uint64_t *pairs, const int64_t *doff, const int64_t *__restrict__ total, int64_t cap,
                                                   CopyPlan *plan) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= c.nd) return;
  CopyPlan pl{0, 0, 0};
  const int p = c.streams[2 * c.nd + d].page;
  const PageState ps = c.state[p];
  // (nothing when the output is too small for the call's bytes, or when the offsets are about to be computed again)
  if (*total <= cap && (pairs || c.chunk_status[kLayoutMismatch] == 0) && c.pages[p].encoding == PQD_ENC_DELTA_LENGTH_BYTE_ARRAY && ps.status == 0 &&
      ps.cnt > 0) {
    const int64_t first = ps.dense_base;
    pl.src = pairs ? (int64_t)(pairs[first] & kSrcMask) : ps.aux0;
    pl.dst = doff[first];
    pl.n = doff[first + ps.cnt] - doff[first];
  }
  plan[d] = pl;
}

__global__ __launch_bounds__(256) void k_copy_dlba_pages(ChunkCtx c, const CopyPlan *__restrict__ plan, uint8_t *bytes_out, int strided) {
  const CopyPlan pl = plan[strided ? blockIdx.x : blockIdx.y];
  const int64_t n = pl.n;
  const int64_t piece0 = strided ? 0 : (int64_t)blockIdx.x * kCopyPiece;
  if (piece0 >= n) return;
  const uint8_t *src = c.in + pl.src;
  uint8_t *dst = bytes_out + pl.dst;
  // The stores are aligned (whole lines leave a wavefront), the loads take whatever alignment that leaves them: the first
  // bytes up to the output's next 16-byte boundary are copied one by one.
  const int64_t head = n < 16 ? n : (int64_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u);
  const bool first_piece = strided ? blockIdx.y == 0 : blockIdx.x == 0;
  if (first_piece && threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
  int64_t b, step, lim;
  if (!strided) { b = head + piece0 + threadIdx.x * 16; step = 256 * 16; lim = head + piece0 + kCopyPiece; if (lim > n) lim = n; }
  else { b = head + ((int64_t)blockIdx.y * 256 + threadIdx.x) * 16; step = (int64_t)gridDim.y * 256 * 16; lim = n; }     // (more pages than grid.y allows)
  for (; b < lim; b += step) {
    if (b + 16 <= n) {
      const V16U x = *reinterpret_cast<const V16U *>(src + b);
      *reinterpret_cast<uint4 *>(dst + b) = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    } else {
      for (int k = 0; b + k < n; k++) dst[b + k] = src[b + k];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// DELTA_BYTE_ARRAY: out[i] = out[i-1][:p_i] + suffix_i, the chain restarting at each page's first value
// (whose own prefix length is ignored, encodingread.go:427).
//
// Byte k < p_i of string i is owned by the nearest earlier string j with p_j <= k (byte k - p_j of its
// suffix).  parent[i] = nearest earlier string with a smaller prefix length; following parents from i-1
// visits exactly the candidates in order of decreasing prefix length.  The parents are found by pointer
// jumping: cand[i] starts at i-1 and jumps to cand[cand[i]] while p[cand[i]] >= p[i].
// ---------------------------------------------------------------------------------------------
constexpr int kChainMax = 1024;  // longest string the in-order chain kernel keeps in LDS

// Per page: the first prefix length is ignored (encodingread.go:427), prefix lengths are range checked, and pages
// holding a string longer than kChainMax are marked for the parent-chain kernels (flag 1); the others go through
// k_dba_chain (flag 0).
// For pages of the first kind it also leaves the smallest prefix length of every 64 strings in bmin[dense_base + block]
// (the page's own stretch of an array with one int32 per value): k_dba_chain skips by them when it looks backwards.
__global__ __launch_bounds__(256) void k_dba_prepare(ChunkCtx c, int64_t *prefix, const uint64_t *pairs, int32_t *page_flag, int32_t *bmin) {
  __shared__ int s_bad, s_long;
  const int d = blockIdx.x, tid = threadIdx.x;
  const int p = c.streams[2 * c.nd + d].page;
  if (tid == 0) page_flag[d] = 0;
  if (c.pages[p].encoding != PQD_ENC_DELTA_BYTE_ARRAY) return;
  PageState *ps = &c.state[p];
  if (ps->status != 0 || ps->cnt <= 0) return;
  const int64_t base = ps->dense_base;
  const int cnt = ps->cnt;
  if (tid == 0) { prefix[base] = 0; s_bad = 0; s_long = 0; }
  __syncthreads();
  int bad = 0, lng = 0;
  for (int i0 = 0; i0 < cnt; i0 += 256) {
    const int i = i0 + tid;
    int64_t pmin = 0x7fffffff;
    if (i < cnt) {
      const int64_t pi = prefix[base + i];
      if (pi < 0 || pi > kMaxStrLen) bad = 1;
      else if (pi + (int64_t)(pairs[base + i] >> 40) > kChainMax) lng = 1;
      // a prefix longer than the previous string is a slice out of range in the reference (encodingread.go:430)
      if (i > 0 && pi > (i == 1 ? 0 : prefix[base + i - 1]) + (int64_t)(pairs[base + i - 1] >> 40)) bad = 1;
      pmin = pi < 0 ? 0 : (pi > 0x7fffffff ? 0x7fffffff : pi);
    }
    // (a wavefront holds the 64 strings of one block)
    int v = (int)pmin;
    for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(v, o); v = y < v ? y : v; }
    if ((tid & 63) == 0 && i < cnt) bmin[base + (i >> 6)] = v;
  }
  if (bad) s_bad = 1;
  if (lng) s_long = 1;
  __syncthreads();
  if (tid == 0) {
    if (s_bad) set_status_b(&ps->status, PQD_ERR_MALFORMED);
    else page_flag[d] = s_long;
  }
}

// ---------------------------------------------------------------------------------------------
// A chunk of DELTA_BYTE_ARRAY pages only, every string at most kChainMax bytes: no descriptor per value.  Where a string's
// suffix lies and where the string goes follow from sums of the two decoded length arrays, kept per block of 64 strings:
// k_dba_sums leaves {suffix bytes, string bytes} of every block (and its smallest prefix length), k_dba_page_scan turns
// them into offsets inside the page, k_dba_page_bases gives every page its place in the output, and k_dba_chain works
// out the rest (one wave scan per block) as it walks the strings, writing the offsets on the way.  These kernels decide
// nothing about the input: anything but clean pages - a length out of range, a prefix longer than the string before it,
// a string above kChainMax, suffixes running past the page, a page that failed earlier - sets
// chunk_status[kLayoutMismatch] and the call is decoded again by the kernels that keep a descriptor per value.
// Block b of a page lies at index dense_base + b of the arrays (a page has fewer blocks than strings).
// ---------------------------------------------------------------------------------------------
// (a thread takes four consecutive strings, so a block of 64 is one row of 16 lanes and its sums are row reductions: four
//  blocks per wavefront step, a dozen instructions per block)
__global__ __launch_bounds__(256) void k_dba_sums(ChunkCtx c, int32_t *prefix, const int32_t *__restrict__ lens, uint2 *__restrict__ blk,
                                                  int32_t *__restrict__ bmin) {
  const int d = blockIdx.x, lane = threadIdx.x & 63;
  const int p = c.streams[2 * c.nd + d].page;
  const PageState *ps = &c.state[p];
  if (ps->status != 0) {     // (its strings read as empty ones: the other kernels know how)
    if (blockIdx.y == 0 && threadIdx.x == 0) c.chunk_status[kLayoutMismatch] = 1;
    return;
  }
  if (ps->cnt <= 0) return;
  const int64_t base = ps->dense_base;
  const int cnt = ps->cnt;
  struct I4 { int32_t v[4]; } __attribute__((packed));
  bool odd = false;
  for (int g = (int)blockIdx.y * 4 + ((int)threadIdx.x >> 6); 256 * g < cnt; g += (int)gridDim.y * 4) {
    const int i0 = 256 * g + 4 * lane;
    int P[4] = {0, 0, 0, 0}, S[4] = {0, 0, 0, 0};
    if (i0 + 3 < cnt) {
      const I4 pv = *reinterpret_cast<const I4 *>(prefix + base + i0), sv = *reinterpret_cast<const I4 *>(lens + base + i0);
#pragma unroll
      for (int k = 0; k < 4; k++) { P[k] = pv.v[k]; S[k] = sv.v[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (i0 + k < cnt) { P[k] = prefix[base + i0 + k]; S[k] = lens[base + i0 + k]; }
    }
    // (lane 0: the length of the string in front of the group)
    int before0 = 0;
    if (lane == 0 && g > 0) before0 = prefix[base + i0 - 1] + lens[base + i0 - 1];     // (i0 - 1 >= 255: not the page's first string)
    if (i0 == 0) { P[0] = 0; prefix[base] = 0; }      // (the first prefix length is ignored, encodingread.go:427)
    int sum_s = 0, sum_l = 0, mn = 0x7fffffff, L[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if ((uint32_t)P[k] > (uint32_t)kChainMax || (uint32_t)S[k] > (uint32_t)kChainMax || (uint32_t)(P[k] + S[k]) > (uint32_t)kChainMax) { odd = true; P[k] = 0; S[k] = 0; }
      L[k] = P[k] + S[k];
      sum_s += S[k]; sum_l += L[k];
      if (i0 + k < cnt) mn = P[k] < mn ? P[k] : mn;
    }
    // a prefix longer than the string before it: a slice out of range in the reference (encodingread.go:430)
    int before = __builtin_amdgcn_update_dpp(0, L[3], 0x138, 0xF, 0xF, false);    // wave_shr:1
    if (lane == 0) before = before0;
    if (i0 > 0 && i0 < cnt && P[0] > before) odd = true;
#pragma unroll
    for (int k = 1; k < 4; k++)
      if (i0 + k < cnt && P[k] > L[k - 1]) odd = true;
    // sums and minimum over the 16 lanes of a row, in the row's last lane
#define PQD_ROW_STEP(CTRL) { \
      sum_s += __builtin_amdgcn_update_dpp(0, sum_s, CTRL, 0xF, 0xF, false); \
      sum_l += __builtin_amdgcn_update_dpp(0, sum_l, CTRL, 0xF, 0xF, false); \
      const int y = __builtin_amdgcn_update_dpp(0x7fffffff, mn, CTRL, 0xF, 0xF, false); \
      mn = y < mn ? y : mn; }
    PQD_ROW_STEP(0x111) PQD_ROW_STEP(0x112) PQD_ROW_STEP(0x114) PQD_ROW_STEP(0x118)      // row_shr:1, 2, 4, 8
#undef PQD_ROW_STEP
    const int b = i0 >> 6;
    if ((lane & 15) == 15 && 64 * b < cnt) { blk[base + b] = make_uint2((uint32_t)sum_s, (uint32_t)sum_l); bmin[base + b] = mn; }
  }
  if (__any(odd) && lane == 0) c.chunk_status[kLayoutMismatch] = 1;
}

// The block sums of a page -> where every block's suffixes start (from the page's first suffix byte) and where its strings
// go (from the page's first output byte); page_tot[d] = the page's output bytes.  One workgroup per page.
__global__ __launch_bounds__(256) void k_dba_page_scan(ChunkCtx c, uint2 *__restrict__ blk, int64_t *__restrict__ page_tot) {
  __shared__ uint32_t s_ws[4], s_wl[4];
  __shared__ uint32_t s_cs, s_cl;
  __shared__ int s_over;
  const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = c.streams[2 * c.nd + d].page;
  const PageState *ps = &c.state[p];
  if (ps->status != 0 || ps->cnt <= 0) { if (tid == 0) page_tot[d] = 0; return; }
  const int64_t base = ps->dense_base;
  const int n_blk = (ps->cnt + 63) >> 6;
  if (tid == 0) { s_cs = 0; s_cl = 0; s_over = 0; }
  __syncthreads();
  for (int b0 = 0; b0 < n_blk; b0 += 256) {
    const uint32_t cs = s_cs, cl = s_cl;
    const int b = b0 + tid;
    const uint2 v = b < n_blk ? blk[base + b] : make_uint2(0u, 0u);
    uint32_t xs = v.x, xl = v.y;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t ys = (uint32_t)__shfl_up((int)xs, o), yl = (uint32_t)__shfl_up((int)xl, o);
      if (lane >= o) { xs += ys; xl += yl; }
    }
    if (lane == 63) { s_ws[wave] = xs; s_wl[wave] = xl; }
    __syncthreads();   // also: every thread has read the carries
    uint32_t es = cs + xs - v.x, el = cl + xl - v.y;
    for (int k = 0; k < wave; k++) { es += s_ws[k]; el += s_wl[k]; }
    if (b < n_blk) blk[base + b] = make_uint2(es, el);
    // (a block holds at most 64 KiB, so 32 bits hold the sums of 32 Ki blocks: beyond that the page is left to the other kernels)
    if (el + v.y < el || es + v.x < es || (el + v.y) > 0x7FFFFFFFu) s_over = 1;
    if (tid == 255) { s_cs = es + v.x; s_cl = el + v.y; }
    __syncthreads();
  }
  if (tid == 0) {
    page_tot[d] = (int64_t)s_cl;
    // suffixes that run past the page (the reference would zero pad a cut string, or hit EOF at the next one)
    if (s_over || ps->aux0 + (int64_t)s_cs > ps->val_off + ps->val_len) c.chunk_status[kLayoutMismatch] = 1;
  }
}

// page_tot[d] -> page_base[d] = output bytes of the pages in front, page_base[nd] = their total; one workgroup.
__global__ __launch_bounds__(1024) void k_dba_page_bases(int nd, const int64_t *__restrict__ page_tot, int64_t *__restrict__ page_base) {
  __shared__ int64_t s_w[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (nd + 1023) / 1024;
  const int lo = tid * per, hi = lo + per < nd ? lo + per : nd;
  int64_t sum = 0;
  for (int d = lo; d < hi; d++) sum += page_tot[d];
  int64_t incl = sum;
  for (int o = 1; o < 64; o <<= 1) {
    const int64_t y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  int64_t before = 0, total = 0;
#pragma unroll
  for (int q = 0; q < 16; q++) { if (q < wv) before += s_w[q]; total += s_w[q]; }
  int64_t run = before + incl - sum;
  for (int d = lo; d < hi; d++) { const int64_t t = page_tot[d]; page_base[d] = run; run += t; }
  if (tid == 0) page_base[nd] = total;
}

// Pages without long strings, strings in order: lane j carries byte j of the previous string; byte j of the next
// one is that byte when j lies in the shared prefix, else a byte of the suffix.  Suffixes follow each other in the
// page and so do the outputs, so only the two lengths are per-string inputs.
//
// A string with prefix length 0 does not depend on its predecessor: the page is cut into kChainWaves ranges that
// each begin at such a string (the first one at or after an even split), one wavefront per range.  Large pages
// without such strings near the split (sorted keys: every string continues the one before) are cut at the split
// itself, and the wavefront first works out the bytes its first string shares with the string before it.
//
// A range is walked in batches of 64 strings (lane u holds the lengths of string u): the suffix bytes of the next
// half batch and the lengths of the batch after the next are in flight while the current half is chained together.
// The chain costs issue slots, not bandwidth, so the per-string work is pared down to three lane reads, one select,
// one buffer load and one LDS write.  The loads go through a buffer descriptor (offset register plus lane offset,
// checked against the record count as gfx950 does): one past the end of the chunk returns zero, and no execution
// mask is needed.  Batches holding a string of 65 .. kChainMax bytes keep the previous string in LDS instead.
struct DbaBatch {
  int P, L;            // this lane's string: prefix length and whole length
  uint32_t ldoff;      // this lane's string: suffix start minus prefix length, relative to in_base
  int mx;              // longest string of the batch
  int64_t in_base;     // chunk offset the batch's loads are relative to
  const uint8_t *src;  // suffix bytes of the batch, back to back
  uint8_t *out;        // output of the batch's first string
};

constexpr int kChainWaves = 8;
constexpr int kStageBytes = 32 * 64 + 16 + 64 + 48;  // half a batch of 64-byte strings, the misalignment, the last string's overhang
constexpr int kRsrcFlags = 0x00020000;  // raw buffer, 32-bit data format

__device__ __forceinline__ int64_t uniform64(int64_t v) {  // a value every lane holds alike, moved to scalar registers
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// LEAN: no descriptors and no offsets come in.  `lens` holds the suffix lengths, `blk` the block offsets k_dba_page_scan left,
// `page_base` the pages' places in the output; a batch is a block of 64 strings of the page (the strings of a block that lie
// outside the wavefront's range take part in the sums and are passed over by the chain), and the offsets of the strings go
// out to `doff_out` as the batches are set up - also when the byte output is too small for the bytes themselves, so that
// everything that depends on the offsets alone is right after the first launch.
struct DbaLean {
  const int32_t *lens;       // (and `prefix` points at 32-bit words too)
  const uint2 *blk;
  const int64_t *page_base;
  int64_t *doff_out;
};

template <bool LEAN>
__global__ __launch_bounds__(64 * kChainWaves, 4) void k_dba_chain(ChunkCtx c, const int64_t *__restrict__ prefix,
                                                                const uint64_t *__restrict__ pairs, const int64_t *__restrict__ doff,
                                                                const int32_t *__restrict__ page_flag, const int32_t *__restrict__ bmin_all,
                                                                uint8_t *__restrict__ bytes_out, const int64_t *__restrict__ total, int64_t cap,
                                                                DbaLean ln) {
  __shared__ uint8_t s_cur_all[kChainWaves][kChainMax + 64];
  __shared__ int s_cut[kChainWaves];
  __shared__ __align__(16) uint8_t s_stage_all[kChainWaves][kStageBytes];
  __shared__ int s_bnd[kChainWaves + 1];
  const bool over = *total > cap;     // (uniform)
  if (over && !LEAN) return;
  // (LEAN: a chunk that is not clean is decoded again by the other kernels; its lengths may be anything)
  if (LEAN && c.chunk_status[kLayoutMismatch] != 0) return;
  const int d = blockIdx.x, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  uint8_t *s_cur = s_cur_all[wave];
  uint8_t *stage = s_stage_all[wave];
  const int p = c.streams[2 * c.nd + d].page;
  if (c.pages[p].encoding != PQD_ENC_DELTA_BYTE_ARRAY || (!LEAN && page_flag[d] != 0)) return;
  PageState *ps = &c.state[p];
  if (ps->status != 0 || ps->cnt <= 0) return;
  const int64_t base = uniform64(ps->dense_base);
  const int cnt = __builtin_amdgcn_readfirstlane(ps->cnt);
  auto PF = [&](int64_t i) -> int64_t { return LEAN ? (int64_t)reinterpret_cast<const int32_t *>(prefix)[i] : prefix[i]; };

  // this wave's range [r_lo, r_hi): from the first string with prefix length 0 at or after an even split when there
  // is one close by; otherwise (large pages) from the split itself, the start state then being worked out by looking
  // backwards (below)
  {
    int lo = cnt, cut = 0;
    const int from = wave == 0 ? 0 : (int)((int64_t)cnt * wave / kChainWaves);
    const bool may_cut = cnt >= kChainWaves * 512;      // (boundaries stay in order: the search below covers 256 strings)
    const int upto = may_cut ? (from + 256 < cnt ? from + 256 : cnt) : cnt;
    for (int j0 = from; j0 < upto && lo == cnt; j0 += 64) {
      const int j = j0 + lane;
      const bool zero = j < upto && (j == 0 || PF(base + j) == 0);
      const uint64_t m = __ballot(zero);
      if (m) lo = j0 + (int)__builtin_ctzll(m);
    }
    if (lo == cnt && may_cut && from < cnt) {
      if (PF(base + from) <= 64) { lo = from; cut = 1; }
      else {
        for (int j0 = upto; j0 < cnt && lo == cnt; j0 += 64) {
          const int j = j0 + lane;
          const bool zero = j < cnt && PF(base + j) == 0;
          const uint64_t m = __ballot(zero);
          if (m) lo = j0 + (int)__builtin_ctzll(m);
        }
      }
    }
    if (lane == 0) { s_bnd[wave] = lo; s_cut[wave] = cut; }
    if (threadIdx.x == 0) s_bnd[kChainWaves] = cnt;
  }
  __syncthreads();
  const int r_lo = __builtin_amdgcn_readfirstlane(s_bnd[wave]);
  if (wave > 0 && s_bnd[wave - 1] == r_lo) return;  // same start as the wave before: that wave does the range
  int r_hi = cnt;
  for (int k = kChainWaves - 1; k > wave; k--)
    if (s_bnd[k] > r_lo) r_hi = s_bnd[k];            // starts never decrease with k
  r_hi = __builtin_amdgcn_readfirstlane(r_hi);
  if (r_lo >= r_hi) return;

  // (LEAN: i0 is a multiple of 64, the batches are the page's blocks one after the other from the range's first on: where a
  //  block's suffixes and strings start is looked up for the first one and carried along from there)
  struct Fetched { int P, S; uint64_t pr, pr0; int64_t o0; };
  const int64_t pay0 = LEAN ? uniform64(ps->aux0) : 0;                       // the page's first suffix byte
  const int64_t out0 = LEAN ? uniform64(ln.page_base[d]) : 0;                // the page's first output byte
  auto fetch_lengths = [&](int i0) {
    Fetched f{0, 0, 0, 0, 0};
    if (i0 < r_hi) {
      if (LEAN) {
        if (i0 + lane < cnt) {
          f.P = reinterpret_cast<const int32_t *>(prefix)[base + i0 + lane];
          f.S = ln.lens[base + i0 + lane];
        }
      } else {
        if (i0 + lane < r_hi) { f.P = (int)prefix[base + i0 + lane]; f.pr = pairs[base + i0 + lane]; f.S = (int)(f.pr >> 40); }
        f.pr0 = (uint64_t)uniform64((int64_t)pairs[base + i0]);
        f.o0 = uniform64(doff[base + i0]);
      }
    }
    return f;
  };
  uint32_t run_s = 0, run_l = 0;     // LEAN: suffix bytes and string bytes of the page in front of the next batch
  auto make_batch = [&](const Fetched &f, int i0) {
    DbaBatch b;
    if (LEAN) {
      // (32-bit sums relative to the page's first suffix byte and first output byte; the scans run on the DPP path, which
      //  needs no registers besides the value)
      const uint32_t L = (uint32_t)(f.P + f.S);
      const uint32_t is = dba_wave_incl_sum((uint32_t)f.S), il = dba_wave_incl_sum(L);
      const uint32_t srel = run_s + is - (uint32_t)f.S;
      const uint32_t drel = run_l + il - L;
      run_s += (uint32_t)__builtin_amdgcn_readlane((int)is, 63);
      run_l += (uint32_t)__builtin_amdgcn_readlane((int)il, 63);
      const bool live = i0 + lane >= r_lo && i0 + lane < r_hi;
      if (live) ln.doff_out[base + i0 + lane] = out0 + (int64_t)drel;
      const int first = i0 < r_lo ? r_lo - i0 : 0;      // the block's first string inside the range
      const int64_t src0 = pay0 + (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)srel, first);
      // (one opaque scalar: pieces of this sum kept per lane across the loop cost registers the byte loads need)
      b.out = reinterpret_cast<uint8_t *>(uniform64(reinterpret_cast<int64_t>(bytes_out) + out0 + (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)drel, first)));
      b.in_base = src0 > 64 ? src0 - 64 : 0;
      b.src = c.in + src0;
      b.ldoff = live ? srel - (uint32_t)f.P - (uint32_t)(b.in_base - pay0) : 0u;
      // (a string outside the range keeps the chain's state as it is - every lane below "prefix length 64" - and takes no room)
      b.P = live ? f.P : 64; b.L = live ? (int)L : 0;
      int mx = b.L;
      for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(mx, o); mx = y > mx ? y : mx; }
      b.mx = __builtin_amdgcn_readfirstlane(mx);
      return b;
    }
    b.P = f.P; b.L = f.P + f.S;
    int mx = b.L;
    for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(mx, o); mx = y > mx ? y : mx; }
    b.mx = __builtin_amdgcn_readfirstlane(mx);
    const int64_t src0 = (int64_t)(f.pr0 & kSrcMask);
    b.in_base = src0 > 64 ? src0 - 64 : 0;  // a prefix is at most 64 bytes here and never longer than the suffix bytes before it
    b.ldoff = (uint32_t)((int64_t)(f.pr & kSrcMask) - b.in_base - f.P);
    b.src = c.in + src0;
    b.out = bytes_out + f.o0;
    return b;
  };
  // byte `lane` of every string of half a batch, wherever it comes from (strings of at most 64 bytes); lanes
  // outside the suffix load a neighbouring byte that the chain never selects
  auto issue_loads = [&](const DbaBatch &b, int half, uint32_t ld[32]) {
    if (b.mx > 64) return;
    const int64_t left = c.in_len - b.in_base;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(c.in + b.in_base), 0, left > 0x7fffffff ? 0x7fffffff : (int)left, kRsrcFlags);
    if (half == 0) {
#pragma unroll
      for (int u = 0; u < 32; u++) ld[u] = __builtin_amdgcn_raw_buffer_load_b8(rs, lane, (int)__builtin_amdgcn_readlane(b.ldoff, u), 0);
    } else {
#pragma unroll
      for (int u = 0; u < 32; u++) ld[u] = __builtin_amdgcn_raw_buffer_load_b8(rs, lane, (int)__builtin_amdgcn_readlane(b.ldoff, u + 32), 0);
    }
  };
  // Half a batch is chained into an LDS staging area, every string written whole-wave at its offset (the bytes a
  // string writes beyond its length are overwritten by its successors, LDS operations of a wave being executed in
  // order), and leaves for memory as 128-bit stores; the staging area starts at the output's misalignment so that
  // LDS and memory chunks coincide, and the partial chunks at both ends go out bytewise.
  // (lanes beyond the batch hold lengths 0, 0: nothing is kept for them)
  uint32_t cur = 0;   // byte `lane` of the previous string
  if (__builtin_amdgcn_readfirstlane(s_cut[wave])) {
    // The range starts in the middle of a chain: lane k < P needs byte k of the string before the range.  That byte
    // belongs to the nearest earlier string whose prefix length is <= k (byte k - prefix of its suffix).  Looking
    // backwards, every string whose prefix length is below the smallest seen so far (m) owns the bytes [its prefix
    // length, m); blocks of 64 strings whose smallest prefix length is not below m are skipped by the minima that
    // k_dba_prepare left.  The page's first string has prefix length 0, so m reaches 0.
    const int32_t *bmin = bmin_all + base;
    int m = (int)uniform64(PF(base + r_lo));     // (<= 64)
    int own_j = 0, own_p = 0;
    int64_t own_src = 0;      // LEAN: where the owner's suffix starts
    int blk = (r_lo - 1) >> 6;
    while (m > 0) {
      const int j = 64 * blk + lane;
      int pj = 0x7fffffff;
      if (j < r_lo) pj = j == 0 ? 0 : (int)PF(base + j);
      int sx = 0;             // LEAN: suffix bytes of the block in front of string j
      int64_t blk_src = 0;
      if (LEAN) {
        const int sj = j < cnt ? (int)ln.lens[base + j] : 0;
        int x = sj;
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
        sx = x - sj;
        blk_src = pay0 + (int64_t)ln.blk[base + blk].x;
      }
      for (;;) {
        const uint64_t mk = __ballot(pj < m);
        if (!mk) break;
        const int u = 63 - (int)__builtin_clzll(mk);
        const int pu = __builtin_amdgcn_readlane(pj, u);
        const int su = LEAN ? __builtin_amdgcn_readlane(sx, u) : 0;
        if (lane >= pu && lane < m) { own_j = 64 * blk + u; own_p = pu; own_src = blk_src + su; }
        m = pu;
        if (lane >= u) pj = 0x7fffffff;            // the strings from u on are behind us
      }
      if (m == 0 || blk == 0) break;
      int b0 = blk - 1;                              // the nearest earlier block holding a prefix length below m
      for (;;) {
        const int nb = b0 - lane;
        const uint64_t hk = __ballot(nb >= 0 && bmin[nb] < m);
        if (hk) { blk = b0 - (int)__builtin_ctzll(hk); break; }
        b0 -= 64;
        if (b0 < 0) { blk = 0; break; }              // (block 0 holds the page's first string: never reached)
      }
    }
    if (lane < (int)uniform64(PF(base + r_lo))) {
      const int64_t at = (LEAN ? own_src : (int64_t)(pairs[base + own_j] & kSrcMask)) + (lane - own_p);
      cur = at < c.in_len ? c.in[at] : 0u;
    }
  }
  auto chain_half = [&](const DbaBatch &b, int half, const uint32_t ld[32], uint32_t out_off) {
    uint8_t *dst = b.out + out_off;
    const uint32_t a = (uint32_t)reinterpret_cast<uintptr_t>(dst) & 15u;
    uint8_t *wp = stage + a + lane;
    uint32_t total = 0;
#pragma unroll
    for (int u = 0; u < 32; u++) {
      const int pk = __builtin_amdgcn_readlane(b.P, half ? u + 32 : u), len = __builtin_amdgcn_readlane(b.L, half ? u + 32 : u);
      cur = lane < pk ? cur : ld[u];
      *wp = (uint8_t)cur;
      wp += len;
      total += (uint32_t)len;
    }
    const uint32_t end = a + total;
    uint8_t *g0 = dst - a;
    const uint32_t n_chunks = (end + 15) >> 4;
    for (uint32_t j = lane; j < n_chunks; j += 64) {
      const uint32_t lo = 16 * j;
      if (lo >= a && lo + 16 <= end) *reinterpret_cast<uint4 *>(g0 + lo) = *reinterpret_cast<const uint4 *>(stage + lo);
    }
    // first chunk (lanes 0-15) and last chunk (lanes 16-31) when they are partial
    {
      const uint32_t last = n_chunks ? 16 * (n_chunks - 1) : 0;
      const uint32_t pos = lane < 16 ? lane : last + (lane - 16);
      const bool head = lane < 16 && (a != 0 || end < 16);
      const bool tail = lane >= 16 && lane < 32 && n_chunks > 1 && (end & 15) != 0;
      if ((head || tail) && pos >= a && pos < end) g0[pos] = stage[pos];
    }
    return out_off + total;
  };

  // the byte loads run half a batch ahead of the chain, the lengths two batches ahead
  uint32_t ld_a[32], ld_b[32];
  const int i_first = LEAN ? (r_lo & ~63) : r_lo;
  if (LEAN) {
    const uint2 bo = ln.blk[base + (i_first >> 6)];
    run_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)bo.x);
    run_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)bo.y);
  }
  Fetched f1 = fetch_lengths(i_first);
  DbaBatch cb = make_batch(f1, i_first);
  if (!over) issue_loads(cb, 0, ld_a);
  f1 = fetch_lengths(i_first + 64);
  for (int i0 = i_first; i0 < r_hi; i0 += 64) {
    const int nb = r_hi - i0 < 64 ? r_hi - i0 : 64;
    const DbaBatch nbatch = make_batch(f1, i0 + 64);
    const Fetched f2 = fetch_lengths(i0 + 128);
    if (over) {
      // (LEAN, the byte output being too small: only the offsets, which make_batch has written)
    } else if (cb.mx <= 64) {
      if (nb > 32) issue_loads(cb, 1, ld_b);
      const uint32_t mid = chain_half(cb, 0, ld_a, 0);
      if (i0 + 64 < r_hi) issue_loads(nbatch, 0, ld_a);
      if (nb > 32) chain_half(cb, 1, ld_b, mid);
    } else {
      if (i0 + 64 < r_hi) issue_loads(nbatch, 0, ld_a);
      const uint8_t *src = cb.src;
      uint8_t *out = cb.out;
      s_cur[lane] = (uint8_t)cur;
      for (int k = 0; k < nb; k++) {
        const int pk = __builtin_amdgcn_readlane(cb.P, k), len = __builtin_amdgcn_readlane(cb.L, k);
        if (len == 0) continue;     // (also the strings of the block that lie outside the range)
        for (int j = lane; j < len; j += 64) {
          uint8_t b;
          if (j >= pk) { b = src[j - pk]; s_cur[j] = b; }
          else b = s_cur[j];
          out[j] = b;
        }
        __builtin_amdgcn_wave_barrier();
        src += len - pk;
        out += len;
      }
      cur = s_cur[lane];
    }
    cb = nbatch;
    f1 = f2;
  }
}

__global__ __launch_bounds__(1024) void k_dba_parents(ChunkCtx c, int64_t *prefix, int32_t *parent, const int32_t *page_flag) {
  __shared__ int s_changed;
  __shared__ int s_bad;
  const int d = blockIdx.x, tid = threadIdx.x;
  const int p = c.streams[2 * c.nd + d].page;
  if (c.pages[p].encoding != PQD_ENC_DELTA_BYTE_ARRAY || page_flag[d] == 0) return;
  PageState *ps = &c.state[p];
  if (ps->status != 0 || ps->cnt <= 0) return;
  const int64_t base = ps->dense_base;
  const int cnt = ps->cnt;
  if (tid == 0) { prefix[base] = 0; s_bad = 0; }
  __syncthreads();
  // parent index is page-relative; -1 = none (prefix 0 needs none)
  for (int i = tid; i < cnt; i += 1024) {
    const int64_t pi = prefix[base + i];
    if (pi < 0 || pi > kMaxStrLen) s_bad = 1;
    parent[base + i] = i - 1;
  }
  __syncthreads();
  if (s_bad) { if (tid == 0) set_status_b(&ps->status, PQD_ERR_MALFORMED); return; }
  for (int round = 0; round < 64; round++) {
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int i = tid; i < cnt; i += 1024) {
      int j = parent[base + i];
      if (j >= 0 && prefix[base + j] >= prefix[base + i]) {
        parent[base + i] = parent[base + j];  // may read a value another lane updates this round: still a valid jump
        s_changed = 1;
      }
    }
    __syncthreads();
    if (!s_changed) break;
    __syncthreads();
  }
  // stragglers (adversarial inputs): finish serially per element
  for (int i = tid; i < cnt; i += 1024) {
    int j = parent[base + i];
    const int64_t pi = prefix[base + i];
    while (j >= 0 && prefix[base + j] >= pi) j = parent[base + j];
    parent[base + i] = j;
  }
}

// one thread per string: suffix bytes straight, prefix bytes through the parent chain
__global__ __launch_bounds__(256) void k_dba_resolve(ChunkCtx c, const int64_t *prefix, const uint64_t *pairs, const int32_t *parent,
                                                     const int64_t *doff, const int32_t *page_flag, uint8_t *bytes_out,
                                                     const int64_t *__restrict__ total, int64_t cap) {
  if (*total > cap) return;
  const int d = blockIdx.x;
  const int p = c.streams[2 * c.nd + d].page;
  if (c.pages[p].encoding != PQD_ENC_DELTA_BYTE_ARRAY || page_flag[d] == 0) return;
  PageState *ps = &c.state[p];
  if (ps->status != 0 || ps->cnt <= 0) return;
  const int64_t base = ps->dense_base;
  const int cnt = ps->cnt;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < cnt; i += gridDim.y * 256) {
    const int64_t pi = prefix[base + i];
    const uint64_t pr = pairs[base + i];
    const int sl = (int)(pr >> 40);
    uint8_t *dst = bytes_out + doff[base + i];
    const uint8_t *suf = c.in + (pr & kSrcMask);
    for (int b = 0; b < sl; b++) dst[pi + b] = suf[b];
    if (pi > 0) {
      // prefix longer than the previous string: the reference panics on the slice
      const int64_t prev_len = prefix[base + i - 1] + (int64_t)(pairs[base + i - 1] >> 40);
      if (pi > prev_len) { set_status_b(&ps->status, PQD_ERR_MALFORMED); continue; }
      int j = i - 1;
      int64_t pj = prefix[base + j];
      const uint8_t *sj = c.in + (pairs[base + j] & kSrcMask);
      for (int64_t k = pi - 1; k >= 0; k--) {
        while (pj > k) {
          j = parent[base + j];
          pj = prefix[base + j];
          sj = c.in + (pairs[base + j] & kSrcMask);
        }
        if (k - pj >= (int64_t)(pairs[base + j] >> 40)) { set_status_b(&ps->status, PQD_ERR_MALFORMED); break; }
        dst[k] = sj[k - pj];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Null scatter of the offsets: slot s gets the offset of its value, null slots repeat the next offset
// (zero length).  rank via the per-tile valid counts (exclusive-scanned) as in the fixed-width scatter.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scatter_offsets(const int32_t *def, int64_t n_slots, int max_def, const int64_t *partial,
                                                         const int64_t *doff, int64_t *slot_off) {
  __shared__ int s_wave[4];
  const int64_t base = (int64_t)blockIdx.x * 4096;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t s0 = base + (int64_t)tid * 16;
  uint32_t valid = 0;
  for (int j = 0; j < 16; j++)
    if (s0 + j < n_slots && def[s0 + j] == max_def) valid |= 1u << j;
  int cnt = __builtin_popcount(valid);
  int incl = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) s_wave[wv] = incl;
  __syncthreads();
  int wbase = 0;
  for (int q = 0; q < wv; q++) wbase += s_wave[q];
  int64_t rank = partial[blockIdx.x] + wbase + incl - cnt;
  for (int j = 0; j < 16; j++) {
    const int64_t s = s0 + j;
    if (s >= n_slots) break;
    slot_off[s] = doff[rank];
    rank += (valid >> j) & 1u;
  }
  if (s0 <= n_slots && n_slots < s0 + 16) slot_off[n_slots] = doff[rank];  // the grid covers index n_slots
}

// ---------------------------------------------------------------------------------------------
hipError_t launch_plain_ba_walk(const ChunkCtx &c, const int32_t *dict_pages, int n_dict, uint64_t *pairs_out, hipStream_t stream) {
  if (dict_pages) { if (n_dict > 0) hipLaunchKernelGGL(k_plain_ba_walk, dim3((n_dict + 63) / 64), dim3(64), 0, stream, c, dict_pages, n_dict, pairs_out); }
  else if (c.nd > 0) hipLaunchKernelGGL(k_plain_ba_walk, dim3((c.nd + 63) / 64), dim3(64), 0, stream, c, dict_pages, 0, pairs_out);
  return hipGetLastError();
}

hipError_t launch_dlba_pairs(const ChunkCtx &c, int encoding, const int64_t *lens, uint64_t *pairs_out, hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_dlba_pairs, dim3(c.nd), dim3(kPairsThreads), 0, stream, c, encoding, lens, pairs_out, (const int64_t *)nullptr, (int64_t *)nullptr);
  return hipGetLastError();
}

hipError_t launch_dlba_offsets(const ChunkCtx &c, const int32_t *lens, int64_t *page_base, const int64_t *n_dense, int64_t n_slots, bool tail,
                               int64_t *doff, hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_dlba_page_bases, dim3(1), dim3(1024), 0, stream, c, page_base);
  hipLaunchKernelGGL(k_dlba_offsets32, dim3(c.nd), dim3(kPairsThreads), 0, stream, c, lens, (const int64_t *)page_base, doff);
  // (a page without values writes nothing: the entries from the last value on are filled in whenever slots can lie behind it,
  //  and index n_slots always)
  const int64_t from = tail ? 0 : n_slots;
  hipLaunchKernelGGL(k_fill_offsets_tail, dim3((unsigned)((n_slots - from + 256) / 256)), dim3(256), 0, stream, doff, from, n_slots,
                     n_dense, (const int64_t *)(page_base + c.nd));
  return hipGetLastError();
}

// Values of pages that have failed by now get length 0: their length words may hold anything, and the chunk-wide
// offsets (and the size of the byte output) are sums over all of them.
__global__ __launch_bounds__(256) void k_void_failed_pages(ChunkCtx c, uint64_t *pairs, int64_t *prefix) {
  const int d = blockIdx.x;
  const int p = c.streams[2 * c.nd + d].page;
  const PageState *ps = &c.state[p];
  if (ps->status == 0 || ps->cnt <= 0) return;
  const int64_t base = ps->dense_base;
  for (int i = threadIdx.x; i < ps->cnt; i += 256) {
    pairs[base + i] = 0;
    if (prefix) prefix[base + i] = 0;
  }
}

hipError_t launch_void_failed_pages(const ChunkCtx &c, uint64_t *pairs, int64_t *prefix, hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_void_failed_pages, dim3(c.nd), dim3(256), 0, stream, c, pairs, prefix);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// FIXED_LEN_BYTE_ARRAY columns whose pages are DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY (layout/page.go:684-706: the
// reference hands the decoded strings on whatever the column's type).  The column's image holds type_length bytes
// per value, so value i lies at i * type_length: the offsets are known without a scan, and the byte copies of the
// BYTE_ARRAY path write the image directly.  A string of any other length has no place in it: the page is refused.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flba_delta_check(ChunkCtx c, const uint64_t *__restrict__ pairs, const int64_t *__restrict__ prefix) {
  const int d = blockIdx.x;
  const int p = c.streams[2 * c.nd + d].page;
  const int enc = c.pages[p].encoding;
  if (enc != PQD_ENC_DELTA_LENGTH_BYTE_ARRAY && enc != PQD_ENC_DELTA_BYTE_ARRAY) return;
  PageState *ps = &c.state[p];
  if (ps->status != 0 || ps->cnt <= 0) return;
  const int64_t base = ps->dense_base;
  int odd = 0;
  for (int i = threadIdx.x; i < ps->cnt; i += 256) {
    int64_t ln = (int64_t)(pairs[base + i] >> 40);
    if (enc == PQD_ENC_DELTA_BYTE_ARRAY && i > 0) ln += prefix[base + i];   // (the first prefix length is ignored, encodingread.go:427)
    if (ln != c.type_length) odd = 1;
  }
  if (odd) set_status_b(&ps->status, PQD_ERR_REFUSED);
}

__global__ __launch_bounds__(256) void k_flba_offsets(int64_t n, int64_t type_length, int64_t *__restrict__ doff) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i <= n) doff[i] = i * type_length;
}

hipError_t launch_flba_delta_offsets(const ChunkCtx &c, const uint64_t *pairs, const int64_t *prefix, int64_t n, int64_t *doff,
                                     hipStream_t stream) {
  if (c.nd > 0) hipLaunchKernelGGL(k_flba_delta_check, dim3(c.nd), dim3(256), 0, stream, c, pairs, prefix);
  hipLaunchKernelGGL(k_flba_offsets, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, stream, n, (int64_t)c.type_length, doff);
  return hipGetLastError();
}

hipError_t launch_len_offsets(const uint64_t *pairs, const int64_t *prefix, int64_t n, int64_t *partial, int64_t *doff,
                              hipStream_t stream) {
  const int n_tiles = (int)((n + kScanTile - 1) / kScanTile);
  if (n_tiles > 0) hipLaunchKernelGGL(k_len_tile_sums, dim3(n_tiles), dim3(256), 0, stream, pairs, prefix, n, partial);
  hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, stream, partial, n_tiles);
  hipLaunchKernelGGL(k_len_offsets, dim3(n_tiles > 0 ? n_tiles : 1), dim3(256), 0, stream, pairs, prefix, n, partial, n_tiles, doff);
  return hipGetLastError();
}

// One-pass offsets.  work: (n / kLbTile + 2) * 8 + 64 bytes of device memory (tile words, then the ticket).
size_t len_scan_work_bytes(int64_t n) { return ((size_t)(n / kLbTile) + 2) * 8 + 64; }

hipError_t launch_len_scan(const ChunkCtx &c, const uint64_t *pairs, const int64_t *prefix, const int64_t *lens, const int64_t *n_dense,
                           int64_t n, void *work, int64_t *doff, int64_t *total_out, int32_t *page_flags, hipStream_t stream) {
  const int n_tiles = (int)(n / kLbTile) + 1;        // (the tile that holds index n included)
  hipError_t e = hipMemsetAsync(work, 0, len_scan_work_bytes(n), stream);
  if (e != hipSuccess) return e;
  LenScanArgs a;
  a.pairs = pairs; a.prefix = prefix; a.lens = lens; a.n_dense = n_dense; a.n = n;
  a.state = static_cast<uint64_t *>(work);
  a.ticket = reinterpret_cast<uint32_t *>(a.state + n_tiles + 1);
  a.doff = doff; a.total_out = total_out; a.page_flags = page_flags;
  if (lens) {
    if (c.nd > 0) {
      e = hipMemsetAsync(page_flags, 0, 4 * (size_t)c.nd, stream);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_void_failed_lens, dim3(c.nd), dim3(256), 0, stream, c, const_cast<int64_t *>(lens));
    }
    hipLaunchKernelGGL(k_len_scan<1>, dim3(n_tiles), dim3(256), 0, stream, c, a);
    if (c.nd > 0) hipLaunchKernelGGL(k_dlba_check, dim3((c.nd + 63) / 64), dim3(64), 0, stream, c, doff, page_flags);
  } else {
    hipLaunchKernelGGL(k_len_scan<0>, dim3(n_tiles), dim3(256), 0, stream, c, a);
  }
  return hipGetLastError();
}

hipError_t launch_copy_values(const ChunkCtx &c, const uint64_t *pairs, const int64_t *doff, int flat_dlba, uint8_t *bytes_out,
                              const int64_t *total, int64_t cap, hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_copy_values, dim3(c.nd, c.nd > 4096 ? 1 : 16), dim3(256), 0, stream, c, pairs, doff, flat_dlba, bytes_out, total, cap);
  return hipGetLastError();
}

size_t copy_plan_bytes(int nd) { return sizeof(CopyPlan) * ((size_t)nd + 1); }

hipError_t launch_copy_dlba_pages(const ChunkCtx &c, const uint64_t *pairs, const int64_t *doff, uint8_t *bytes_out,
                                  const int64_t *total, int64_t cap, int64_t max_page_bytes, void *plan, hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  CopyPlan *pl = static_cast<CopyPlan *>(plan);
  hipLaunchKernelGGL(k_copy_plan, dim3((c.nd + 255) / 256), dim3(256), 0, stream, c, pairs, doff, total, cap, pl);
  const int64_t pieces = (max_page_bytes + 16 + kCopyPiece - 1) / kCopyPiece;
  if (c.nd <= 65535 && pieces >= 1 && pieces <= 0x7FFFFFFF)
    hipLaunchKernelGGL(k_copy_dlba_pages, dim3((unsigned)pieces, c.nd), dim3(256), 0, stream, c, (const CopyPlan *)pl, bytes_out, 0);
  else
    hipLaunchKernelGGL(k_copy_dlba_pages, dim3(c.nd, 2), dim3(256), 0, stream, c, (const CopyPlan *)pl, bytes_out, 1);
  return hipGetLastError();
}

// phase 0: before the offsets exist (prefix checks, page classes, parent links of the pages with long strings);
// phase 1: the copies
hipError_t launch_dba_resolve(const ChunkCtx &c, int64_t *prefix, const uint64_t *pairs, int32_t *parent, const int64_t *doff,
                              int32_t *page_flag, uint8_t *bytes_out, int phase, const int64_t *total, int64_t cap,
                              hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  if (phase == 0) {
    hipLaunchKernelGGL(k_dba_prepare, dim3(c.nd), dim3(256), 0, stream, c, prefix, pairs, page_flag, parent);
    hipLaunchKernelGGL(k_dba_parents, dim3(c.nd), dim3(1024), 0, stream, c, prefix, parent, page_flag);
  } else {
    hipLaunchKernelGGL(k_dba_chain<false>, dim3(c.nd), dim3(64 * kChainWaves), 0, stream, c, prefix, pairs, doff, page_flag, parent, bytes_out, total, cap, DbaLean{});
    hipLaunchKernelGGL(k_dba_resolve, dim3(c.nd, 16), dim3(256), 0, stream, c, prefix, pairs, parent, doff, page_flag, bytes_out, total, cap);
  }
  return hipGetLastError();
}

// A chunk of DELTA_BYTE_ARRAY pages only, step 1: block sums, offsets inside the pages, the pages' places, the byte total
// (doff[n_slots], and every entry from the last value on: slots of a nullable column can lie behind it).
// blk: 8 bytes per value slot; bmin: 4 bytes per value slot; page_work: (2 nd + 2) * 8 bytes.
hipError_t launch_dba_lean_offsets(const ChunkCtx &c, int32_t *prefix, const int32_t *lens, void *blk, int32_t *bmin, int64_t *page_work,
                                   const int64_t *n_dense, int64_t n_slots, bool tail, int64_t *doff, hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  int64_t *page_tot = page_work, *page_base = page_work + c.nd;
  hipLaunchKernelGGL(k_dba_sums, dim3(c.nd, c.nd > 4096 ? 1 : 8), dim3(256), 0, stream, c, prefix, lens, static_cast<uint2 *>(blk), bmin);
  hipLaunchKernelGGL(k_dba_page_scan, dim3(c.nd), dim3(256), 0, stream, c, static_cast<uint2 *>(blk), page_tot);
  hipLaunchKernelGGL(k_dba_page_bases, dim3(1), dim3(1024), 0, stream, c.nd, (const int64_t *)page_tot, page_base);
  const int64_t from = tail ? 0 : n_slots;
  hipLaunchKernelGGL(k_fill_offsets_tail, dim3((unsigned)((n_slots - from + 256) / 256)), dim3(256), 0, stream, doff, from, n_slots,
                     n_dense, (const int64_t *)(page_base + c.nd));
  return hipGetLastError();
}

// ... step 2: the strings, and the offsets of all values in front of the last one
hipError_t launch_dba_lean_chain(const ChunkCtx &c, const int32_t *prefix, const int32_t *lens, const void *blk, const int32_t *bmin,
                                 const int64_t *page_work, int64_t *doff, uint8_t *bytes_out, const int64_t *total, int64_t cap,
                                 hipStream_t stream) {
  if (c.nd <= 0) return hipSuccess;
  DbaLean ln;
  ln.lens = lens; ln.blk = static_cast<const uint2 *>(blk); ln.page_base = page_work + c.nd; ln.doff_out = doff;
  hipLaunchKernelGGL(k_dba_chain<true>, dim3(c.nd), dim3(64 * kChainWaves), 0, stream, c, reinterpret_cast<const int64_t *>(prefix), (const uint64_t *)nullptr, (const int64_t *)nullptr,
                     (const int32_t *)nullptr, bmin, bytes_out, total, cap, ln);
  return hipGetLastError();
}

hipError_t launch_scatter_offsets(const int32_t *def, int64_t n_slots, int max_def, const int64_t *partial, const int64_t *doff,
                                  int64_t *slot_off, hipStream_t stream) {
  const int n_tiles = (int)((n_slots + 1 + 4095) / 4096);  // covers index n_slots itself
  hipLaunchKernelGGL(k_scatter_offsets, dim3(n_tiles), dim3(256), 0, stream, def, n_slots, max_def, partial, doff, slot_off);
  return hipGetLastError();
}

}  // namespace pqd
