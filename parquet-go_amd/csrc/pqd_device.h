// Device helpers shared by the expansion kernels (k_expand.hip, k_fused.hip): unaligned 16-byte stores, the unpack of
// one bit-packed group with compile-time bit positions, the stores of one work item.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pqd {

struct U32x4Unaligned { uint32_t v[4]; } __attribute__((packed));
struct U64x2Unaligned { uint64_t v[2]; } __attribute__((packed));
struct U64Unaligned { uint64_t v; } __attribute__((packed));

// One bit-packed group: W bytes at byte offset `rel` of the staging buffer -> eight values.
// The bytes are fetched as aligned dwords, realigned by the byte offset, then every value sits at a
// compile-time bit position.
template <int W>
__device__ __forceinline__ void unpack8(const uint32_t *s_stage, uint32_t rel, uint32_t x[8]) {
  if (W == 0) {
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = 0;
    return;
  }
  constexpr int ND = W == 0 ? 1 : (W + 3) / 4;
  constexpr uint32_t mask = W >= 32 ? 0xFFFFFFFFu : ((1u << (W & 31)) - 1u);
  const uint32_t a = rel >> 2, sb = (rel & 3u) * 8u;
  uint32_t raw[ND + 1];
#pragma unroll
  for (int i = 0; i <= ND; i++) raw[i] = s_stage[a + i];
  uint32_t d[ND + 1];
#pragma unroll
  for (int i = 0; i < ND; i++) d[i] = __builtin_amdgcn_alignbit(raw[i + 1], raw[i], sb);
  d[ND] = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int bit = j * W, lo = bit >> 5, sh = bit & 31;
    if (sh + W <= 32) x[j] = (d[lo] >> sh) & mask;
    else x[j] = __builtin_amdgcn_alignbit(d[lo + 1], d[lo], (uint32_t)sh) & mask;
  }
}

// Stores values j0 .. j1-1 of an item (eight output elements starting at `outp`).  An item that starts with its first
// value - all but a tile's first - goes out as whole vectors and at most three single elements, whatever its length: the
// lanes of a wavefront hold items of all lengths, and every form of store that any of them needs is issued for all.
__device__ __forceinline__ void store_item(uint32_t *outp, const uint32_t y[8], int j0, int j1) {
  if (j0 == 0) {
    if (j1 >= 4) {
      U32x4Unaligned o0 = {{y[0], y[1], y[2], y[3]}};
      reinterpret_cast<U32x4Unaligned *>(outp)[0] = o0;
    }
    if (j1 >= 8) {
      U32x4Unaligned o1 = {{y[4], y[5], y[6], y[7]}};
      reinterpret_cast<U32x4Unaligned *>(outp)[1] = o1;
    } else if (j1 > 0) {
      const bool hi = j1 >= 4;
      const int r = j1 & 3;
      uint32_t *q = outp + (hi ? 4 : 0);
      const uint32_t s0 = hi ? y[4] : y[0], s1 = hi ? y[5] : y[1], s2 = hi ? y[6] : y[2];
      if (r > 0) q[0] = s0;
      if (r > 1) q[1] = s1;
      if (r > 2) q[2] = s2;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (j >= j0 && j < j1) outp[j] = y[j];
  }
}
__device__ __forceinline__ void store_item(uint64_t *outp, const uint64_t y[8], int j0, int j1) {
  if (j0 == 0 && j1 == 8) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      U64x2Unaligned o = {{y[2 * q], y[2 * q + 1]}};
      reinterpret_cast<U64x2Unaligned *>(outp)[q] = o;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (j >= j0 && j < j1) outp[j] = y[j];
  }
}

}  // namespace pqd
