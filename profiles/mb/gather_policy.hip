// Random 4-byte lookups into an L2-resident table under different cache policies.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
__device__ __forceinline__ uint32_t look(const uint32_t *tab, __amdgpu_buffer_rsrc_t rs, uint32_t idx) {
  if (V == 0) return tab[idx];
  if (V == 1) return __builtin_nontemporal_load(&tab[idx]);
  if (V == 2) return __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4u, 0, 0);
  if (V == 3) return __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4u, 0, 1);
  if (V == 4) return __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4u, 0, 2);
  if (V == 5) return __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4u, 0, 16);
  if (V == 6) return __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4u, 0, 17);
  if (V == 7) return __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4u, 0, 3);
  if (V == 8) return __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4u, 0, 19);
  return 0;
}

template <int V>
__global__ __launch_bounds__(256) void k(const uint32_t *tab, uint32_t n_mask, uint32_t n, int iters, uint32_t *out) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)(n * 4u), 0x00020000);
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      s ^= s << 13; s ^= s >> 17; s ^= s << 5;
      uint32_t i = s & n_mask;
      if (i >= n) i -= n;
      x[j] = look<V>(tab, rs, i);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) acc += x[j];
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int V>
static void run(const char *name, const uint32_t *tab, uint32_t n, uint32_t *out) {
  uint32_t mask = 1; while (mask < n) mask <<= 1; mask -= 1;
  const int blocks = 256 * 8, iters = 256;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, tab, mask, n, 8, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, tab, mask, n, iters, out);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double lookups = (double)blocks * 256 * iters * 8;
  printf("%-28s n=%6u  %.3f ms  %.1f G lookups/s\n", name, n, ms, lookups / ms / 1e6);
}

int main() {
  uint32_t *tab, *out;
  CK(hipMalloc(&tab, 1 << 20)); CK(hipMalloc(&out, 256 * 8 * 256 * 4));
  CK(hipMemset(tab, 1, 1 << 20));
  const uint32_t sizes[] = {65536, 34048, 8192};
  for (uint32_t n : sizes) {
    run<0>("plain", tab, n, out);
    run<1>("nontemporal", tab, n, out);
    run<2>("buffer aux=0", tab, n, out);
    run<3>("buffer sc0", tab, n, out);
    run<4>("buffer nt", tab, n, out);
    run<5>("buffer sc1", tab, n, out);
    run<6>("buffer sc0 sc1", tab, n, out);
    run<7>("buffer sc0 nt", tab, n, out);
    run<8>("buffer sc0 sc1 nt", tab, n, out);
  }
  return 0;
}
