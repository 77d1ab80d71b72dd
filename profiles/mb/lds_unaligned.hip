// Throughput of unaligned dword accesses: LDS writes and buffer loads, against aligned ones and byte writes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_lds(int iters, int mis, uint32_t *out) {
  __shared__ __align__(16) uint8_t s[4][2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane >> 4, rl = lane & 15;
  uint8_t *base = s[wave] + row * 512 + 4 * rl + mis;
  uint32_t v = threadIdx.x * 2654435761u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      uint8_t *wp = base + 4 * ((u * 9) & 63);
      if (MODE == 0) *reinterpret_cast<uint32_t *>(wp) = v;
      else {
        typedef volatile __attribute__((address_space(3))) uint8_t *LdsBytes; LdsBytes vp = (LdsBytes)wp;
        vp[0] = (uint8_t)v; vp[1] = (uint8_t)(v >> 8); vp[2] = (uint8_t)(v >> 16); vp[3] = (uint8_t)(v >> 24);
      }
      v = v * 3 + 1;
      asm volatile("" ::: "memory");
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = *reinterpret_cast<uint32_t *>(s[wave] + 4 * lane) + v;
}

__global__ __launch_bounds__(256) void k_buf(const uint8_t *in, uint32_t n, int iters, int mis, uint32_t *out) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)n, 0x00020000);
  const int lane = threadIdx.x & 63;
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 1u, acc = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      s = s * 1664525u + 1013904223u;
      // a wavefront reads 256 consecutive bytes somewhere in the buffer (64-byte aligned start, plus the misalignment)
      const uint32_t start = (uint32_t)__builtin_amdgcn_readfirstlane((int)((s >> 8) % (n / 64 - 8))) * 64u;
      x[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(start + 4u * lane + (uint32_t)mis), 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) acc += x[j];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  uint32_t *out; uint8_t *in;
  const uint32_t n = 64 << 20;
  CK(hipMalloc(&out, 256 * 8 * 256 * 4)); CK(hipMalloc(&in, n)); CK(hipMemset(in, 1, n));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 256 * 8;
  for (int mode = 0; mode < 2; mode++)
    for (int mis = 0; mis < 4; mis++) {
      const int iters = 256;
      if (mode == 0) hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(256), 0, 0, 16, mis, out); else hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 0, 0, 16, mis, out);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(256), 0, 0, iters, mis, out); else hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 0, 0, iters, mis, out);
      CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double wi = (double)blocks * 4 * iters * 16;     // wavefront write steps (one dword per lane each)
      printf("lds %s mis=%d: %.3f ms, %.1f G dword-steps/s chip-wide, %.2f clk per step per CU at 2.4 GHz\n", mode ? "4 x b8" : "b32   ", mis, ms, wi / ms / 1e6,
             ms * 1e-3 * 2.4e9 / (wi / 256));
    }
  for (int mis = 0; mis < 4; mis++) {
    const int iters = 512;
    hipLaunchKernelGGL(k_buf, dim3(blocks), dim3(256), 0, 0, in, n, 8, mis, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_buf, dim3(blocks), dim3(256), 0, 0, in, n, iters, mis, out);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double wi = (double)blocks * 4 * iters * 8;
    printf("buffer_load_dword mis=%d: %.3f ms, %.1f G loads/s chip-wide (wavefront instructions), %.2f clk per instruction per CU\n", mis, ms, wi / ms / 1e6,
           ms * 1e-3 * 2.4e9 / (wi / 256));
  }
  return 0;
}
