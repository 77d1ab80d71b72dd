// Do unaligned dword accesses to LDS and through a buffer descriptor behave, and does row_newbcast exist on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k(const uint8_t *in, uint32_t n, uint8_t *lds_dump, uint32_t *buf_out, uint32_t *dpp_out, uint32_t *rd_out) {
  __shared__ __align__(16) uint8_t s[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) s[i] = 0xEE;
  __syncthreads();
  // unaligned dword writes, stride 5 from byte 3
  *reinterpret_cast<uint32_t *>(s + 3 + 5 * lane) = 0x04030201u + 0x10101010u * (uint32_t)(lane & 15);
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) lds_dump[i] = s[i];
  // unaligned dword reads from LDS
  rd_out[lane] = *reinterpret_cast<const uint32_t *>(s + 3 + 5 * lane);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)n, 0x00020000);
  buf_out[lane] = __builtin_amdgcn_raw_buffer_load_b32(rs, 1 + 7 * lane, 0, 0);
  buf_out[64 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rs, n - 2 + lane, 0, 0);   // straddles / beyond the end
  dpp_out[lane] = (uint32_t)__builtin_amdgcn_update_dpp(0, lane, 0x150 + 3, 0xF, 0xF, false);
  dpp_out[64 + lane] = (uint32_t)__builtin_amdgcn_update_dpp(0, lane, 0x150 + 15, 0xF, 0xF, false);
}

int main() {
  const uint32_t n = 1000;
  uint8_t h_in[1000];
  for (uint32_t i = 0; i < n; i++) h_in[i] = (uint8_t)(i * 7 + 1);
  uint8_t *in, *dump; uint32_t *bo, *dp, *rd;
  CK(hipMalloc(&in, 4096)); CK(hipMalloc(&dump, 1024)); CK(hipMalloc(&bo, 512)); CK(hipMalloc(&dp, 512)); CK(hipMalloc(&rd, 256));
  CK(hipMemset(in, 0xAA, 4096));
  CK(hipMemcpy(in, h_in, n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, n, dump, bo, dp, rd);
  CK(hipDeviceSynchronize());
  uint8_t h_dump[1024]; uint32_t h_bo[128], h_dp[128], h_rd[64];
  CK(hipMemcpy(h_dump, dump, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_bo, bo, 512, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h_dp, dp, 512, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_rd, rd, 256, hipMemcpyDeviceToHost));
  int bad_lds = 0, bad_rd = 0, bad_buf = 0, bad_dpp = 0;
  uint8_t ref[1024]; memset(ref, 0xEE, 1024);
  for (int l = 0; l < 64; l++) { uint32_t v = 0x04030201u + 0x10101010u * (uint32_t)(l & 15); memcpy(ref + 3 + 5 * l, &v, 4); }
  for (int i = 0; i < 1024; i++) bad_lds += ref[i] != h_dump[i];
  for (int l = 0; l < 64; l++) { uint32_t v; memcpy(&v, ref + 3 + 5 * l, 4); bad_rd += v != h_rd[l]; }
  for (int l = 0; l < 64; l++) { uint32_t v; memcpy(&v, h_in + 1 + 7 * l, 4); bad_buf += v != h_bo[l]; }
  printf("lds unaligned write: %d bad bytes; lds unaligned read: %d bad; buffer unaligned load: %d bad\n", bad_lds, bad_rd, bad_buf);
  printf("buffer loads at the end (n-2+lane):");
  for (int l = 0; l < 8; l++) printf(" %08x", h_bo[64 + l]);
  printf("\n  bytes at the end: %02x %02x | beyond: AA...\n", h_in[n - 2], h_in[n - 1]);
  for (int l = 0; l < 64; l++) { bad_dpp += h_dp[l] != (uint32_t)((l & ~15) + 3); bad_dpp += h_dp[64 + l] != (uint32_t)((l & ~15) + 15); }
  printf("row_newbcast: %d bad (lane 20 -> %u, lane 63 -> %u)\n", bad_dpp, h_dp[20], h_dp[63]);
  return 0;
}
