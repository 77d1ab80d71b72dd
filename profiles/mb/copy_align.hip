#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct V16U { uint32_t v[4]; } __attribute__((packed));
__global__ __launch_bounds__(256) void k_copy(const uint8_t *src, uint8_t *dst, int64_t page, int npages) {
  const int d = blockIdx.x;
  const uint8_t *s = src + (int64_t)d * page; uint8_t *o = dst + (int64_t)d * page;
  for (int64_t b = ((int64_t)blockIdx.y * 256 + threadIdx.x) * 16; b + 16 <= page; b += (int64_t)gridDim.y * 256 * 16) {
    const V16U x = *reinterpret_cast<const V16U *>(s + b);
    *reinterpret_cast<V16U *>(o + b) = x;
  }
}
// aligned loads, data of the next lane by DPP, funnel shift
__global__ __launch_bounds__(256) void k_copy_sh(const uint8_t *src, uint8_t *dst, int64_t page, int npages) {
  const int d = blockIdx.x, lane = threadIdx.x & 63;
  const uint8_t *s = src + (int64_t)d * page; uint8_t *o = dst + (int64_t)d * page;
  const uint32_t k = (uint32_t)(reinterpret_cast<uintptr_t>(s) & 15u);
  const uint8_t *sa = s - k;
  const uint32_t q = k >> 2, r = (k & 3u) * 8u;
  for (int64_t b = ((int64_t)blockIdx.y * 256 + threadIdx.x) * 16; b + 16 <= page; b += (int64_t)gridDim.y * 256 * 16) {
    const uint4 a = *reinterpret_cast<const uint4 *>(sa + b);
    uint32_t D[8] = {a.x, a.y, a.z, a.w, 0, 0, 0, 0};
    // next lane's vector (wave_shl:1); lane 63 loads it
    D[4] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.x, 0x130, 0xF, 0xF, false);
    D[5] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.y, 0x130, 0xF, 0xF, false);
    D[6] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.z, 0x130, 0xF, 0xF, false);
    D[7] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.w, 0x130, 0xF, 0xF, false);
    if (lane == 63) { const uint4 n = *reinterpret_cast<const uint4 *>(sa + b + 16); D[4] = n.x; D[5] = n.y; D[6] = n.z; D[7] = n.w; }
    uint32_t E[5];
    switch (q) {
      case 0: E[0]=D[0];E[1]=D[1];E[2]=D[2];E[3]=D[3];E[4]=D[4];break;
      case 1: E[0]=D[1];E[1]=D[2];E[2]=D[3];E[3]=D[4];E[4]=D[5];break;
      case 2: E[0]=D[2];E[1]=D[3];E[2]=D[4];E[3]=D[5];E[4]=D[6];break;
      default: E[0]=D[3];E[1]=D[4];E[2]=D[5];E[3]=D[6];E[4]=D[7];break;
    }
    uint4 y;
    y.x = __builtin_amdgcn_alignbit(E[1], E[0], r); y.y = __builtin_amdgcn_alignbit(E[2], E[1], r);
    y.z = __builtin_amdgcn_alignbit(E[3], E[2], r); y.w = __builtin_amdgcn_alignbit(E[4], E[3], r);
    *reinterpret_cast<V16U *>(o + b) = V16U{{y.x, y.y, y.z, y.w}};
  }
}
int main() {
  const int np = 1526; const int64_t page = 2359296;   // 2.25 MiB
  const int64_t n = np * page;
  uint8_t *a, *b; hipMalloc(&a, n + 256); hipMalloc(&b, n + 256);
  hipMemset(a, 1, n + 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int so = 0; so <= 7; so += 7) for (int dof = 0; dof <= 5; dof += 5) for (int var = 0; var < 2; var++) {
    float best = 1e9;
    for (int it = 0; it < 5; it++) {
      hipEventRecord(e0);
      if (var == 0) hipLaunchKernelGGL(k_copy, dim3(np, 32), dim3(256), 0, 0, a + so, b + dof, page, np);
      else hipLaunchKernelGGL(k_copy_sh, dim3(np, 32), dim3(256), 0, 0, a + 16 + so, b + dof, page, np);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("src+%d dst+%d %s: %.3f ms  %.2f TB/s\n", so, dof, var ? "aligned loads + shift" : "plain", best, 2.0 * n / best / 1e9);
  }
  float best = 1e9;
  for (int it = 0; it < 5; it++) { hipEventRecord(e0); hipMemcpyAsync(b, a, n, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  printf("hipMemcpy D2D: %.3f ms %.2f TB/s\n", best, 2.0 * n / best / 1e9);
  return 0;
}
