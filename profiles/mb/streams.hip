// Copy of 3.4 GB as many interleaved sequential streams of a few hundred bytes per visit, against the same bytes as one
// ordered sweep: what the memory system makes of the access pattern of the string chain (a row of 16 lanes per stream).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0: every row of 16 lanes owns one contiguous stream of `per_stream` bytes and walks it `visit` bytes at a time.
// MODE 1: the same loads and stores, but the rows of all wavefronts sweep the buffer together (visit v of row r of the
//         grid lies at (v * n_rows + r) * visit): neighbouring rows touch neighbouring memory.
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t per_stream, int visit, size_t n_rows) {
  const size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int rl = threadIdx.x & 15;
  const int vec_per_visit = visit / 16;            // 16-byte vectors per visit
  const size_t visits = per_stream / (size_t)visit;
  for (size_t v = 0; v < visits; v++) {
    const size_t base = MODE == 0 ? (row * per_stream + v * (size_t)visit) / 16 : ((v * n_rows + row) * (size_t)visit) / 16;
    uint4 x[4];
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) { const int i = rl + 16 * k2; x[k2] = i < vec_per_visit ? in[base + i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) { const int i = rl + 16 * k2; if (i < vec_per_visit) out[base + i] = x[k2]; }
  }
}

int main() {
  const size_t n_rows = 24416;                // 1526 pages x 16 ranges
  const int visits_list[] = {256, 576, 1024};
  const size_t total = (size_t)3400 << 20;
  uint4 *in, *out;
  CK(hipMalloc(&in, total + (1 << 20))); CK(hipMalloc(&out, total + (1 << 20)));
  CK(hipMemset(in, 1, total)); CK(hipMemset(out, 0, total));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int visit : visits_list) {
    const size_t per_stream = total / n_rows / (size_t)visit * (size_t)visit;
    for (int mode = 0; mode < 2; mode++) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((unsigned)(n_rows / 16)), dim3(256), 0, 0, in, out, per_stream, visit, n_rows);
        else hipLaunchKernelGGL(k<1>, dim3((unsigned)(n_rows / 16)), dim3(256), 0, 0, in, out, per_stream, visit, n_rows);
        CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      const double bytes = 2.0 * (double)per_stream * (double)n_rows;
      printf("visit %4d B, %s: %.3f ms, %.2f TB/s read + written\n", visit, mode == 0 ? "24416 streams, one per row " : "the rows sweep side by side", best,
             bytes / best / 1e9);
    }
  }
  return 0;
}
