// What a per-level planar (x-de-interleaved) copy of a 256x640x480 batch costs: the pre-pass of the k_scan variant that
// loads de-interleaved tiles with plain LDS-DMA.  plane p of a row holds the pixels x = i*s + p, so
// the x-neighbouring windows of a level with step s read consecutive bytes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/planar_bench tools/planar_bench.hip && /tmp/planar_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one wave per row, four rows per workgroup: the row goes through LDS, every lane writes one dword (4 plane entries)
__global__ __launch_bounds__(256) void k_planar(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int rows,
                                                int s, int plane_w) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[4][1024];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wv;
  if (row >= rows) return;
  const uint4* g = (const uint4*)(src + (size_t)row * W);
  for (int j = lane; j < (W >> 4); j += 64) ((uint4*)tile[wv])[j] = g[j];
  __builtin_amdgcn_wave_barrier();
  uint32_t* d = (uint32_t*)(dst + (size_t)row * s * plane_w);
  const int dw_per_plane = plane_w >> 2, total = s * dw_per_plane;
  for (int j = lane; j < total; j += 64) {
    const int p = j / dw_per_plane, i = (j - p * dw_per_plane) << 2;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int x = (i + k) * s + p;
      v |= (uint32_t)(x < W ? tile[wv][x] : 0) << (8 * k);
    }
    d[j] = v;
  }
}

int main() {
  const int N = 256, W = 640, H = 480, R = 4;           // four rotating batches like the bench (314 MB: past the Infinity Cache)
  const size_t fbytes = (size_t)N * W * H;
  uint8_t* src; CK(hipMalloc(&src, fbytes * R));
  CK(hipMemset(src, 7, fbytes * R));
  const int steps[3] = {5, 7, 8};                        // levels 57 / 71 / 88 px of the canonical call (46 px: step 4, conflict-free as it is)
  uint8_t* dst[3]; int pw[3];
  for (int l = 0; l < 3; l++) { pw[l] = (((W + steps[l] - 1) / steps[l]) + 15) & ~15; CK(hipMalloc(&dst[l], (size_t)N * H * steps[l] * pw[l])); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rows = N * H;
  for (int rep = 0; rep < 3; rep++) {
    float tot = 0;
    for (int l = 0; l < 3; l++) {
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < 8; it++)
        hipLaunchKernelGGL(k_planar, dim3((rows + 3) / 4), dim3(256), 0, 0, src + (size_t)(it % R) * fbytes, dst[l], W, rows, steps[l], pw[l]);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
      const double bytes = (double)fbytes + (double)N * H * steps[l] * pw[l];
      if (rep == 2) printf("step %d: plane width %d, %.1f us per batch, %.2f TB/s (read + write)\n", steps[l], pw[l], ms * 1e3, bytes / ms / 1e9);
      tot += ms;
    }
    if (rep == 2) printf("three levels: %.1f us per 256x640x480 batch\n", tot * 1e3);
  }
  return 0;
}
