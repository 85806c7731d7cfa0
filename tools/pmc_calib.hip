// Calibration kernels for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950
// (guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 of a 16 B/lane streaming
// read; other widths are uncalibrated).  k_scan stages its pixel tiles with
// 4 B/lane row loads, so that pattern is measured here on a buffer far larger
// than the 256 MiB Infinity Cache.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void calib_read4(const uint32_t* __restrict__ p, size_t n, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_read16(const uint4* __restrict__ p, size_t n, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_write4(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

// k_finish's regression gather (BASELINE.json configs[4]: rows of 136 floats = 544 bytes, K rows per window and stage):
// every wave gathers `rows_per_wave` pseudo-random rows of `row_floats` floats from a table of n_rows rows, lanes =
// coordinates (4 bytes per lane, 64 + 64 + 8 lanes per row), 32 rows in flight like the kernel.  Useful bytes per
// dispatch = waves * rows_per_wave * row_floats * 4.
__global__ __launch_bounds__(64) void calib_gather_rows(const float* __restrict__ tab, unsigned n_rows, int row_floats,
                                                        int rows_per_wave, float* out) {
  const int lane = threadIdx.x;
  unsigned h = (blockIdx.x + 1u) * 2654435761u;
  float acc = 0.f;
  for (int k0 = 0; k0 < rows_per_wave; k0 += 32) {
    unsigned row[32];
#pragma unroll
    for (int u = 0; u < 32; u++) { h = h * 1664525u + 1013904223u; row[u] = (h >> 8) % n_rows; }
    for (int d = lane; d < row_floats; d += 64) {
      float r[32];
#pragma unroll
      for (int u = 0; u < 32; u++) r[u] = tab[(size_t)row[u] * row_floats + d];
#pragma unroll
      for (int u = 0; u < 32; u++) acc += r[u];
    }
  }
  if (acc == 1.2345678f) out[0] = acc;
}

// tab_bytes: size of the row table; waves x rows_per_wave rows of row_floats floats are gathered per dispatch
extern "C" int pmc_calib_gather(size_t tab_bytes, int row_floats, unsigned waves, int rows_per_wave) {
  float *tab = nullptr, *out = nullptr;
  if (hipMalloc(&tab, tab_bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return -1;
  hipMemset(tab, 0, tab_bytes);
  hipDeviceSynchronize();
  const unsigned n_rows = (unsigned)(tab_bytes / ((size_t)row_floats * 4));
  for (int rep = 0; rep < 2; rep++)
    hipLaunchKernelGGL(calib_gather_rows, dim3(waves), dim3(64), 0, 0, tab, n_rows, row_floats, rows_per_wave, out);
  hipDeviceSynchronize();
  hipFree(tab); hipFree(out);
  return 0;
}

extern "C" int pmc_calib_run(size_t bytes) {
  uint32_t *buf = nullptr, *out = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return -1;
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(calib_read4, dim3(4096), dim3(256), 0, 0, buf, bytes / 4, out);
    hipLaunchKernelGGL(calib_read16, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(calib_write4, dim3(4096), dim3(256), 0, 0, buf, bytes / 4);
  }
  hipDeviceSynchronize();
  hipFree(buf); hipFree(out);
  return 0;
}
