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
