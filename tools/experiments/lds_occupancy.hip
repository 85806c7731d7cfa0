// How many workgroups of B threads with X bytes of dynamic LDS does a CU of this GPU hold?  (r06: is the LDS allocated in
// granules that cost k_scan's 54,432-byte workgroups their third slot per CU?)  Measured, not asked: every workgroup
// registers on its CU, spins until a deadline and records the peak number of co-resident workgroups on that CU.
// hipcc --offload-arch=gfx950 -O2 tools/experiments/lds_occupancy.hip -o /tmp/lds_occ && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* cu_now, int* cu_peak, long long spin) {
  extern __shared__ unsigned char lds[];
  if (threadIdx.x == 0) {
    lds[0] = 1;
    const unsigned hw = __builtin_amdgcn_s_getreg(63492);      // HW_ID: cu 8..11, sh 12, se 13..15
    const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 0xf; // XCC_ID
    const int cu = (int)(((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15));
    const int now = atomicAdd(&cu_now[cu], 1) + 1;
    atomicMax(&cu_peak[cu], now);
    const long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    atomicSub(&cu_now[cu], 1);
  }
  __syncthreads();
}
int main() {
  int *now, *peak;
  hipMalloc(&now, 4096 * 4); hipMalloc(&peak, 4096 * 4);
  const int sizes[] = {40960, 52224, 53248, 53760, 54272, 54432, 54528, 54613, 55296, 65536, 81792, 81920, 82048};
  for (int B : {512, 256}) for (int X : sizes) {
    hipMemset(now, 0, 4096 * 4); hipMemset(peak, 0, 4096 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, X);
    hipLaunchKernelGGL(k, dim3(4096), dim3(B), X, 0, now, peak, 2000000LL);
    hipDeviceSynchronize();
    std::vector<int> h(4096); hipMemcpy(h.data(), peak, 4096 * 4, hipMemcpyDeviceToHost);
    int mx = 0, cus = 0; long long sum = 0;
    for (int v : h) if (v) { mx = v > mx ? v : mx; cus++; sum += v; }
    int api = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k, B, X);
    printf("block %3d  LDS %6d B: peak workgroups per CU max %d mean %.2f over %d CUs; occupancy API says %d\n", B, X, mx, cus ? (double)sum / cus : 0.0, cus, api);
  }
  return 0;
}
