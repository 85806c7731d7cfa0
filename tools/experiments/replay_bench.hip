// Score-replay microbenchmark (gfx950): shader clocks per cart of the forms of kernels_common.h:replay_scores, for a
// wave that has its SIMD to itself (k_finish_wide: wave 0 replays while the others wait) and with 16 one-wave
// workgroups per CU (k_finish: four waves per SIMD, all replaying).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/replay_bench tools/experiments/replay_bench.hip && /tmp/replay_bench
//   A  systolic chain over the lanes: one v_add_f32_dpp wave_shr:1 per cart (the product's form)
//   B  wave-uniform running score: v_readlane (leaf score of cart j) + v_add + v_cmp against every lane's threshold,
//      bit j of the compare kept with scalar ops
//   C  B without the per-cart compare (lower bound of a uniform chain: readlane + add)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REPS = 256;         // groups of 64 carts per measurement

template <int FORM>
__global__ __launch_bounds__(64) void k(const float* __restrict__ ls_tab, const float* __restrict__ th_tab, float* out,
                                        unsigned long long* clocks) {
  const int lane = threadIdx.x;
  float score = 0.25f;
  unsigned long long rej_all = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REPS; r++) {
    const float ls = ls_tab[(r * 64 + lane) & 4095];
    const float th = th_tab[(r * 64 + lane) & 4095];
    if (FORM == 0) {
      float pf = score + ls;
#pragma unroll
      for (int i = 0; i < 63; i++)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pf) : "v"(ls));
      const unsigned long long rej = __ballot(pf < th);
      rej_all |= rej;
      score = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pf), 63));
    } else {
      float acc = score;
      unsigned long long rej = 0;
#pragma unroll
      for (int j = 0; j < 64; j++) {
        acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ls), j));
        if (FORM == 1) rej |= __ballot(acc < th) & (1ull << j);
      }
      if (FORM == 2) rej = __ballot(acc < th);
      rej_all |= rej;
      score = acc;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[blockIdx.x * 2] = score; out[blockIdx.x * 2 + 1] = (float)(unsigned)(rej_all ^ (rej_all >> 32)); clocks[blockIdx.x] = t1 - t0; }
}

int main() {
  std::vector<float> ls(4096), th(4096);
  srand(1);
  for (int i = 0; i < 4096; i++) { ls[i] = (float)((rand() % 2001) - 1000) / 1024.f; th[i] = -3.f + (float)(rand() % 100) / 64.f; }
  float *d_ls, *d_th, *d_out; unsigned long long* d_clk;
  const int max_blocks = 256 * 16;
  CHECK(hipMalloc(&d_ls, 4096 * 4)); CHECK(hipMalloc(&d_th, 4096 * 4)); CHECK(hipMalloc(&d_out, max_blocks * 8)); CHECK(hipMalloc(&d_clk, max_blocks * 8));
  CHECK(hipMemcpy(d_ls, ls.data(), 4096 * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_th, th.data(), 4096 * 4, hipMemcpyHostToDevice));
  const char* names[3] = {"A dpp systolic", "B readlane+add+cmp", "C readlane+add"};
  for (int blocks : {256, 256 * 16}) {
    for (int form = 0; form < 3; form++) {
      std::vector<unsigned long long> clk(blocks);
      std::vector<float> out(blocks * 2);
      for (int rep = 0; rep < 2; rep++) {
        if (form == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d_ls, d_th, d_out, d_clk);
        if (form == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d_ls, d_th, d_out, d_clk);
        if (form == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d_ls, d_th, d_out, d_clk);
        CHECK(hipDeviceSynchronize());
      }
      CHECK(hipMemcpy(clk.data(), d_clk, blocks * 8, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(out.data(), d_out, blocks * 8, hipMemcpyDeviceToHost));
      double s = 0; for (auto c : clk) s += (double)c;
      // s_memtime counts at 100 MHz on this part (10 ns per tick); shader clock 2.4 GHz
      const double ns = s / blocks / (REPS * 64.0) * 10.0;
      printf("%-5d workgroups  %-22s %7.2f ns per cart = %6.1f shader clocks   (score %.6f rej %.0f)\n", blocks, names[form], ns, ns * 2.4, out[0], out[1]);
    }
  }
  return 0;
}
