// LDS / L1 read-rate microbenchmark for the access patterns of k_scan (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bench tools/lds_bench.hip && /tmp/lds_bench
// Every workgroup fills a 32 KiB LDS buffer, then each wave issues ITER x 16 independent loads of one kind and
// folds the results.  Reported: LDS-pipe clocks per wave-instruction = CUs x clock x time / wave-instructions
// (the pipe is saturated: 16 waves per CU, 16 loads in flight per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LDS_BYTES = 32768;
constexpr int ITER = 256;

enum Kind { U8 = 0, U16, B32, B64, B128, GLB_B64, BFE32, G_ROW4, G_ROW16, G_ROW16x4, G_PIX, G_PIX16 };

template <int KIND>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ addr_tab, int mask_and, int stride_mul,
                                         const unsigned long long* __restrict__ gtab, unsigned* out, const void* __restrict__ gbig = nullptr) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x;
  for (int i = tid; i < LDS_BYTES / 4; i += 256) ((unsigned*)lds)[i] = i * 2654435761u;
  __syncthreads();
  // per-lane base address: table entry (random) or lane * stride
  unsigned a = (KIND == G_ROW4 || KIND == G_ROW16 || KIND == G_ROW16x4) ? (unsigned)((tid >> 6) * 7919 + blockIdx.x * 104729) : stride_mul >= 0 ? (unsigned)((tid & 63) * stride_mul) : addr_tab[blockIdx.x % 64 * 256 + tid];
  unsigned acc = 0;
  const int nl = KIND == G_ROW4 ? 54 : KIND == G_ROW16 ? 14 : KIND == G_ROW16x4 ? 56 : KIND == G_PIX16 ? 16 : 64;
  if ((tid & 63) < nl)
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const unsigned ad = (a + (unsigned)(u * 1237 + it * 4099)) & (unsigned)mask_and;
      if (KIND == U8) acc += lds[ad];
      else if (KIND == U16) acc += *(const unsigned short*)(lds + (ad & ~1u));
      else if (KIND == B32) acc += *(const unsigned*)(lds + (ad & ~3u));
      else if (KIND == BFE32) { const unsigned d = *(const unsigned*)(lds + (ad & ~3u)); acc += (d >> ((ad & 3u) * 8u)) & 0xffu; }
      else if (KIND == B64) { const uint2 v = *(const uint2*)(lds + (ad & ~7u)); acc += v.x ^ v.y; }
      else if (KIND == B128) { const uint4 v = *(const uint4*)(lds + (ad & ~15u)); acc += v.x ^ v.y ^ v.z ^ v.w; }
      else if (KIND == G_ROW4) {           // a 216-byte weight row, 4 bytes per lane, 54 lanes (k_finish regression today)
        const int l = tid & 63;
        acc += ((const unsigned*)gbig)[((ad * 977u) % 4320u) * 54u + l];
      } else if (KIND == G_ROW16) {        // a 224-byte row, 16 bytes per lane, 14 lanes
        const int l = tid & 63;
        { const uint4 v = ((const uint4*)gbig)[((ad * 977u) % 4320u) * 14u + l]; acc += v.x ^ v.y ^ v.z ^ v.w; }
      } else if (KIND == G_ROW16x4) {      // four 224-byte rows per instruction, 16 bytes per lane, 56 lanes
        const int l = tid & 63;
        { const uint4 v = ((const uint4*)gbig)[(((ad + (l / 14) * 131u) * 977u) % 4320u) * 14u + (l % 14)]; acc += v.x ^ v.y ^ v.z ^ v.w; }
      } else if (KIND == G_PIX || KIND == G_PIX16) {   // a byte per lane, random inside a 64x64 window of a 640-wide frame
        const int l = tid & 63;
        const unsigned r = (a * 2654435761u + u * 40503u + it * 7919u);
        const unsigned off = ((r >> 8) & 63u) * 640u + ((r >> 20) & 63u) + (blockIdx.x % 256) * 307200u / 8u;
        acc += ((const unsigned char*)gbig)[off];
      }
      else if (KIND == GLB_B64) { const unsigned long long v = gtab[(ad & 0xfffu) >> 3]; acc += (unsigned)v ^ (unsigned)(v >> 32); }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

static void* g_big = nullptr;

template <int KIND>
static double run(const char* name, const unsigned* d_tab, int mask_and, int stride_mul, const unsigned long long* d_g,
                  unsigned* d_out, int wgs_per_cu) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = 256 * wgs_per_cu * 4;
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d_tab, mask_and, stride_mul, d_g, d_out, (const void*)g_big);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d_tab, mask_and, stride_mul, d_g, d_out, (const void*)g_big);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_instr = (double)grid * 4 * ITER * 16;
  const double clk = 2.4e9;
  const double per = 256.0 * clk * ms * 1e-3 / wave_instr;
  printf("%-46s %8.3f ms  %6.2f CU-clk per wave-instruction (wgs/CU %d)\n", name, ms, per, wgs_per_cu);
  return per;
}

int main() {
  std::vector<unsigned> tab(64 * 256);
  unsigned s = 12345;
  for (auto& v : tab) { s = s * 1664525u + 1013904223u; v = (s >> 8) % LDS_BYTES; }
  unsigned* d_tab; unsigned long long* d_g; unsigned* d_out;
  CHECK(hipMalloc(&d_tab, tab.size() * 4)); CHECK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_g, 4096)); CHECK(hipMemset(d_g, 1, 4096)); CHECK(hipMalloc(&d_out, 64));
  CHECK(hipMalloc(&g_big, 16 << 20)); CHECK(hipMemset(g_big, 3, 16 << 20));
  const int M = LDS_BYTES - 1;
  for (int w : {4}) {
    run<U8>("ds_read_u8   lane*4 (conflict-free)", d_tab, M, 4, d_g, d_out, w);
    run<U8>("ds_read_u8   lane*1 (4 lanes per dword)", d_tab, M, 1, d_g, d_out, w);
    run<U8>("ds_read_u8   lane*5", d_tab, M, 5, d_g, d_out, w);
    run<U8>("ds_read_u8   lane*7", d_tab, M, 7, d_g, d_out, w);
    run<U8>("ds_read_u8   lane*8", d_tab, M, 8, d_g, d_out, w);
    run<U8>("ds_read_u8   random", d_tab, M, -1, d_g, d_out, w);
    run<U8>("ds_read_u8   broadcast (one address)", d_tab, M, 0, d_g, d_out, w);
    run<U16>("ds_read_u16  random", d_tab, M, -1, d_g, d_out, w);
    run<B32>("ds_read_b32  lane*4", d_tab, M, 4, d_g, d_out, w);
    run<B32>("ds_read_b32  lane*8", d_tab, M, 8, d_g, d_out, w);
    run<B32>("ds_read_b32  random", d_tab, M, -1, d_g, d_out, w);
    run<BFE32>("ds_read_b32 + byte extract, random", d_tab, M, -1, d_g, d_out, w);
    run<B32>("ds_read_b32  broadcast", d_tab, M, 0, d_g, d_out, w);
    run<B64>("ds_read_b64  broadcast", d_tab, M, 0, d_g, d_out, w);
    run<B64>("ds_read_b64  lane*8", d_tab, M, 8, d_g, d_out, w);
    run<B64>("ds_read_b64  random in 64 B (node records of a cart)", d_tab, 63, -1, d_g, d_out, w);
    run<B64>("ds_read_b64  random", d_tab, M, -1, d_g, d_out, w);
    run<B128>("ds_read_b128 broadcast", d_tab, M, 0, d_g, d_out, w);
    run<B128>("ds_read_b128 random", d_tab, M, -1, d_g, d_out, w);
    run<GLB_B64>("global_load_dwordx2 one address (L1)", d_tab, M, 0, d_g, d_out, w);
    run<GLB_B64>("global_load_dwordx2 random in 64 B (L1)", d_tab, 63, -1, d_g, d_out, w);
    run<GLB_B64>("global_load_dwordx2 random in 4 KiB (L1)", d_tab, M, -1, d_g, d_out, w);
  }
  run<G_ROW4>("global 216-B row, 4 B x 54 lanes (L2)", d_tab, M, -1, d_g, d_out, 4);
  run<G_ROW16>("global 224-B row, 16 B x 14 lanes (L2)", d_tab, M, -1, d_g, d_out, 4);
  run<G_ROW16x4>("global 4 rows, 16 B x 56 lanes (L2)", d_tab, M, -1, d_g, d_out, 4);
  run<G_PIX>("global byte, 64 lanes random in a 64x64 window", d_tab, M, -1, d_g, d_out, 4);
  run<G_PIX16>("global byte, 16 lanes random in a 64x64 window", d_tab, M, -1, d_g, d_out, 4);
  run<U8>("ds_read_u8   random, 1 wg/CU", d_tab, M, -1, d_g, d_out, 1);
  run<U8>("ds_read_u8   random, 2 wg/CU", d_tab, M, -1, d_g, d_out, 2);
  run<B32>("ds_read_b32  random, 2 wg/CU", d_tab, M, -1, d_g, d_out, 2);
  return 0;
}
