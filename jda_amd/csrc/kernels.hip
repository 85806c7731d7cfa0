// HIP kernels of the JDA detect path for gfx950 (MI355X, wave64).
//
// Built with -ffp-contract=off -fno-gpu-flush-denormals-to-zero: every fp
// operation below must round exactly like the reference's scalar C/C++ (no FMA,
// IEEE division, denormals kept).
//
//   k_resize        bilinear pyramid image        reference c/jda.c:203-230
//   k_resize_cv     cv::resize(INTER_LINEAR) restated (dialect CPP pyramids, cascador.cpp:302,330-331)
//   k_prep_stage0   stage-0 feature offsets per level (hoisted c/jda.c:370-389)
//   k_scan          first `handoff` carts of stage 0: LDS pixel tile, lane = window,
//                   survivors compacted by ballot/prefix-sum every chunk of carts; late
//                   phases spread (window, cart) pairs over all lanes and replay the scores
//                   16 carts at a time in registers      reference c/jda.c:357-402
//   k_enqueue       windows of levels k_scan does not cover -> hand-off queue at cart 0
//   k_finish        wave = window, for every survivor: lanes = carts for a stage's
//                   tree walks (c/jda.c:366-400, cart.cpp:392-404), the score
//                   recurrence replayed in cart order (c/jda.c:395-399), then
//                   lanes = shape coordinates for the regression gather in cart
//                   order (c/jda.c:404-411, btcart.cpp:407-424), final cut
//                   (c/jda.c:414) and emit
//   k_stage         dense mode: one whole stage for a 16x16 tile of windows per workgroup,
//                   lane = window, tables and weight rows of a chunk of carts shared in LDS
//                   (same references as k_finish)
//   k_trace_fill    per-window trace defaults (parity instrumentation)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <type_traits>

#include "kernels.h"

namespace jda {

namespace {

constexpr unsigned kFnvSeed = 2166136261u;
__device__ __forceinline__ unsigned fnv_step(unsigned h, int v) { return (h ^ (unsigned)v) * 16777619u; }

// float/double -> int the way the reference build does it (x86 cvttss2si /
// cvttsd2si: truncation, and INT_MIN for NaN or out-of-range values; the GPU
// conversion saturates instead).
__device__ __forceinline__ int to_int_x86(float v) {
  const int r = (int)v;
  return (fabsf(v) < 2147483648.f) ? r : INT_MIN;
}
__device__ __forceinline__ int to_int_x86(double v) {
  const int r = (int)v;
  return (v > -2147483649.0 && v < 2147483648.0) ? r : INT_MIN;
}

__device__ __forceinline__ int clamp_win(int v, int win) { return v < 0 ? 0 : (v >= win ? win - 1 : v); }

// ---- numeric dialects ---------------------------------------------------------

struct DialectC {           // reference c/jda.c
  using Real = float;
  using Node = NodeF;
  // c/jda.c:373-381: fp32 add, fp32 multiply by the window side, truncate
  static __device__ __forceinline__ int coord(float s, float o, int win) {
    const float v = (s + o) * (float)win;
    return to_int_x86(v);
  }
  // clamp_win(coord(s, o, win), win) in 4 instructions instead of 8 (the walks are VALU bound).
  // v_cvt_i32_f32 saturates: NaN -> 0, v >= 2^31 -> INT_MAX, v <= -2^31 -> INT_MIN, where x86
  // gives INT_MIN for all three, which the clamp then turns into 0.  No float below 2^31
  // converts to INT_MAX (the largest is 2^31 - 128), so INT_MAX marks exactly the positive
  // overflow; r + 1 wraps it to INT_MIN, and median(r + 1, 1, win) - 1 is the clamped pixel:
  // NaN -> 0, either overflow -> 0, r < 0 -> 0, r >= win -> win - 1.
  static __device__ __forceinline__ int pixel(float s, float o, int win) {
    const float v = (s + o) * (float)win;
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    const int r1 = (int)((unsigned)r + 1u);
    int m;
    asm("v_med3_i32 %0, %1, 1, %2" : "=v"(m) : "v"(r1), "v"(win));
    return m - 1;
  }
  // The same for an (x, y) pair, returned 1-BASED (kBias): the caller folds the -1s into its
  // tile base address.  The add and the multiply are packed (two independent IEEE operations).
  static constexpr int kBias = 1;
  static __device__ __forceinline__ void pixel_pair(float sx, float sy, float ox, float oy, int win, int* x, int* y) {
    typedef float V2 __attribute__((ext_vector_type(2)));
    const float fw = (float)win;
    const V2 v = (V2{sx, sy} + V2{ox, oy}) * V2{fw, fw};
    int rx, ry;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(rx) : "v"(v.x));
    asm("v_cvt_i32_f32 %0, %1" : "=v"(ry) : "v"(v.y));
    const int rx1 = (int)((unsigned)rx + 1u), ry1 = (int)((unsigned)ry + 1u);
    asm("v_med3_i32 %0, %1, 1, %2" : "=v"(*x) : "v"(rx1), "v"(win));
    asm("v_med3_i32 %0, %1, 1, %2" : "=v"(*y) : "v"(ry1), "v"(win));
  }
};

struct DialectCPP {         // reference src/jda (Validate / CalcFeatureValue)
  using Real = double;
  using Node = NodeD;
  // data.cpp:40-47: fp64, round half away from zero
  static __device__ __forceinline__ int coord(double s, double o, int win) {
    const double v = (s + o) * (double)win;
    return to_int_x86(round(v));
  }
  static __device__ __forceinline__ int pixel(double s, double o, int win) { return clamp_win(coord(s, o, win), win); }
  static constexpr int kBias = 0;
  static __device__ __forceinline__ void pixel_pair(double sx, double sy, double ox, double oy, int win, int* x, int* y) {
    *x = clamp_win(coord(sx, ox, win), win); *y = clamp_win(coord(sy, oy, win), win);
  }
};

__device__ __forceinline__ int wave_lane() { return threadIdx.x & 63; }

// value of lane j (wave-uniform j), for any lane mask
__device__ __forceinline__ int rl(int v, int j) { return __builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ float rl(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ double rl(double v, int j) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ unsigned long long lanes_below(int lane) {
  return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

__device__ __forceinline__ unsigned long long* shard_counter(unsigned long long* counters, int idx) {
  return counters + (size_t)(blockIdx.x % kCntShards) * kCntStride + idx;
}

}  // namespace

// =============================================================================
// pyramid resize
// =============================================================================

__global__ void k_resize(const uint8_t* __restrict__ src, size_t src_stride, int sw, int sh,
                         uint8_t* __restrict__ dst, size_t dst_stride, int dw, int dh,
                         float rx, float ry) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  const int f = blockIdx.z;
  if (j >= dw || i >= dh) return;
  const uint8_t* s = src + (size_t)f * src_stride;
  // c/jda.c:215-226, operation for operation
  const float fx = rx * (float)j;
  const float fy = ry * (float)i;
  const int x = (int)fx;
  const int y = (int)fy;
  const float xd = fx - (float)x;
  const float yd = fy - (float)y;
  const int idx = y * sw + x;
  const float a = (float)(int)s[idx], b = (float)(int)s[idx + 1];
  const float c = (float)(int)s[idx + sw], d = (float)(int)s[idx + sw + 1];
  const float one_x = 1.f - xd, one_y = 1.f - yd;
  float v = a * one_x * one_y;
  v = v + b * xd * one_y;
  v = v + c * one_x * yd;
  v = v + d * xd * yd;
  dst[(size_t)f * dst_stride + (size_t)i * dw + j] = (uint8_t)(int)v;
}

hipError_t launch_resize(const uint8_t* src, size_t src_stride, int n, int sw, int sh,
                         uint8_t* dst, size_t dst_stride, int dw, int dh, float rx, float ry,
                         hipStream_t stream) {
  if (dw <= 0 || dh <= 0 || n <= 0) return hipSuccess;
  dim3 block(256), grid((dw + 255) / 256, dh, n);
  hipLaunchKernelGGL(k_resize, grid, block, 0, stream, src, src_stride, sw, sh, dst, dst_stride, dw, dh, rx, ry);
  return hipGetLastError();
}

// cv::resize(INTER_LINEAR) for 8-bit single-channel images as dialect CPP uses it for the
// half/quarter images (cascador.cpp:329-331) and the method-0 pyramid (cascador.cpp:300-303):
// 11-bit fixed-point bilinear of OpenCV's 2.4/3.x imgwarp.cpp, with its routing of an exact
// 2x2 down-scale to the box average.  PARITY UNPINNED (no OpenCV here to compare with);
// bit-exact against the oracle's restatement of the same algorithm.
__global__ void k_resize_cv(const uint8_t* __restrict__ src, size_t src_stride, int sw, int sh,
                            uint8_t* __restrict__ dst, size_t dst_stride, int dw, int dh,
                            double scale_x, double scale_y, int area_fast) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  const int dy = blockIdx.y;
  const int f = blockIdx.z;
  if (dx >= dw || dy >= dh) return;
  const uint8_t* s = src + (size_t)f * src_stride;
  uint8_t* d = dst + (size_t)f * dst_stride;
  if (area_fast) {
    const uint8_t* p = s + (size_t)(2 * dy) * sw + 2 * dx;
    d[(size_t)dy * dw + dx] = (uint8_t)((p[0] + p[1] + p[sw] + p[sw + 1] + 2) >> 2);
    return;
  }
  float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  const bool edge = sx + 1 >= sw;            // dx >= xmax in OpenCV's loop
  if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
  float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
  const int sy = (int)floorf(fy);
  fy -= (float)sy;
  auto sat_short = [](float v) { int i = __float2int_rn(v); return i < -32768 ? -32768 : (i > 32767 ? 32767 : i); };
  const int a0 = sat_short((1.f - fx) * 2048.f), a1 = sat_short(fx * 2048.f);
  const int b0 = sat_short((1.f - fy) * 2048.f), b1 = sat_short(fy * 2048.f);
  const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
  const uint8_t* S0 = s + (size_t)y0 * sw;
  const uint8_t* S1 = s + (size_t)y1 * sw;
  int r0, r1;
  if (!edge) { r0 = S0[sx] * a0 + S0[sx + 1] * a1; r1 = S1[sx] * a0 + S1[sx + 1] * a1; }
  else { r0 = S0[sx] * 2048; r1 = S1[sx] * 2048; }
  d[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

hipError_t launch_resize_cv(const uint8_t* src, size_t src_stride, int n, int sw, int sh,
                            uint8_t* dst, size_t dst_stride, int dw, int dh, hipStream_t stream) {
  if (dw <= 0 || dh <= 0 || n <= 0) return hipSuccess;
  const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
  const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
  const int area = (fabs(scale_x - 2.) < 2.220446049250313e-16 && fabs(scale_y - 2.) < 2.220446049250313e-16) ? 1 : 0;
  dim3 block(256), grid((dw + 255) / 256, dh, n);
  hipLaunchKernelGGL(k_resize_cv, grid, block, 0, stream, src, src_stride, sw, sh, dst, dst_stride, dw, dh,
                     scale_x, scale_y, area);
  return hipGetLastError();
}

// =============================================================================
// stage-0 offset table
// =============================================================================

template <typename DL>
__global__ void k_prep_stage0(const DevPlan* __restrict__ plan, const typename DL::Node* __restrict__ nodes,
                              const typename DL::Real* __restrict__ mean_shape, int K, int node_n,
                              S0Node* __restrict__ table) {
  const int l = blockIdx.y;
  const DevLevel lv = plan->lv[l];
  if (!lv.tiled) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * node_n) return;
  const typename DL::Node nd = nodes[i];  // stage 0 occupies the first K*node_n nodes
  const int win = lv.win;
  int x1 = clamp_win(DL::coord(mean_shape[nd.lm1x2], nd.o1x, win), win);
  int y1 = clamp_win(DL::coord(mean_shape[nd.lm1x2 + 1], nd.o1y, win), win);
  int x2 = clamp_win(DL::coord(mean_shape[nd.lm2x2], nd.o2x, win), win);
  int y2 = clamp_win(DL::coord(mean_shape[nd.lm2x2 + 1], nd.o2y, win), win);
  // feature is a difference of two bytes: thresholds beyond [-256,255] behave like the ends
  const int th = nd.th < -256 ? -256 : (nd.th > 255 ? 255 : nd.th);
  S0Node o;
  if (lv.tiled == 1) {
    o.lo = (uint32_t)(y1 * lv.pitch + x1) | ((uint32_t)(y2 * lv.pitch + x2) << 16);
    o.hi = (uint32_t)th;
  } else {
    const unsigned long long v = (unsigned long long)(uint32_t)(y1 * lv.pitch + x1) |
                                 ((unsigned long long)(uint32_t)(y2 * lv.pitch + x2) << kS0GlobalOffBits) |
                                 ((unsigned long long)(uint32_t)(th + 256) << (2 * kS0GlobalOffBits));
    o.lo = (uint32_t)v; o.hi = (uint32_t)(v >> 32);
  }
  table[lv.s0_table + i] = o;
}

hipError_t launch_prep_stage0(int dialect, const DevPlan* d_plan, const DevPlan& h_plan,
                              const void* nodes, const void* mean_shape, int K, int node_n,
                              S0Node* table, hipStream_t stream) {
  dim3 block(256), grid((K * node_n + 255) / 256, h_plan.n_levels);
  if (dialect == 0)
    hipLaunchKernelGGL(k_prep_stage0<DialectC>, grid, block, 0, stream, d_plan, (const NodeF*)nodes,
                       (const float*)mean_shape, K, node_n, table);
  else
    hipLaunchKernelGGL(k_prep_stage0<DialectCPP>, grid, block, 0, stream, d_plan, (const NodeD*)nodes,
                       (const double*)mean_shape, K, node_n, table);
  return hipGetLastError();
}

// =============================================================================
// stage-0 scan
// =============================================================================

// Carts of stage 0 that k_scan evaluates before handing survivors to k_finish:
// the tables of all of them (nodes, leaf scores, cart parameters) are staged in
// LDS once per workgroup, so the count is capped by a 16 KiB table budget.
int scan_handoff_cap(int node_n, int leaf_n, int real_bytes) {
  const int per_cart = node_n * (int)sizeof(S0Node) + leaf_n * real_bytes + 4 * real_bytes;
  int c = (16 * 1024) / per_cart;
  if (c < 8) c = 8;
  return c & ~7;
}

namespace {

template <typename Real>
struct CartPar {          // per-cart parameters as k_scan reads them from LDS
  Real th, mean, std;
  Real norm;              // != 0 where (mean,std) != (0,1)
};

template <typename Real, bool TRACE>
struct ScanLds {
  // byte offsets inside dynamic LDS
  int pix, nodes, leaf, par, q_widx, q_score, q_hash, lfbuf, misc, total;
  __host__ __device__ ScanLds(int pix_bytes, int carts, int node_n, int leaf_n, int m_max) {
    int o = 0;
    pix = o; o += (pix_bytes + 15) & ~15;
    nodes = o; o += carts * node_n * (int)sizeof(S0Node); o = (o + 15) & ~15;
    leaf = o; o += carts * leaf_n * (int)sizeof(Real); o = (o + 15) & ~15;
    par = o; o += carts * (int)sizeof(CartPar<Real>);
    q_score = o; o += 2 * m_max * (int)sizeof(Real);
    q_widx = o; o += 2 * m_max * 2; o = (o + 15) & ~15;
    q_hash = o; if (TRACE) o += 2 * m_max * 4;
    lfbuf = o; o += 2048;               // leaf indices [window][cart of the round], n_pad x (2048 / n_pad) (late phases)
    misc = o; o += 64;
    total = o;
  }
};

}  // namespace

size_t scan_lds_bytes(int pix_bytes, int carts, int node_n, int leaf_n, int real_bytes, bool trace) {
  if (real_bytes == 4) return trace ? ScanLds<float, true>(pix_bytes, carts, node_n, leaf_n, 512).total
                                    : ScanLds<float, false>(pix_bytes, carts, node_n, leaf_n, 512).total;
  return trace ? ScanLds<double, true>(pix_bytes, carts, node_n, leaf_n, 512).total
               : ScanLds<double, false>(pix_bytes, carts, node_n, leaf_n, 512).total;
}

// Feature test of one resolved stage-0 node for the window at `base`: true = go left
// (feature <= threshold, c/jda.c:391-393).
template <bool GLB>
__device__ __forceinline__ bool s0_left(const S0Node r, const uint8_t* __restrict__ pix, int base) {
  if (!GLB) {
    const int a = pix[base + (int)(r.lo & 0xffffu)];
    const int b = pix[base + (int)(r.lo >> 16)];
    return a - b <= (int)r.hi;
  }
  const uint32_t o1 = r.lo & 0x1fffffu;
  const uint32_t o2 = __builtin_amdgcn_alignbit(r.hi, r.lo, 21) & 0x1fffffu;
  const int a = pix[base + (int)o1];
  const int b = pix[base + (int)o2];
  return a - b + 256 <= (int)(r.hi >> 10);
}

// One cart of stage 0 for one window -> node index reached below the last split level:
// D-1 dependent (node record, 2 pixels) reads.  The node table is in LDS; pixels come
// from the LDS tile (GLB = false) or from the frame through L1/L2 (GLB = true).
// (Testing the root and both children at once -- 2 dependent round trips instead of 3,
// 8 pixel reads instead of 6 -- was measured 15 % SLOWER: the kernel is sensitive to LDS
// instruction count, see DESIGN.md.)
template <int DEPTH, bool GLB>
__device__ __forceinline__ int scan_tree(const S0Node* __restrict__ tbl, const uint8_t* __restrict__ pix,
                                         int base, int depth_rt) {
  int node = 0;
  const int levels = DEPTH > 0 ? DEPTH - 1 : depth_rt - 1;
#pragma unroll
  for (int d = 0; d < levels; d++) {
    const S0Node r = tbl[node];
    node = 2 * node + (s0_left<GLB>(r, pix, base) ? 1 : 2);
  }
  return node;
}

// global -> LDS copy of n elements by the whole workgroup with all loads of a
// thread in flight before its first store (one memory latency, not one per element)
template <typename T, int BLOCK, int UNROLL>
__device__ __forceinline__ void stage_to_lds(T* __restrict__ dst, const T* __restrict__ src, int n, int tid) {
  for (int i0 = 0; i0 < n; i0 += BLOCK * UNROLL) {
    T v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i = i0 + u * BLOCK + tid;
      if (i < n) v[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i = i0 + u * BLOCK + tid;
      if (i < n) dst[i] = v[u];
    }
  }
}

// Stages the pw x ph pixel tile whose origin in the frame is (x0, y0) into LDS at row pitch
// `pitch` (a multiple of 16).  Returns the x offset of the tile origin inside the LDS rows.
//  * frame 16-byte aligned (base, stride, width, x0): LDS-DMA -- `global_load_lds_dwordx4` moves
//    16 bytes per lane straight from the frame into LDS (no VGPR round trip); a wave instruction
//    fills 1 KiB of consecutive LDS = 64 consecutive chunks of the row-major tile; all of a
//    wave's loads are in flight together.  The CALLER waits vmcnt(0) before its barrier.
//  * 4-byte aligned: coalesced dword rows, 8 rows in flight per wave;  * else: byte rows.
template <int BLOCK>
__device__ __forceinline__ int load_tile(unsigned char* lds_pix, const uint8_t* frames, size_t frame_stride,
                                         const uint8_t* img, int W, int x0, int y0, int pw, int ph, int pitch,
                                         int tid) {
  constexpr int NW = BLOCK / 64;
  const int lane = tid & 63, wv = tid >> 6;
  if (((W & 15) | (x0 & 15) | (int)(frame_stride & 15) | (int)(((uintptr_t)frames) & 15)) == 0) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    const int cpr = pitch >> 4;                          // chunks per tile row
    const int nchunks = ph * cpr;
    const int maxcol = ((W - x0) >> 4) - 1;              // last chunk that ends inside the frame row
    int i = wv * 64 + lane;
    int row = i / cpr, col = i - row * cpr;
    const int dr = BLOCK / cpr, dc = BLOCK - dr * cpr;
    const uint8_t* g0 = img + (size_t)y0 * W + x0;
    for (int base = wv * 64; base < nchunks; base += BLOCK) {
      if (i < nchunks) {
        const uint8_t* g = g0 + (size_t)row * W + (min(col, maxcol) << 4);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(lds_pix + (base << 4)), 16, 0, 0);
      }
      i += BLOCK; row += dr; col += dc;
      if (col >= cpr) { col -= cpr; row++; }
    }
    return 0;
  }
  const bool al4 = ((W & 3) == 0) && ((frame_stride & 3) == 0) && ((((uintptr_t)frames) & 3) == 0);
  if (al4) {
    const int x0a = x0 & ~3, xshift = x0 - x0a;
    const int ndw = (xshift + pw + 3) >> 2;
    const int w4 = W >> 2, p4 = pitch >> 2;
    const uint32_t* g = (const uint32_t*)(img + (size_t)y0 * W + x0a);
    uint32_t* d = (uint32_t*)lds_pix;
    for (int c0 = 0; c0 < ndw; c0 += 64) {
      const int c = c0 + lane;
      const bool cok = c < ndw;
      for (int r0 = wv * 8; r0 < ph; r0 += NW * 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) if (cok && r0 + u < ph) v[u] = g[(size_t)(r0 + u) * w4 + c];
#pragma unroll
        for (int u = 0; u < 8; u++) if (cok && r0 + u < ph) d[(r0 + u) * p4 + c] = v[u];
      }
    }
    return xshift;
  }
  const uint8_t* g = img + (size_t)y0 * W + x0;
  for (int c0 = 0; c0 < pw; c0 += 64) {
    const int c = c0 + lane;
    const bool cok = c < pw;
    for (int r0 = wv * 8; r0 < ph; r0 += NW * 8) {
      uint8_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) if (cok && r0 + u < ph) v[u] = g[(size_t)(r0 + u) * W + c];
#pragma unroll
      for (int u = 0; u < 8; u++) if (cok && r0 + u < ph) lds_pix[(r0 + u) * pitch + c] = v[u];
    }
  }
  return 0;
}

// global -> LDS by LDS-DMA (16 bytes per lane, no VGPR round trip); falls back to stage_to_lds
// for sources that are not 16-byte aligned and for the tail.  The caller waits vmcnt(0) + barrier.
template <int BLOCK>
__device__ __forceinline__ void dma_to_lds(unsigned char* lds_dst, const void* __restrict__ src, int nbytes, int tid) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  const unsigned char* g = (const unsigned char*)src;
  int done = 0;
  if ((((uintptr_t)g) & 15) == 0) {
    const int chunks = nbytes >> 4;
    const int lane = tid & 63, wv = tid >> 6;
    for (int base = wv * 64; base < chunks; base += BLOCK) {
      if (base + lane < chunks)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + ((size_t)(base + lane) << 4)), (lds_ptr_t)(lds_dst + (base << 4)), 16, 0, 0);
    }
    done = chunks << 4;
  }
  stage_to_lds<unsigned char, BLOCK, 4>(lds_dst + done, g + done, nbytes - done, tid);
}

template <typename Real, int DEPTH, bool TRACE, bool GLB>
__global__ __launch_bounds__(256) void k_scan(const DevPlan* __restrict__ plan, DevModelT<Real> m,
                                              const S0Node* __restrict__ table, WorkT<Real> w,
                                              int level, int tiles_total, int pix_bytes, int handoff, int cp_max) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int BLOCK = 256;
  constexpr int M_MAX = 512;
  const int node_n = m.node_n, leaf_n = m.leaf_n;
  const int K = min(m.K, handoff);     // this kernel stops here and hands survivors to k_finish
  const ScanLds<Real, TRACE> L(pix_bytes, K, node_n, leaf_n, M_MAX);
  const uint8_t* pix = lds + L.pix;
  S0Node* t_nodes = (S0Node*)(lds + L.nodes);
  Real* t_leaf = (Real*)(lds + L.leaf);
  CartPar<Real>* t_par = (CartPar<Real>*)(lds + L.par);
  Real* q_score = (Real*)(lds + L.q_score);
  uint16_t* q_widx = (uint16_t*)(lds + L.q_widx);
  unsigned* q_hash = (unsigned*)(lds + L.q_hash);
  uint8_t* lfbuf = lds + L.lfbuf;
  int* misc = (int*)(lds + L.misc);   // [0],[1] queue counts; [2] global base; [4..11] counter partials

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
#ifdef JDA_SCAN_TIMING
  unsigned long long stamps[15];
  int n_stamp = 0;
  int items_at[15];
#define JDA_STAMP(v) do { if (n_stamp < 15) { items_at[n_stamp] = (v); stamps[n_stamp++] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define JDA_STAMP(v) do { } while (0)
#endif
  JDA_STAMP(0);

  // XCD-aware block -> (frame, tile): blocks b, b+8, b+16.. land on one XCD
  // (MI355X dispatches block b to XCD b % 8), so the 8 frames of a group each
  // stay inside one XCD's L2.
  // level < 0: one launch covers every level of this pixel mode (GLB levels all
  // need the same LDS); tiles_total = their tiles per frame.
  int tiles_per_frame = tiles_total;
  if (level >= 0) tiles_per_frame = plan->lv[level].tiles_x * plan->lv[level].tiles_y;
  const int b = blockIdx.x;
  const int group = b / (8 * tiles_per_frame);
  const int r = b - group * (8 * tiles_per_frame);
  const int frame = group * 8 + (r & 7);
  int trel = r >> 3;
  if (frame >= w.n_frames) return;
  if (level < 0) {
    level = 0;
    for (int i = 0; i < plan->n_levels; i++) {
      const DevLevel* c = &plan->lv[i];
      if (c->tiled != (GLB ? 2 : 1)) continue;
      const int cnt = c->tiles_x * c->tiles_y;
      if (trel < cnt) { level = i; break; }
      trel -= cnt;
    }
  }
  const DevLevel lv = plan->lv[level];
  const int ty = trel / lv.tiles_x, tx = trel - ty * lv.tiles_x;
  const int wx0 = tx * lv.tw, wy0 = ty * lv.th;                 // first window of the tile
  const int twe = min(lv.tw, lv.nx - wx0), the = min(lv.th, lv.ny - wy0);
  const int x0 = wx0 * lv.step, y0 = wy0 * lv.step;             // tile origin in the frame
  const int pw = lv.win + (twe - 1) * lv.step, ph = lv.win + (the - 1) * lv.step;
  const uint8_t* img = w.frames + (size_t)frame * w.frame_stride;

  // ---- stage the pixel tile (LDS-DMA when the frame is 16-byte aligned) ----
  const int W = plan->width;
  int xshift = 0;
  if (GLB) pix = img + (size_t)y0 * W + x0;       // window origins are offsets from the tile origin in the frame
  else xshift = load_tile<BLOCK>(lds + L.pix, w.frames, w.frame_stride, img, W, x0, y0, pw, ph, lv.pitch, tid);
  // ---- stage the tables of carts [0, K): resolved nodes, leaf scores, cart parameters ----
  dma_to_lds<BLOCK>(lds + L.nodes, table + lv.s0_table, K * node_n * (int)sizeof(S0Node), tid);
  dma_to_lds<BLOCK>(lds + L.leaf, m.leaf, K * leaf_n * (int)sizeof(Real), tid);
  dma_to_lds<BLOCK>(lds + L.par, m.par0, K * (int)sizeof(CartPar<Real>), tid);
  if (tid == 0) { misc[0] = 0; misc[1] = 0; }
  __builtin_amdgcn_s_waitcnt(0);     // vmcnt(0): LDS-DMA tile loads have landed (a barrier does not drain VMEM)
  __syncthreads();
  JDA_STAMP(lv.tw * lv.th);

  const int n_tile = lv.tw * lv.th;      // phase 0 enumerates the full tile; edge windows are filtered
  int n_items = n_tile;
  int cur = 0;
  unsigned my_carts = 0;
  const int gid0 = frame * plan->windows + lv.base;

  // Phases: carts [0,8) [8,16) [16,32) [32,64) [64,128) ...; after each phase the
  // survivors are compacted so that later phases run on full waves.
  for (int c0 = 0; c0 < K;) {
    const int c1 = min(K, c0 + (c0 < 8 ? 8 : min(c0, 64)));

    // Applies carts [k, k+CNT) to this lane's window: the CNT trees first (they do
    // not depend on the score, so their LDS round trips overlap), then the scores
    // strictly in cart order with the per-cart reject test.
    auto apply = [&](auto cnt_tag, int k, const int* lf, bool& alive, Real& score, unsigned& hash, int gid) {
      constexpr int CNT = decltype(cnt_tag)::value;
      // all table reads first and unconditionally (they depend on the leaves only), so that
      // their LDS round trips overlap; the dependent part below is pure arithmetic
      CartPar<Real> p[CNT];
      Real lsv[CNT];
#pragma unroll
      for (int u = 0; u < CNT; u++) { p[u] = t_par[k + u]; lsv[u] = t_leaf[(k + u) * leaf_n + lf[u]]; }
      Real s = score;
      bool dead = false;
      int kd = k;
#pragma unroll
      for (int u = 0; u < CNT; u++) {
        if (!dead) {
          s = s + lsv[u];                                                  // c/jda.c:396
          if (p[u].norm != (Real)0) s = (s - p[u].mean) / p[u].std;        // c/jda.c:397
          if (TRACE) hash = fnv_step(hash, lf[u]);
          kd = k + u;
          dead = s < p[u].th;                                              // c/jda.c:399
        }
      }
      score = s;
      if (dead) {
        alive = false;
        my_carts += kd + 1;
        if (TRACE) { w.tr_carts[gid] = kd + 1; w.tr_score[gid] = s; w.tr_hash[gid] = hash; }
      }
    };

    const bool cart_parallel = n_items <= cp_max && c1 - c0 >= 16 && leaf_n <= 256;
    if (!cart_parallel) {
      // ---- lane = window, every wave walks the whole chunk for its own windows ----
      for (int i0 = 0; i0 < n_items; i0 += BLOCK) {
        const int i = i0 + tid;
        bool alive = i < n_items;
        int widx = 0;
        Real score = 0;
        unsigned hash = kFnvSeed;
        if (alive) {
          if (c0 == 0) {
            widx = i;
          } else {
            widx = q_widx[cur * M_MAX + i];
            score = q_score[cur * M_MAX + i];
            if (TRACE) hash = q_hash[cur * M_MAX + i];
          }
        }
        const int wy = widx / lv.tw, wx = widx - wy * lv.tw;
        if (c0 == 0) alive = alive && wx < twe && wy < the;
        const int base = (wy * lv.step) * lv.pitch + wx * lv.step + xshift;
        const int gid = gid0 + (wy0 + wy) * lv.nx + wx0 + wx;

        int k = c0;
        if (GLB) {
          // pixels come through L1/L2 here: each tree level is a global-load round trip, so
          // twice as many independent trees are kept in flight
          for (; k + 8 <= c1; k += 8) {
            if (__ballot(alive) == 0ull) break;
            if (alive) {
              int lf[8];
#pragma unroll
              for (int u = 0; u < 8; u++) lf[u] = scan_tree<DEPTH, GLB>(t_nodes + (k + u) * node_n, pix, base, m.D) - node_n;
              apply(std::integral_constant<int, 8>{}, k, lf, alive, score, hash, gid);
            }
          }
        }
        for (; k + 4 <= c1; k += 4) {
          if (__ballot(alive) == 0ull) break;
          if (alive) {
            int lf[4];
#pragma unroll
            for (int u = 0; u < 4; u++) lf[u] = scan_tree<DEPTH, GLB>(t_nodes + (k + u) * node_n, pix, base, m.D) - node_n;
            apply(std::integral_constant<int, 4>{}, k, lf, alive, score, hash, gid);
          }
        }
        for (; k < c1; k++) {
          if (alive) {
            int lf[1];
            lf[0] = scan_tree<DEPTH, GLB>(t_nodes + k * node_n, pix, base, m.D) - node_n;
            apply(std::integral_constant<int, 1>{}, k, lf, alive, score, hash, gid);
          }
        }
        // ---- compact survivors into the next queue (ballot + prefix popcount) ----
        const unsigned long long mask = __ballot(alive);
        if (mask) {
          int wbase = 0;
          if (lane == 0) wbase = atomicAdd(&misc[cur ^ 1], __popcll(mask));
          wbase = __shfl(wbase, 0);
          if (alive) {
            const int pos = wbase + __popcll(mask & lanes_below(lane));
            q_widx[(cur ^ 1) * M_MAX + pos] = (uint16_t)widx;
            q_score[(cur ^ 1) * M_MAX + pos] = score;
            if (TRACE) q_hash[(cur ^ 1) * M_MAX + pos] = hash;
          }
        }
      }
    } else {
      // ---- at most two waves of windows left: (window, cart) PAIRS are spread over the 256
      //      lanes (trees only), then the scores of the round are replayed in cart order, lane =
      //      window, from the leaf indices in LDS.  These phases are latency and issue bound (few
      //      windows, long cart ranges), so a lane should walk as few trees as possible:
      //      n_pad = windows rounded up to 16/32/64/128; lane -> window tid % n_pad, carts
      //      tid / n_pad + j * (256 / n_pad); a round covers as many carts as lfbuf holds
      //      (2048 / n_pad: 128, 64, 32, 16), i.e. at most 8 trees per lane, walked as one batch.
      const int lg = n_items <= 16 ? 4 : (n_items <= 32 ? 5 : (n_items <= 64 ? 6 : 7));
      const int n_pad = 1 << lg;
      const int rc = 2048 >> lg;                        // carts per round
      const int cstride = 256 >> lg;                    // cart stride of a lane
      const int item = tid & (n_pad - 1);
      const bool has_item = item < n_items;
      const bool replayer = tid < n_pad;                // lanes [0, n_pad) also own the windows' scores
      const int replay_waves = n_pad <= 64 ? 1 : 2;
      bool alive = replayer && has_item;
      int widx = 0;
      Real score = 0;
      unsigned hash = kFnvSeed;
      if (has_item) {
        widx = q_widx[cur * M_MAX + item];
        if (replayer) { score = q_score[cur * M_MAX + item]; if (TRACE) hash = q_hash[cur * M_MAX + item]; }
      }
      const int wy = widx / lv.tw, wx = widx - wy * lv.tw;
      const int base = (wy * lv.step) * lv.pitch + wx * lv.step + xshift;
      const int gid = gid0 + (wy0 + wy) * lv.nx + wx0 + wx;
      for (int r0 = c0; r0 < c1; r0 += rc) {
        const int r1 = min(c1, r0 + rc);
        const int ka = r0 + (tid >> lg);
        if (has_item) {
          if (ka + 7 * cstride < r1) {
            int lf8[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
              lf8[u] = scan_tree<DEPTH, GLB>(t_nodes + (ka + u * cstride) * node_n, pix, base, m.D) - node_n;
#pragma unroll
            for (int u = 0; u < 8; u++) lfbuf[item * rc + (ka + u * cstride - r0)] = (uint8_t)lf8[u];
          } else {
            for (int k = ka; k < r1; k += 4 * cstride) {
              int lf4[4];
#pragma unroll
              for (int u = 0; u < 4; u++)
                lf4[u] = scan_tree<DEPTH, GLB>(t_nodes + min(k + u * cstride, r1 - 1) * node_n, pix, base, m.D) - node_n;
#pragma unroll
              for (int u = 0; u < 4; u++)
                if (k + u * cstride < r1) lfbuf[item * rc + (k + u * cstride - r0)] = (uint8_t)lf4[u];
            }
          }
        }
        __syncthreads();
        JDA_STAMP(-100 - (r0 - c0));          // timing build: trees of this round done
        if (wv < replay_waves) {
          int k = r0;
          // 16 carts at a time when none of them is normalised: all leaf scores and thresholds
          // are fetched first (two LDS round trips for the batch instead of two per 4 carts),
          // then the recurrence runs in registers, strictly in cart order (c/jda.c:395-399)
          for (; k + 16 <= r1; k += 16) {
            if (__ballot(alive) == 0ull) break;
            const CartPar<Real> pm = t_par[k + (lane & 15)];       // lane u (mod 16): cart k+u
            if (__ballot(pm.norm != (Real)0) != 0ull) break;       // rare: the generic loop below takes over
            // thresholds: lane u holds cart k+u's; broadcast with readlane HERE, with the whole wave
            // active -- inside the divergent block below the lanes without a live window would not
            // have loaded theirs
            Real thv[16];
#pragma unroll
            for (int u = 0; u < 16; u++) thv[u] = rl(pm.th, u);
            if (alive) {
              int lf[16];
              Real lsv[16];
              // the window's 16 leaf indices are 16 consecutive bytes of lfbuf[window][cart]
              const uint4 pk = *(const uint4*)(lfbuf + item * rc + (k - r0));
              const unsigned pw4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
              for (int u = 0; u < 16; u++) lf[u] = (int)((pw4[u >> 2] >> (8 * (u & 3))) & 0xffu);
#pragma unroll
              for (int u = 0; u < 16; u++) lsv[u] = t_leaf[(k + u) * leaf_n + lf[u]];
              // branch-free: the 16 partial sums (the same adds in the same order), a bit per
              // rejecting cart, then the first set bit names the cart the window died at
              Real sums[16];
              Real sc = score;
              unsigned rej = 0u;
#pragma unroll
              for (int u = 0; u < 16; u++) {
                sc = sc + lsv[u];                                          // c/jda.c:396 (no normalisation here)
                sums[u] = sc;
                rej |= (sc < thv[u]) ? (1u << u) : 0u;                     // c/jda.c:399
              }
              if (rej) {
                const int j = __ffs((int)rej) - 1;
                Real sd = sums[0];
#pragma unroll
                for (int u = 1; u < 16; u++) sd = (j >= u) ? sums[u] : sd;
                if (TRACE)
                  for (int u = 0; u <= j; u++) hash = fnv_step(hash, lf[u]);
                score = sd;
                alive = false;
                my_carts += k + j + 1;
                if (TRACE) { w.tr_carts[gid] = k + j + 1; w.tr_score[gid] = sd; w.tr_hash[gid] = hash; }
              } else {
                if (TRACE) {
#pragma unroll
                  for (int u = 0; u < 16; u++) hash = fnv_step(hash, lf[u]);
                }
                score = sc;
              }
            }
          }
          for (; k < r1; k += 4) {
            if (__ballot(alive) == 0ull) break;
            if (alive) {
              int lf[4];
#pragma unroll
              for (int u = 0; u < 4; u++) lf[u] = (k + u < r1) ? (int)lfbuf[item * rc + (k + u - r0)] : 0;
              if (k + 4 <= r1) {
                apply(std::integral_constant<int, 4>{}, k, lf, alive, score, hash, gid);
              } else {
                for (int u = 0; k + u < r1 && alive; u++)
                  apply(std::integral_constant<int, 1>{}, k + u, lf + u, alive, score, hash, gid);
              }
            }
          }
        }
        __syncthreads();
      }
      if (wv < replay_waves) {
        const unsigned long long mask = __ballot(alive);
        if (mask) {
          int wbase = 0;
          if (lane == 0) wbase = atomicAdd(&misc[cur ^ 1], __popcll(mask));
          wbase = __shfl(wbase, 0);
          if (alive) {
            const int pos = wbase + __popcll(mask & lanes_below(lane));
            q_widx[(cur ^ 1) * M_MAX + pos] = (uint16_t)widx;
            q_score[(cur ^ 1) * M_MAX + pos] = score;
            if (TRACE) q_hash[(cur ^ 1) * M_MAX + pos] = hash;
          }
        }
      }
    }
    __syncthreads();
    cur ^= 1;
    n_items = misc[cur];
    c0 = c1;
    JDA_STAMP(n_items);
    if (n_items == 0) break;
    if (tid == 0) misc[cur ^ 1] = 0;     // next phase's output counter; readers of it are past the barrier
    __syncthreads();
  }

  // ---- windows still alive after cart K-1 -> hand-off queue (k_finish continues at cart K) ----
  unsigned handed = 0;
  if (n_items > 0) {
    if (tid == 0) misc[2] = (int)atomicAdd(&w.counters[kCntTail], (unsigned long long)n_items);
    __syncthreads();
    const unsigned gbase = (unsigned)misc[2];
    for (int i = tid; i < n_items; i += BLOCK) {
      const int widx = q_widx[cur * M_MAX + i];
      const int wy = widx / lv.tw, wx = widx - wy * lv.tw;
      const unsigned slot = gbase + i;
      if (slot < w.cap) {
        w.q_gid[slot] = (uint32_t)(gid0 + (wy0 + wy) * lv.nx + wx0 + wx);
        w.q_score[slot] = q_score[cur * M_MAX + i];
        w.q_kstart[slot] = (uint32_t)K;
        w.q_xy[slot] = (uint32_t)((wx0 + wx) * lv.step) | ((uint32_t)((wy0 + wy) * lv.step) << 16);
        w.q_wf[slot] = (uint32_t)lv.win | ((uint32_t)frame << 16);
        if (TRACE) w.q_hash[slot] = q_hash[cur * M_MAX + i];
      }
      handed += K;
    }
  }
  // ---- counters: rejected windows are final (DetectionStatisic.cart_gothrough_n);
  //      handed-off windows are counted by k_finish when they terminate.  One
  //      atomic set per workgroup, on this workgroup's counter shard. ----
  unsigned v = my_carts, hv = handed;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o); hv += __shfl_xor(hv, o); }
  __syncthreads();
  if (lane == 0) { misc[4 + wv] = (int)v; misc[8 + wv] = (int)hv; }
  __syncthreads();
  if (tid == 0) {
    const unsigned long long sv = (unsigned)misc[4] + (unsigned)misc[5] + (unsigned)misc[6] + (unsigned)misc[7];
    const unsigned long long sh = (unsigned)misc[8] + (unsigned)misc[9] + (unsigned)misc[10] + (unsigned)misc[11];
    if (sv) atomicAdd(shard_counter(w.counters, kCntCarts), sv);
    atomicAdd(shard_counter(w.counters, kCntCartsScan), sv + sh);
    atomicAdd(shard_counter(w.counters, kCntWinScan), (unsigned long long)(twe * the));
  }
#ifdef JDA_SCAN_TIMING
  JDA_STAMP(-1);
  if (tid == 0 && w.dbg && blockIdx.x < 65536) {
    unsigned long long* o = w.dbg + (size_t)blockIdx.x * 32;
    o[0] = (unsigned long long)n_stamp | ((unsigned long long)level << 32);
    for (int i = 0; i < n_stamp; i++) { o[1 + i] = stamps[i]; o[16 + i] = (unsigned long long)(long long)items_at[i]; }
  }
#endif
}

template <typename Real, bool TRACE>
static hipError_t launch_scan_depth(const DevPlan* d_plan, const DevPlan& h_plan, const DevModelT<Real>& m,
                                    const S0Node* table, const WorkT<Real>& w, int level, int handoff,
                                    hipStream_t stream) {
  // level >= 0: that LDS-tiled level; level == -1: every global-pixel level in one launch;
  // level == -2: every LDS-tiled level in one launch sized for the largest tile (small
  // batches, where one launch per level would only add launch latency)
  const bool glb = level == -1;
  int tiles = 0, pix_bytes = 0;
  if (level < 0) {
    for (int i = 0; i < h_plan.n_levels; i++) {
      const DevLevel& lv = h_plan.lv[i];
      if (lv.tiled != (glb ? 2 : 1)) continue;
      tiles += lv.tiles_x * lv.tiles_y;
      if (!glb) pix_bytes = std::max(pix_bytes, lv.pitch * (lv.win + (lv.th - 1) * lv.step));
    }
  } else {
    const DevLevel& lv = h_plan.lv[level];
    tiles = lv.tiles_x * lv.tiles_y;
    pix_bytes = lv.pitch * (lv.win + (lv.th - 1) * lv.step);
  }
  if (tiles == 0) return hipSuccess;
  handoff = std::min(handoff, scan_handoff_cap(m.node_n, m.leaf_n, (int)sizeof(Real)));
  const int carts = std::min(m.K, handoff);
  const ScanLds<Real, TRACE> L(pix_bytes, carts, m.node_n, m.leaf_n, 512);
  // windows left in a tile at or below which a phase splits the carts over the waves (64: one item
  // group, 128: two); experiments: JDA_CP_MAX / JDA_CP_MAX_GLB
  int cp_max = glb ? 128 : 128;
  if (const char* e = getenv(glb ? "JDA_CP_MAX_GLB" : "JDA_CP_MAX")) cp_max = std::max(0, std::min(128, atoi(e)));
  const int groups = (w.n_frames + 7) / 8;
  dim3 grid((unsigned)(groups * 8 * tiles)), block(256);
  auto go = [&](auto kern) {
    if (L.total > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    hipLaunchKernelGGL(kern, grid, block, L.total, stream, d_plan, m, table, w, level < 0 ? -1 : level, tiles,
                       pix_bytes, handoff, cp_max);
  };
  if (glb) {
    if (m.D == 4) go(k_scan<Real, 4, TRACE, true>);
    else if (m.D == 6) go(k_scan<Real, 6, TRACE, true>);
    else go(k_scan<Real, 0, TRACE, true>);
  } else {
    if (m.D == 4) go(k_scan<Real, 4, TRACE, false>);
    else if (m.D == 6) go(k_scan<Real, 6, TRACE, false>);
    else go(k_scan<Real, 0, TRACE, false>);
  }
  return hipGetLastError();
}

template <typename Real>
hipError_t launch_scan(int level, bool trace, int handoff, const DevPlan* d_plan, const DevPlan& h_plan,
                       const DevModelT<Real>& m, const S0Node* table, const WorkT<Real>& w,
                       hipStream_t stream) {
  if (w.n_frames == 0) return hipSuccess;
  if (level >= 0 && h_plan.lv[level].tiled != 1) return hipSuccess;
  return trace ? launch_scan_depth<Real, true>(d_plan, h_plan, m, table, w, level, handoff, stream)
               : launch_scan_depth<Real, false>(d_plan, h_plan, m, table, w, level, handoff, stream);
}

template hipError_t launch_scan<float>(int, bool, int, const DevPlan*, const DevPlan&, const DevModelT<float>&,
                                       const S0Node*, const WorkT<float>&, hipStream_t);
template hipError_t launch_scan<double>(int, bool, int, const DevPlan*, const DevPlan&, const DevModelT<double>&,
                                        const S0Node*, const WorkT<double>&, hipStream_t);

// =============================================================================
// windows k_scan does not cover -> head of the hand-off queue (k_start = 0)
// =============================================================================

template <typename Real>
__global__ void k_enqueue(const DevPlan* __restrict__ plan, WorkT<Real> w, int per_frame, int all_levels) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)per_frame * w.n_frames;
  if (idx == 0) w.counters[kCntTail] = (unsigned long long)total;   // k_scan appends behind these
  if (idx >= total) return;
  const int frame = (int)(idx / per_frame);
  int r = (int)(idx - (long long)frame * per_frame);
  int wid = -1, lvl = 0, rel = 0;
  for (int i = 0; i < plan->n_levels; i++) {
    const DevLevel* c = &plan->lv[i];
    if (!all_levels && c->tiled) continue;
    const int cnt = c->nx * c->ny;
    if (wid < 0 && r < cnt) { wid = c->base + r; lvl = i; rel = r; }
    r -= cnt;
  }
  {
    const DevLevel* c = &plan->lv[lvl];
    const int iy = rel / c->nx, ix = rel - iy * c->nx;
    w.q_xy[idx] = (uint32_t)(ix * c->step) | ((uint32_t)(iy * c->step) << 16);
    w.q_wf[idx] = (uint32_t)c->win | ((uint32_t)frame << 16);
  }
  w.q_gid[idx] = (uint32_t)(frame * plan->windows + wid);
  w.q_score[idx] = (Real)0;
  w.q_kstart[idx] = 0u;
  if (w.q_hash) w.q_hash[idx] = kFnvSeed;
}

template <typename Real>
hipError_t launch_enqueue(const DevPlan* d_plan, const DevPlan& h_plan, bool all_levels,
                          const WorkT<Real>& w, hipStream_t stream) {
  long long per_frame = 0;
  for (int i = 0; i < h_plan.n_levels; i++)
    if (all_levels || !h_plan.lv[i].tiled) per_frame += (long long)h_plan.lv[i].nx * h_plan.lv[i].ny;
  const long long total = per_frame * w.n_frames;
  if (total == 0) return hipSuccess;
  dim3 block(256), grid((unsigned)((total + 255) / 256));
  hipLaunchKernelGGL(k_enqueue<Real>, grid, block, 0, stream, d_plan, w, (int)per_frame, all_levels ? 1 : 0);
  return hipGetLastError();
}
template hipError_t launch_enqueue<float>(const DevPlan*, const DevPlan&, bool, const WorkT<float>&, hipStream_t);
template hipError_t launch_enqueue<double>(const DevPlan*, const DevPlan&, bool, const WorkT<double>&, hipStream_t);

// =============================================================================
// k_finish: one wave per surviving window
// =============================================================================

namespace {

// Similarity transform of dialect CPP's Validate (data.cpp:64-126, data.hpp:18-50).
template <typename Real>
struct Stp { Real scale, r00, r01, r10, r11; };

template <typename Real>
__device__ __forceinline__ void stp_apply(const Stp<Real>& p, Real x, Real y, Real* x2, Real* y2) {   // data.hpp:42-45
  *x2 = p.scale * (p.r00 * x + p.r01 * y);
  *y2 = p.scale * (p.r10 * x + p.r11 * y);
}

// STParameter::Calc(shape, mean_shape) by ONE lane, sequentially, in the reference's order
// (data.cpp:72-112).  cv::norm = sqrt of squares accumulated four at a time, `Mat_ /= s` =
// v*(1./s)+0. (UNPINNED restatements of OpenCV, same as the oracle).  t1/t2: LDS scratch.
__device__ __forceinline__ Stp<double> stp_calc(const double* s1, const double* __restrict__ s2, int L,
                                                double* t1, double* t2) {
  double x1c = 0., y1c = 0., x2c = 0., y2c = 0.;
  for (int i = 0; i < L; i++) { x1c += s1[2 * i]; y1c += s1[2 * i + 1]; x2c += s2[2 * i]; y2c += s2[2 * i + 1]; }
  x1c /= (double)L; y1c /= (double)L; x2c /= (double)L; y2c /= (double)L;
  for (int i = 0; i < L; i++) {
    t1[2 * i] = s1[2 * i] - x1c; t1[2 * i + 1] = s1[2 * i + 1] - y1c;
    t2[2 * i] = s2[2 * i] - x2c; t2[2 * i + 1] = s2[2 * i + 1] - y2c;
  }
  auto cvnorm = [](const double* v, int n) {
    double s = 0.;
    int i = 0;
    for (; i <= n - 4; i += 4) { const double v0 = v[i], v1 = v[i + 1], v2 = v[i + 2], v3 = v[i + 3]; s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3; }
    for (; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
  };
  const double scale1 = cvnorm(t1, 2 * L), scale2 = cvnorm(t2, 2 * L);
  Stp<double> p;
  p.scale = scale1 / scale2;
  const double a1 = 1. / scale1, a2 = 1. / scale2;
  for (int i = 0; i < 2 * L; i++) { t1[i] = t1[i] * a1 + 0.; t2[i] = t2[i] * a2 + 0.; }
  double num = 0., den = 0.;
  for (int i = 0; i < L; i++) {
    num += t1[2 * i + 1] * t2[2 * i] - t1[2 * i] * t2[2 * i + 1];
    den += t1[2 * i] * t2[2 * i] + t1[2 * i + 1] * t2[2 * i + 1];
  }
  const double norm = sqrt(num * num + den * den);
  const double sn = num / norm, cs = den / norm;
  p.r00 = cs; p.r01 = -sn; p.r10 = sn; p.r11 = cs;
  return p;
}

// Where a window reads its pixels for one feature scale.
struct View {
  const uint8_t* img; int w, h, ox, oy;
  int pw;   // side of the patch the feature coordinates are scaled by and clamped to
};

// Feature of one split node for the window whose shape is sh[] (c/jda.c:370-391,
// data.cpp:18-58).
template <typename DL, bool MULTI, bool ST>
__device__ __forceinline__ int node_feature(typename DL::Node nd, const typename DL::Real* sh, int win,
                                            const View& v0, const View& v1, const View& v2,
                                            const Stp<typename DL::Real>& stp, bool apply_st) {
  using Real = typename DL::Real;
  const Real s1x = sh[nd.lm1x2], s1y = sh[nd.lm1x2 + 1];
  const Real s2x = sh[nd.lm2x2], s2y = sh[nd.lm2x2 + 1];
  if (ST && apply_st) {       // stp_mc.Apply on both offsets, data.cpp:33-34 (stage 0's are pre-applied)
    Real ax, ay, bx, by;
    stp_apply<Real>(stp, nd.o1x, nd.o1y, &ax, &ay);
    stp_apply<Real>(stp, nd.o2x, nd.o2y, &bx, &by);
    nd.o1x = ax; nd.o1y = ay; nd.o2x = bx; nd.o2y = by;
  }
  if (!MULTI) {
    // (DL::pixel's fused clamp does not pay here: k_finish is not VALU bound, measured 3 % slower)
    const int x1 = clamp_win(DL::coord(s1x, nd.o1x, win), win), y1 = clamp_win(DL::coord(s1y, nd.o1y, win), win);
    const int x2 = clamp_win(DL::coord(s2x, nd.o2x, win), win), y2 = clamp_win(DL::coord(s2y, nd.o2y, win), win);
    // rows and widths are below 2^16: 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate)
    const int a = v0.img[__umul24((unsigned)(v0.oy + y1), (unsigned)v0.w) + (unsigned)(v0.ox + x1)];
    const int b = v0.img[__umul24((unsigned)(v0.oy + y2), (unsigned)v0.w) + (unsigned)(v0.ox + x2)];
    return a - b;
  }
  // Multi-scale models.  Dialect C scales and clamps with the FULL window side for
  // every scale (c/jda.c:347-354: ps[1].w = ps[2].w = win_size) and can therefore
  // leave the half/quarter image: reads are clamped to the image (documented
  // divergence from its out-of-bounds reads).  Dialect CPP uses each patch's own
  // size (data.cpp:37-51), which always stays inside the image.
  const View& v = nd.scale == 0 ? v0 : (nd.scale == 1 ? v1 : v2);
  const int pw = v.pw;
  const int x1 = clamp_win(DL::coord(s1x, nd.o1x, pw), pw);
  const int y1 = clamp_win(DL::coord(s1y, nd.o1y, pw), pw);
  const int x2 = clamp_win(DL::coord(s2x, nd.o2x, pw), pw);
  const int y2 = clamp_win(DL::coord(s2y, nd.o2y, pw), pw);
  const int gx1 = min(v.ox + x1, v.w - 1), gy1 = min(v.oy + y1, v.h - 1);
  const int gx2 = min(v.ox + x2, v.w - 1), gy2 = min(v.oy + y2, v.h - 1);
  const int a = v.img[(unsigned)(gy1 * v.w + gx1)];
  const int b = v.img[(unsigned)(gy2 * v.w + gx2)];
  return a - b;
}

// Views of a queued window from its packed (x, y, win, frame) -- the producers of the queues
// know these, so no division or level search is needed here.
template <typename Real>
__device__ __forceinline__ void decode_window(const DevPlan* plan, const WorkT<Real>& w, uint32_t xy, uint32_t wf,
                                              float inv_sqrt2, int* win, View* v0, View* v1, View* v2, bool multi) {
  const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
  const int wn = (int)(wf & 0xffffu), frame = (int)(wf >> 16);
  *win = wn;
  v0->img = w.frames + (size_t)frame * w.frame_stride; v0->w = plan->width; v0->h = plan->height; v0->ox = x; v0->oy = y;
  v0->pw = wn;
  if (multi) {
    v1->img = w.half + (size_t)frame * w.half_stride; v1->w = w.hw; v1->h = w.hh;
    v2->img = w.quarter + (size_t)frame * w.quarter_stride; v2->w = w.qw; v2->h = w.qh;
    if (sizeof(Real) == 4) {
      // dialect C, c/jda.c:345-354: origins by float multiply / integer halving, full-size patches
      v1->ox = (int)((float)x * inv_sqrt2); v1->oy = (int)((float)y * inv_sqrt2); v1->pw = wn;
      v2->ox = x / 2; v2->oy = y / 2; v2->pw = wn;
    } else {
      // dialect CPP, cascador.cpp:340-343: Rect(int(x/r), int(y/r), int(win/r), ..), r = sqrt(2.) in double
      const double r = sqrt(2.0);
      v1->ox = (int)((double)x / r); v1->oy = (int)((double)y / r); v1->pw = (int)((double)wn / r);
      v2->ox = x / 2; v2->oy = y / 2; v2->pw = wn / 2;
    }
  }
}

// value held by lane j, as a wave-uniform scalar

}  // namespace

// Tree walks of G carts (k[0..G)) of one stage for the window whose shape is sh[],
// in lockstep: per tree level the G node records are fetched together, then the
// 2G pixels, so the memory round trips of the G walks overlap.  -> leaf indices.
template <typename DL, int G, bool MULTI, bool ST>
__device__ __forceinline__ void walk_carts(const typename DL::Node* __restrict__ stage_nodes, const int* k,
                                           int depth, int node_n, const typename DL::Real* sh, int win,
                                           const View& v0, const View& v1, const View& v2,
                                           const Stp<typename DL::Real>& stp, bool apply_st, int* leaf) {
  int node[G];
#pragma unroll
  for (int g = 0; g < G; g++) node[g] = 0;
  for (int d = 0; d < depth - 1; d++) {
    typename DL::Node nd[G];
#pragma unroll
    for (int g = 0; g < G; g++) nd[g] = stage_nodes[(unsigned)(k[g] * node_n + node[g])];
    int feat[G];
#pragma unroll
    for (int g = 0; g < G; g++) feat[g] = node_feature<DL, MULTI, ST>(nd[g], sh, win, v0, v1, v2, stp, apply_st);
#pragma unroll
    for (int g = 0; g < G; g++) node[g] = 2 * node[g] + (feat[g] <= nd[g].th ? 1 : 2);   // c/jda.c:392-393
  }
#pragma unroll
  for (int g = 0; g < G; g++) leaf[g] = node[g] - node_n;
}

// Stage-0 walks from the resolved tables k_scan uses (S0Node, one 8-byte record per node with both
// pixel offsets and the threshold): one record load instead of two, no coordinate arithmetic.
// mode 2: offsets are frame offsets (row pitch = frame width); mode 1: offsets are LDS-tile
// offsets y*pitch + x, split back into (y, x) with an exact float division ((off + 0.5) / pitch is
// at least 0.5/pitch away from an integer, the float error is below 1e-4 of that).
template <int G>
__device__ __forceinline__ void walk_carts_s0(const S0Node* __restrict__ tbl, const int* k, int depth, int node_n,
                                              int mode, int pitch, float inv_pitch, const uint8_t* __restrict__ wbase,
                                              int W, int* leaf) {
  int node[G];
#pragma unroll
  for (int g = 0; g < G; g++) node[g] = 0;
  for (int d = 0; d < depth - 1; d++) {
    S0Node r[G];
#pragma unroll
    for (int g = 0; g < G; g++) r[g] = tbl[(unsigned)(k[g] * node_n + node[g])];
    unsigned o1[G], o2[G];
    int th[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
      if (mode == 2) {
        o1[g] = r[g].lo & 0x1fffffu;
        o2[g] = __builtin_amdgcn_alignbit(r[g].hi, r[g].lo, 21) & 0x1fffffu;
        th[g] = (int)(r[g].hi >> 10) - 256;
      } else {
        const unsigned a = r[g].lo & 0xffffu, b = r[g].lo >> 16;
        const unsigned ya = (unsigned)(((float)a + 0.5f) * inv_pitch), yb = (unsigned)(((float)b + 0.5f) * inv_pitch);
        o1[g] = __umul24(ya, (unsigned)W) + (a - __umul24(ya, (unsigned)pitch));
        o2[g] = __umul24(yb, (unsigned)W) + (b - __umul24(yb, (unsigned)pitch));
        th[g] = (int)r[g].hi;
      }
    }
    int pa[G], pb[G];
#pragma unroll
    for (int g = 0; g < G; g++) { pa[g] = wbase[o1[g]]; pb[g] = wbase[o2[g]]; }
#pragma unroll
    for (int g = 0; g < G; g++) node[g] = 2 * node[g] + (pa[g] - pb[g] <= th[g] ? 1 : 2);   // c/jda.c:391-393
  }
#pragma unroll
  for (int g = 0; g < G; g++) leaf[g] = node[g] - node_n;
}

// Score recurrence of c/jda.c:395-399 over the carts held by lanes [jbeg, jend)
// of one 64-cart group, replayed strictly in cart order.  ls/th_k/mean_k/std_k/lf
// are per-lane values of cart (group base + lane).  Returns the lane of the
// rejecting cart or -1; score/hash are left as they stood at that cart.
template <typename Real, bool TRACE>
__device__ __forceinline__ int replay_scores(Real& score, unsigned& hash, Real ls, Real th_k, Real mean_k, Real std_k,
                                             unsigned long long normmask, int lf, int jbeg, int jend) {
  if (jbeg == 0 && jend == 64 && normmask == 0ull) {
    // Common case, branch-free: the running score is wave-uniform; after every
    // add ALL lanes compare it with their own cart's threshold and only bit j
    // of that ballot is kept.  Same adds in the same order as the scalar loop.
    Real s = score;
    unsigned long long rej = 0ull;
    for (int j0 = 0; j0 < 64 && rej == 0ull; j0 += 16) {      // stop at the 16-cart block that rejects
#pragma unroll
      for (int jj = 0; jj < 16; jj++) {
        const int j = j0 + jj;
        s = s + rl(ls, j);                                     // c/jda.c:396
        rej |= __ballot(s < th_k) & (1ull << j);               // c/jda.c:399
      }
    }
    if (rej == 0ull) {
      if (TRACE) {
#pragma unroll 8
        for (int j = 0; j < 64; j++) hash = fnv_step(hash, rl(lf, j));
      }
      score = s;
      return -1;
    }
    const int jr = __ffsll((long long)rej) - 1;
    Real s2 = score;
    for (int j = 0; j <= jr; j++) {                            // the score as it stood at the rejecting cart
      s2 = s2 + rl(ls, j);
      if (TRACE) hash = fnv_step(hash, rl(lf, j));
    }
    score = s2;
    return jr;
  }
  for (int j = jbeg; j < jend; j++) {
    Real s = score + rl(ls, j);                                                     // c/jda.c:396
    if ((normmask >> j) & 1ull) s = (s - rl(mean_k, j)) / rl(std_k, j);             // c/jda.c:397
    score = s;
    if (TRACE) hash = fnv_step(hash, rl(lf, j));
    if (s < rl(th_k, j)) return j;                                                  // c/jda.c:399
  }
  return -1;
}

// Stages [t_begin, t_end) for every window of the input queue.  Windows that are
// still alive after stage t_end-1 go to the mid queue (t_end < T) or, after the
// final threshold, to the detection list (t_end == T).
template <typename DL, bool TRACE, int kG, bool MULTI, bool ST>
__global__ __launch_bounds__(64) void k_finish(const DevPlan* __restrict__ plan, DevModelT<typename DL::Real> m,
                                               WorkT<typename DL::Real> w, int multi_i, float inv_sqrt2,
                                               int t_begin, int t_end, int apply_th, typename DL::Real final_th,
                                               const S0Node* __restrict__ s0_table) {
  using Real = typename DL::Real;
  using Node = typename DL::Node;
  constexpr bool kCpp = sizeof(Real) == 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int T = m.T, K = m.K, node_n = m.node_n, leaf_n = m.leaf_n, dim = m.dim;
  const int dim_pad = (dim + 1) & ~1;
  Real* sh = (Real*)lds;                                     // current shape        [dim_pad]
  Real* sh2 = sh + dim_pad;                                  // shape being built    [dim_pad]
  uint32_t* lbf = (uint32_t*)(sh2 + dim_pad);                // W row (in elements) chosen by every cart [K]
  int* stage_cnt = (int*)(lbf + ((K + 3) & ~3));             // per-block stage counters
  Real* st_tmp = (Real*)(stage_cnt + kMaxStages);            // similarity-transform scratch [2*dim_pad + 8] (ST only)
  constexpr bool multi = MULTI;   // split nodes read the half/quarter images too
  (void)multi_i;
  const int lane = threadIdx.x;
  if (lane < kMaxStages) stage_cnt[lane] = 0;
  const bool from_scan = t_begin == 0;
  const unsigned n = (unsigned)min(w.counters[from_scan ? kCntTail : kCntMid], (unsigned long long)w.cap);
  unsigned long long carts_acc = 0;

  for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
    const uint32_t gid = from_scan ? w.q_gid[i] : w.m_gid[i];
    Real score = from_scan ? w.q_score[i] : w.m_score[i];
    const int kstart = from_scan ? (int)w.q_kstart[i] : 0;
    unsigned hash = kFnvSeed;
    if (TRACE) hash = from_scan ? w.q_hash[i] : w.m_hash[i];
    int win;
    View v0{}, v1{}, v2{};
    const uint32_t xy = from_scan ? w.q_xy[i] : w.m_xy[i];
    const uint32_t wf = from_scan ? w.q_wf[i] : w.m_wf[i];
    decode_window<Real>(plan, w, xy, wf, inv_sqrt2, &win, &v0, &v1, &v2, multi);
    // stage 0 of a window whose level has resolved tables (every level k_scan covers): walk from them
    int s0_mode = 0, s0_pitch = 0;
    float s0_inv = 0.f;
    const S0Node* s0_tbl = nullptr;
    if (!MULTI && s0_table != nullptr && t_begin == 0) {
      const bool hit = lane < plan->n_levels && plan->lv[lane].win == win;
      const unsigned long long mh = __ballot(hit);
      if (mh) {
        const DevLevel lv = plan->lv[__ffsll((long long)mh) - 1];
        if (lv.tiled) { s0_mode = lv.tiled; s0_pitch = lv.pitch; s0_inv = 1.0f / (float)lv.pitch; s0_tbl = s0_table + lv.s0_table; }
      }
    }
    const uint8_t* wbase = v0.img + (size_t)v0.oy * v0.w + v0.ox;
    __syncthreads();                       // previous window's readers are done with sh
    {
      const Real* src = from_scan ? m.mean_shape : w.m_shape + (size_t)i * dim;
      for (int d = lane; d < dim; d += 64) sh[d] = src[d];
    }
    __syncthreads();

    bool alive = true;
    int carts_n = 0;
    for (int t = t_begin; t < t_end; t++) {
      const Node* nodes = (const Node*)m.nodes + (size_t)t * K * node_n;
      const Real* leaf_tab = m.leaf + (size_t)t * K * leaf_n;
      const Real* cth = m.cth + (size_t)t * K;
      const Real* cmean = m.cmean + (size_t)t * K;
      const Real* cstd = m.cstd + (size_t)t * K;
      const uint8_t* cnorm = m.cnorm + (size_t)t * K;
      const int kbeg = t == 0 ? min(kstart, K) : 0;   // first cart whose score is still to be applied
      const int k_first = kbeg & ~63;
      // similarity transform of this stage (cascador.cpp:180); identity unless enabled
      Stp<Real> stp;
      stp.scale = 1; stp.r00 = 1; stp.r01 = 0; stp.r10 = 0; stp.r11 = 1;
      if constexpr (ST) {
        if (lane == 0) {
          const Stp<double> p = stp_calc((const double*)sh, (const double*)m.mean_shape_raw, m.L, (double*)st_tmp,
                                         (double*)st_tmp + dim_pad);
          double* o = (double*)st_tmp + 2 * dim_pad;
          o[0] = p.scale; o[1] = p.r00; o[2] = p.r01; o[3] = p.r10; o[4] = p.r11;
        }
        __syncthreads();
        const Real* o = st_tmp + 2 * dim_pad;
        stp.scale = o[0]; stp.r00 = o[1]; stp.r01 = o[2]; stp.r10 = o[3]; stp.r11 = o[4];
      }
      const bool apply_st = ST && t > 0;              // stage 0's node offsets carry the transform already

      // ---- tree walks, kG groups of 64 carts per round (the shape is fixed during a
      //      stage, so the trees of a stage are independent of each other and of the
      //      score); then the score recurrence replayed in cart order ----
      for (int k0 = k_first; k0 < K && alive; k0 += 64 * kG) {
        int kk[kG], lf[kG], nrm[kG];
        Real ls[kG], thk[kG], mk[kG], sk[kG];
#pragma unroll
        for (int g = 0; g < kG; g++) kk[g] = min(k0 + g * 64 + lane, K - 1);   // clamped lanes repeat cart K-1
        if (t == 0 && s0_mode) walk_carts_s0<kG>(s0_tbl, kk, m.D, node_n, s0_mode, s0_pitch, s0_inv, wbase, v0.w, lf);
        else walk_carts<DL, kG, MULTI, ST>(nodes, kk, m.D, node_n, sh, win, v0, v1, v2, stp, apply_st, lf);
#pragma unroll
        for (int g = 0; g < kG; g++) {
          const int k = k0 + g * 64 + lane;
          ls[g] = 0; thk[g] = 0; mk[g] = 0; sk[g] = 1; nrm[g] = 0;
          if (k < K) {
            lbf[k] = (uint32_t)(k * leaf_n + lf[g]) * (uint32_t)dim;
            ls[g] = leaf_tab[(unsigned)(k * leaf_n + lf[g])];
            thk[g] = cth[k];
            nrm[g] = cnorm[k];
            if (nrm[g]) { mk[g] = cmean[k]; sk[g] = cstd[k]; }
          }
        }
#pragma unroll
        for (int g = 0; g < kG; g++) {
          const int kg = k0 + g * 64;
          if (kg >= K || !alive) break;
          const unsigned long long normmask = __ballot(nrm[g] != 0);
          const int jr = replay_scores<Real, TRACE>(score, hash, ls[g], thk[g], mk[g], sk[g], normmask, lf[g],
                                                    max(0, kbeg - kg), min(64, K - kg));
          if (jr >= 0) { alive = false; carts_n = t * K + kg + jr + 1; }
        }
      }
      if (!alive) break;
      // leaves of the carts k_scan already scored (needed only now that the stage is passed)
      for (int k0 = 0; k0 < k_first; k0 += 128) {
        int kk[2], lf[2];
        kk[0] = min(k0 + lane, k_first - 1); kk[1] = min(k0 + 64 + lane, k_first - 1);
        if (t == 0 && s0_mode) walk_carts_s0<2>(s0_tbl, kk, m.D, node_n, s0_mode, s0_pitch, s0_inv, wbase, v0.w, lf);
        else walk_carts<DL, 2, MULTI, ST>(nodes, kk, m.D, node_n, sh, win, v0, v1, v2, stp, apply_st, lf);
        if (k0 + lane < k_first) lbf[k0 + lane] = (uint32_t)((k0 + lane) * leaf_n + lf[0]) * (uint32_t)dim;
        if (k0 + 64 + lane < k_first) lbf[k0 + 64 + lane] = (uint32_t)((k0 + 64 + lane) * leaf_n + lf[1]) * (uint32_t)dim;
      }
      __syncthreads();
      // ---- stage regression: K weight rows added strictly in cart order
      //      (c/jda.c:404-411); dialect CPP sums the delta from zero and adds it
      //      once (btcart.cpp:407-424) ----
      const Real* wt = m.w + (size_t)t * K * leaf_n * dim;
      for (int d = lane; d < dim; d += 64) {
        Real acc = kCpp ? (Real)0 : sh[d];
        const Real* col = wt + d;
        int k = 0;
        for (; k + 32 <= K; k += 32) {          // 32 row loads in flight, then 32 ordered adds
          Real r[32];
#pragma unroll
          for (int u = 0; u < 32; u++) r[u] = col[lbf[k + u]];
#pragma unroll
          for (int u = 0; u < 32; u++) acc = acc + r[u];
        }
        for (; k < K; k++) acc = acc + col[lbf[k]];
        if (kCpp) {
          // stp_mc.Apply(delta, delta) (btcart.cpp:422, data.hpp:42-45) on the (dx,dy) pair held by
          // lanes d, d^1; with the identity parameter this is the literal 1*(1*x+0*y) / 1*(0*x+1*y)
          const Real other = __shfl_xor(acc, 1);
          acc = (d & 1) ? stp.scale * (stp.r10 * other + stp.r11 * acc) : stp.scale * (stp.r00 * acc + stp.r01 * other);
          acc = sh[d] + acc;
        }
        sh2[d] = acc;
      }
      __syncthreads();
      { Real* tmp = sh; sh = sh2; sh2 = tmp; }
      if (lane == 0) stage_cnt[t] += 1;
    }

    if (!alive || t_end == T) {
      // the window's walk is over: account for it (reference counting: Validate's n)
      if (alive) carts_n = T * K;
      carts_acc += (unsigned long long)carts_n;
      if (TRACE) {
        if (lane == 0) { w.tr_carts[gid] = carts_n; w.tr_score[gid] = score; w.tr_hash[gid] = hash; }
        for (int d = lane; d < dim; d += 64) w.tr_shape[(size_t)gid * dim + d] = sh[d];
      }
      if (alive && !(apply_th && score < final_th)) {            // c/jda.c:414
        unsigned o = 0;
        if (lane == 0) o = (unsigned)atomicAdd(&w.counters[kCntOut], 1ull);
        o = (unsigned)__shfl((int)o, 0);
        if (o < w.cap) {
          if (lane == 0) { w.out_gid[o] = gid; w.out_score[o] = score; }
          for (int d = lane; d < dim; d += 64) w.out_shape[(size_t)o * dim + d] = sh[d];
        }
      }
    } else {
      // alive with stages left: park it in the mid queue for the next launch
      unsigned o = 0;
      if (lane == 0) o = (unsigned)atomicAdd(&w.counters[kCntMid], 1ull);
      o = (unsigned)__shfl((int)o, 0);
      if (o < w.cap) {
        if (lane == 0) { w.m_gid[o] = gid; w.m_score[o] = score; w.m_xy[o] = xy; w.m_wf[o] = wf; if (TRACE) w.m_hash[o] = hash; }
        for (int d = lane; d < dim; d += 64) w.m_shape[(size_t)o * dim + d] = sh[d];
      }
    }
  }
  __syncthreads();
  if (lane < T && stage_cnt[lane]) atomicAdd(shard_counter(w.counters, kCntStage0 + lane), (unsigned long long)stage_cnt[lane]);
  if (lane == 0 && carts_acc) atomicAdd(shard_counter(w.counters, kCntCarts), carts_acc);
}

namespace {
template <typename DL>
hipError_t launch_finish_impl(bool trace, int t_begin, int t_end, bool apply_th, typename DL::Real th,
                              const DevPlan* d_plan, const DevModelT<typename DL::Real>& m,
                              const WorkT<typename DL::Real>& w, int groups, long long n_hint, const S0Node* s0_table,
                              hipStream_t stream) {
  using Real = typename DL::Real;
  const int dim_pad = (m.dim + 1) & ~1;
  const bool st = sizeof(Real) == 8 && m.similarity != 0;
  const size_t lds = 2 * (size_t)dim_pad * sizeof(Real) + (size_t)((m.K + 3) & ~3) * 4 + kMaxStages * sizeof(int) +
                     (st ? (2 * (size_t)dim_pad + 8) * sizeof(Real) : 0);
  const int multi = (w.half != nullptr) ? 1 : 0;
  const float r = 1.f / sqrtf(2.f);
  // n_hint >= 0: the queue length is known on the host -> one window per workgroup (up to
  // 1M workgroups, grid-stride beyond), so the hardware dispatcher balances the very
  // uneven per-window cost; n_hint < 0: fixed grid, windows dealt round-robin.
  unsigned blocks = w.cap;
  if (blocks > 256u * 64u) blocks = 256u * 64u;
  if (n_hint >= 0) blocks = (unsigned)std::min<long long>(n_hint, 1 << 20);
  if (blocks == 0) return hipSuccess;
  auto go = [&](auto kern) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, stream, d_plan, m, w, multi, r, t_begin, t_end,
                       apply_th ? 1 : 0, th, s0_table);
  };
  // groups = 64-cart groups walked speculatively per round: 1 where most windows are
  // rejected within a few carts (throughput), 4 where most pass (latency)
  auto pick = [&](auto trace_tag, auto multi_tag, auto st_tag) {
    constexpr bool TR = decltype(trace_tag)::value, MU = decltype(multi_tag)::value;
    constexpr bool STT = decltype(st_tag)::value && sizeof(Real) == 8;
    if (groups >= 4) go(k_finish<DL, TR, 4, MU, STT>);
    else if (groups == 3) go(k_finish<DL, TR, 3, MU, STT>);
    else if (groups >= 2) go(k_finish<DL, TR, 2, MU, STT>);
    else go(k_finish<DL, TR, 1, MU, STT>);
  };
  auto pick_m = [&](auto trace_tag, auto st_tag) {
    if (multi) pick(trace_tag, std::true_type{}, st_tag); else pick(trace_tag, std::false_type{}, st_tag);
  };
  if (st) { if (trace) pick_m(std::true_type{}, std::true_type{}); else pick_m(std::false_type{}, std::true_type{}); }
  else { if (trace) pick_m(std::true_type{}, std::false_type{}); else pick_m(std::false_type{}, std::false_type{}); }
  return hipGetLastError();
}
}  // namespace

template <>
hipError_t launch_finish<float>(bool trace, int t_begin, int t_end, bool apply_final_th, float final_th,
                                const DevPlan* d_plan, const DevModelT<float>& m, const WorkT<float>& w,
                                int groups, long long n_hint, const S0Node* s0_table, hipStream_t stream) {
  return launch_finish_impl<DialectC>(trace, t_begin, t_end, apply_final_th, final_th, d_plan, m, w, groups, n_hint, s0_table, stream);
}
template <>
hipError_t launch_finish<double>(bool trace, int t_begin, int t_end, bool apply_final_th, double final_th,
                                 const DevPlan* d_plan, const DevModelT<double>& m, const WorkT<double>& w,
                                 int groups, long long n_hint, const S0Node* s0_table, hipStream_t stream) {
  return launch_finish_impl<DialectCPP>(trace, t_begin, t_end, apply_final_th, final_th, d_plan, m, w, groups, n_hint, s0_table, stream);
}

// =============================================================================
// k_stage: dense mode -- one whole stage for a TILE of windows per workgroup
// =============================================================================
//
// When most windows survive (all-pass-like models, weak cascades) the wave-per-window
// k_finish re-reads every node record and weight row from L1/L2 for every window: the whole
// model (hundreds of KB) per window.  Here a workgroup owns a 16 x 16 tile of windows of one
// (frame, level), like k_scan, and walks stage t for all of them with lane = window:
//   * node records, leaf scores, cart parameters AND regression weight rows of `chunk`
//     carts at a time are staged in LDS once (LDS-DMA) and shared by the tile's 256 windows;
//   * pixels come from the LDS tile (or L1/L2 for windows too large for a tile);
//   * per-window shapes sit in LDS as [coordinate][window] (conflict-free column reads);
//   * G carts are walked at once per lane (independent LDS round trips overlap), then their
//     scores are applied strictly in cart order with the per-cart reject test;
//   * the regression sums live in registers and take each cart's weight row (an LDS read the
//     lanes with the same leaf share) in cart order.  A window rejected later in the stage
//     discards its sums, so they need no predicate.
// Per-window state (score, carts evaluated, shape) lives in global arrays between stages.
// Same arithmetic in the same order as k_finish / the reference (c/jda.c:364-411).

constexpr int kDenseTw = 16, kDenseTh = 16, kDenseM = kDenseTw * kDenseTh;

namespace {
template <typename Real, typename Node>
struct DenseLds {
  int pix, sh, nodes, leaf, par, wts, total;
  __host__ __device__ DenseLds(int pix_bytes, int dim, int node_n, int leaf_n, int chunk, int acc) {
    int o = 0;
    pix = o; o += (pix_bytes + 15) & ~15;
    sh = o; o += dim * kDenseM * (int)sizeof(Real);
    nodes = o; o += chunk * node_n * (int)sizeof(Node);
    leaf = o; o += ((chunk * leaf_n * (int)sizeof(Real)) + 15) & ~15;
    par = o; o += chunk * 4 * (int)sizeof(Real);
    wts = o; o += ((chunk * leaf_n * dim + acc) * (int)sizeof(Real) + 15) & ~15;   // + one register row of slack:
    total = o;                                                                   // rows are read ACC wide
  }
};
}  // namespace

template <typename DL, bool TRACE, bool GLB, int ACC>
__global__ __launch_bounds__(kDenseM)
__attribute__((amdgpu_waves_per_eu(ACC * (int)sizeof(typename DL::Real) <= 256 ? 2 : 1)))   // LDS allows 2-3 workgroups per CU
void k_stage(const DevPlan* __restrict__ plan, DevModelT<typename DL::Real> m,
                                                   WorkT<typename DL::Real> w, int level, int t, int pix_bytes,
                                                   int pitch, int chunk, int apply_th, typename DL::Real final_th) {
  using Real = typename DL::Real;
  using Node = typename DL::Node;
  constexpr bool kCpp = sizeof(Real) == 8;
  constexpr int BLOCK = kDenseM;
  constexpr int G = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int T = m.T, K = m.K, node_n = m.node_n, leaf_n = m.leaf_n, dim = m.dim, depth = m.D - 1;
  const DenseLds<Real, Node> L(pix_bytes, dim, node_n, leaf_n, chunk, ACC);
  const uint8_t* pix = lds + L.pix;
  Real* sh = (Real*)(lds + L.sh);                   // [dim][256]
  const Node* t_nodes = (const Node*)(lds + L.nodes);
  const Real* t_leaf = (const Real*)(lds + L.leaf);
  const CartPar<Real>* t_par = (const CartPar<Real>*)(lds + L.par);
  const Real* t_w = (const Real*)(lds + L.wts);
  const int tid = threadIdx.x;

  const DevLevel lv = plan->lv[level];
  const int tiles_x = (lv.nx + kDenseTw - 1) / kDenseTw, tiles_y = (lv.ny + kDenseTh - 1) / kDenseTh;
  const int tiles_per_frame = tiles_x * tiles_y;
  const int b = blockIdx.x;                          // XCD-aware (frame, tile) mapping as in k_scan
  const int group = b / (8 * tiles_per_frame);
  const int r = b - group * (8 * tiles_per_frame);
  const int frame = group * 8 + (r & 7);
  const int trel = r >> 3;
  if (frame >= w.n_frames) return;
  const int ty = trel / tiles_x, tx = trel - ty * tiles_x;
  const int wx0 = tx * kDenseTw, wy0 = ty * kDenseTh;
  const int twe = min(kDenseTw, lv.nx - wx0), the = min(kDenseTh, lv.ny - wy0);
  const int x0 = wx0 * lv.step, y0 = wy0 * lv.step;
  const int pw = lv.win + (twe - 1) * lv.step, ph = lv.win + (the - 1) * lv.step;
  const int W = plan->width, win = lv.win;
  const uint8_t* img = w.frames + (size_t)frame * w.frame_stride;

  // this thread's window and its state
  const int wx = tid & (kDenseTw - 1), wy = tid / kDenseTw;
  const bool valid = wx < twe && wy < the;
  const uint32_t gid = (uint32_t)(frame * plan->windows + lv.base + (wy0 + wy) * lv.nx + wx0 + wx);
  bool alive = valid;
  Real score = 0;
  unsigned hash = kFnvSeed;
  if (t > 0 && valid) {
    alive = w.st_carts[gid] < 0;
    if (alive) { score = w.m_score[gid]; if (TRACE) hash = w.m_hash[gid]; }
  }
  const bool entered = alive;
  if (__syncthreads_or(alive ? 1 : 0) == 0) return;          // nothing left to do in this tile

  int xshift = 0, ppitch = pitch;
  if (GLB) { pix = img + (size_t)y0 * W + x0; ppitch = W; }
  else xshift = load_tile<BLOCK>(lds + L.pix, w.frames, w.frame_stride, img, W, x0, y0, pw, ph, pitch, tid);
  const int base = (wy * lv.step) * ppitch + wx * lv.step + xshift - DL::kBias * (ppitch + 1);   // pixel_pair is kBias-based

  // stage-start shape: LDS column for the tree walks; the regression sums start from it
  // (dialect C) or from zero (dialect CPP, btcart.cpp:407-424).  The sums are kept as (x, y)
  // pairs: rows are read 2 coordinates at a time and added with one packed add (two
  // independent IEEE adds -- same bits as two scalar ones).
  typedef Real Vec2 __attribute__((ext_vector_type(2)));
  Vec2 acc[ACC / 2];
  {
    const Real* src = (t == 0 || !alive) ? m.mean_shape : w.m_shape + (size_t)gid * dim;
#pragma unroll
    for (int d2 = 0; d2 < ACC / 2; d2++) {
      acc[d2] = Vec2{0, 0};
      if (2 * d2 < dim) {
        const Real vx = src[2 * d2], vy = src[2 * d2 + 1];
        sh[(2 * d2) * BLOCK + tid] = vx; sh[(2 * d2 + 1) * BLOCK + tid] = vy;
        if (!kCpp) acc[d2] = Vec2{vx, vy};
      }
    }
  }
  int carts_n = -1;

  const Node* g_nodes = (const Node*)m.nodes + (size_t)t * K * node_n;
  const Real* g_leaf = m.leaf + (size_t)t * K * leaf_n;
  const CartPar<Real>* g_par = (const CartPar<Real>*)m.par0 + (size_t)t * K;
  const Real* g_w = m.w + (size_t)t * K * leaf_n * dim;

#pragma nounroll
  for (int c0 = 0; c0 < K; c0 += chunk) {
    const int cn = min(chunk, K - c0);
    if (__syncthreads_or(alive ? 1 : 0) == 0) break;          // also: previous chunk's table readers are done
    dma_to_lds<BLOCK>(lds + L.nodes, g_nodes + (size_t)c0 * node_n, cn * node_n * (int)sizeof(Node), tid);
    dma_to_lds<BLOCK>(lds + L.leaf, g_leaf + (size_t)c0 * leaf_n, cn * leaf_n * (int)sizeof(Real), tid);
    dma_to_lds<BLOCK>(lds + L.par, g_par + c0, cn * (int)sizeof(CartPar<Real>), tid);
    dma_to_lds<BLOCK>(lds + L.wts, g_w + (size_t)c0 * leaf_n * dim, cn * leaf_n * dim * (int)sizeof(Real), tid);
    __builtin_amdgcn_s_waitcnt(0);                    // vmcnt(0): DMA (tile on the first pass, tables) has landed
    __syncthreads();
#pragma nounroll
    for (int kk = 0; kk < cn; kk += G) {
      if (__ballot(alive) == 0ull) break;
      if (!alive) continue;
      // ---- G trees at once (they do not depend on the score) ----
      int node[G], kc[G];
#pragma unroll
      for (int g = 0; g < G; g++) { node[g] = 0; kc[g] = min(kk + g, cn - 1); }
#pragma nounroll
      for (int d = 0; d < depth; d++) {
        Node nd[G];
#pragma unroll
        for (int g = 0; g < G; g++) nd[g] = t_nodes[kc[g] * node_n + node[g]];
        Real sv[G][4];
#pragma unroll
        for (int g = 0; g < G; g++) {
          sv[g][0] = sh[nd[g].lm1x2 * BLOCK + tid]; sv[g][1] = sh[(nd[g].lm1x2 + 1) * BLOCK + tid];
          sv[g][2] = sh[nd[g].lm2x2 * BLOCK + tid]; sv[g][3] = sh[(nd[g].lm2x2 + 1) * BLOCK + tid];
        }
        int pa[G], pb[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
          int x1, y1, x2, y2;
          DL::pixel_pair(sv[g][0], sv[g][1], nd[g].o1x, nd[g].o1y, win, &x1, &y1);
          DL::pixel_pair(sv[g][2], sv[g][3], nd[g].o2x, nd[g].o2y, win, &x2, &y2);
          pa[g] = pix[__umul24((unsigned)y1, (unsigned)ppitch) + (unsigned)x1 + (unsigned)base];   // 24-bit multiply: full rate
          pb[g] = pix[__umul24((unsigned)y2, (unsigned)ppitch) + (unsigned)x2 + (unsigned)base];
        }
#pragma unroll
        for (int g = 0; g < G; g++) node[g] = 2 * node[g] + ((pa[g] - pb[g] <= nd[g].th) ? 1 : 2);   // c/jda.c:391-393
      }
      // ---- scores strictly in cart order (c/jda.c:395-399); regression rows in cart order
      //      (c/jda.c:404-411) ----
      CartPar<Real> p[G];
      Real lsv[G];
      int lf[G];
#pragma unroll
      for (int g = 0; g < G; g++) { lf[g] = node[g] - node_n; p[g] = t_par[kc[g]]; lsv[g] = t_leaf[kc[g] * leaf_n + lf[g]]; }
#pragma unroll
      for (int g = 0; g < G; g++) {
        if (kk + g < cn) {                                                // wave-uniform
          const Vec2* row = (const Vec2*)(t_w + (size_t)((kk + g) * leaf_n + lf[g]) * dim);   // dim is even: aligned
#pragma unroll
          for (int d2 = 0; d2 < ACC / 2; d2++) acc[d2] = acc[d2] + row[d2];   // coordinates >= dim: slack, never stored
          if (alive) {
            Real sc = score + lsv[g];                                      // c/jda.c:396
            if (p[g].norm != (Real)0) sc = (sc - p[g].mean) / p[g].std;    // c/jda.c:397
            score = sc;
            if (TRACE) hash = fnv_step(hash, lf[g]);
            if (sc < p[g].th) { alive = false; carts_n = t * K + c0 + kk + g + 1; }   // c/jda.c:399
          }
        }
      }
    }
  }

  if (!entered) return;
  if (alive) {
    // stage passed: new shape (dialect CPP adds the identity-transformed delta once, btcart.cpp:407-424)
    Real* dst = w.m_shape + (size_t)gid * dim;
#pragma unroll
    for (int d2 = 0; d2 < ACC / 2; d2++) {
      if (2 * d2 < dim) {
        Real vx = acc[d2].x, vy = acc[d2].y;
        if (kCpp) {
          const Real zero = (Real)0, one = (Real)1;                       // stp_mc.Apply with the identity, data.hpp:42-45
          const Real ax = one * (one * vx + zero * vy), ay = one * (zero * vx + one * vy);
          vx = sh[(2 * d2) * BLOCK + tid] + ax;
          vy = sh[(2 * d2 + 1) * BLOCK + tid] + ay;
        }
        dst[2 * d2] = vx; dst[2 * d2 + 1] = vy;
        if (TRACE && t == T - 1) { w.tr_shape[(size_t)gid * dim + 2 * d2] = vx; w.tr_shape[(size_t)gid * dim + 2 * d2 + 1] = vy; }
      }
    }
    atomicAdd(shard_counter(w.counters, kCntStage0 + t), 1ull);
    if (t == T - 1) carts_n = T * K;
  } else if (TRACE) {
    for (int d = 0; d < dim; d++) w.tr_shape[(size_t)gid * dim + d] = sh[d * BLOCK + tid];
  }
  w.m_score[gid] = score;
  if (TRACE) w.m_hash[gid] = hash;
  w.st_carts[gid] = carts_n;
  if (carts_n >= 0) {
    atomicAdd(shard_counter(w.counters, kCntCarts), (unsigned long long)carts_n);
    if (TRACE) { w.tr_carts[gid] = carts_n; w.tr_score[gid] = score; w.tr_hash[gid] = hash; }
    if (alive && !(apply_th && score < final_th)) {                       // c/jda.c:414
      const unsigned o = (unsigned)atomicAdd(&w.counters[kCntOut], 1ull);
      if (o < w.cap) {
        w.out_gid[o] = gid; w.out_score[o] = score;
        const Real* src = w.m_shape + (size_t)gid * dim;
        for (int d = 0; d < dim; d++) w.out_shape[(size_t)o * dim + d] = src[d];
      }
    }
  }
}

namespace {
inline int dense_acc(int dim) { return dim <= 12 ? 12 : dim <= 32 ? 32 : dim <= 64 ? 64 : 160; }

// Cart chunk and LDS budget of one k_stage launch.  The walk is latency bound (measured: LDS
// and VALU are each under half busy), so resident workgroups count more than chunk length:
// 16-cart chunks cost nothing against 64, 4-cart chunks ~10 %.  Tiers = 5, 4, 3, 2, 1 workgroups
// per CU (160 KB of LDS): first the smallest tier that fits a 16-cart chunk among the tiers
// with >= 3 workgroups, else the smallest tier that fits any chunk (two resident workgroups
// with 4-cart chunks beat one with 32-cart chunks).  lds_max caps the tiers (0: nothing fits).
template <typename Real, typename Node>
int dense_chunk(int pix_bytes, int dim, int node_n, int leaf_n, int K, int lds_max) {
  const int acc = dense_acc(dim);
  const int tiers[5] = {32768, 40960, 53248, 81920, 163840};
  auto fits = [&](int ch, int budget) {
    return DenseLds<Real, Node>(pix_bytes, dim, node_n, leaf_n, ch, acc).total <= (budget < lds_max ? budget : lds_max);
  };
  for (int ti = 0; ti < 3; ti++) {
    if (!fits(16, tiers[ti])) continue;
    int ch = 16;
    while (ch < 64 && fits(ch * 2, tiers[ti])) ch *= 2;
    return ch;
  }
  for (int ti = 0; ti < 5; ti++)
    for (int ch = 64; ch >= 4; ch >>= 1)
      if (fits(ch, tiers[ti])) return ch;
  return 0;
}

template <typename DL>
hipError_t launch_stage_impl(bool trace, int level, int t, bool apply_th, typename DL::Real th, const DevPlan* d_plan,
                             const DevPlan& h_plan, const DevModelT<typename DL::Real>& m,
                             const WorkT<typename DL::Real>& w, int pix_cap, int lds_max, hipStream_t stream) {
  using Real = typename DL::Real;
  using Node = typename DL::Node;
  const DevLevel& lv = h_plan.lv[level];
  const int pw = lv.win + (kDenseTw - 1) * lv.step, ph = lv.win + (kDenseTh - 1) * lv.step;
  int pitch = (pw + 15) & ~15;
  if ((pitch & 127) == 0) pitch += 16;
  const long long need = (long long)pitch * ph;
  const bool glb = need > pix_cap;
  const int pix_bytes = glb ? 0 : (int)need;
  const int acc = dense_acc(m.dim);
  int chunk = dense_chunk<Real, Node>(pix_bytes, m.dim, m.node_n, m.leaf_n, m.K, lds_max);
  if (chunk == 0) return hipErrorInvalidValue;
  if (const char* e = getenv("JDA_DENSE_CHUNK")) chunk = std::max(4, std::min(chunk, atoi(e)));   // experiments
  const DenseLds<Real, Node> L(pix_bytes, m.dim, m.node_n, m.leaf_n, chunk, acc);
  const int tiles = ((lv.nx + kDenseTw - 1) / kDenseTw) * ((lv.ny + kDenseTh - 1) / kDenseTh);
  const int groups = (w.n_frames + 7) / 8;
  dim3 grid((unsigned)(groups * 8 * tiles)), block(kDenseM);
  auto go = [&](auto kern) {
    if (L.total > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    hipLaunchKernelGGL(kern, grid, block, L.total, stream, d_plan, m, w, level, t, pix_bytes, pitch, chunk,
                       apply_th ? 1 : 0, th);
  };
  auto pick = [&](auto trace_tag, auto glb_tag) {
    constexpr bool TR = decltype(trace_tag)::value, GL = decltype(glb_tag)::value;
    switch (acc) {
      case 12: go(k_stage<DL, TR, GL, 12>); break;
      case 32: go(k_stage<DL, TR, GL, 32>); break;
      case 64: go(k_stage<DL, TR, GL, 64>); break;
      default: go(k_stage<DL, TR, GL, 160>); break;
    }
  };
  if (trace) { if (glb) pick(std::true_type{}, std::true_type{}); else pick(std::true_type{}, std::false_type{}); }
  else { if (glb) pick(std::false_type{}, std::true_type{}); else pick(std::false_type{}, std::false_type{}); }
  return hipGetLastError();
}
}  // namespace

template <>
hipError_t launch_stage<float>(bool trace, int level, int t, bool apply_final_th, float final_th, const DevPlan* d_plan,
                               const DevPlan& h_plan, const DevModelT<float>& m, const WorkT<float>& w, int pix_cap,
                               int lds_max, hipStream_t stream) {
  return launch_stage_impl<DialectC>(trace, level, t, apply_final_th, final_th, d_plan, h_plan, m, w, pix_cap, lds_max, stream);
}
template <>
hipError_t launch_stage<double>(bool trace, int level, int t, bool apply_final_th, double final_th, const DevPlan* d_plan,
                                const DevPlan& h_plan, const DevModelT<double>& m, const WorkT<double>& w, int pix_cap,
                                int lds_max, hipStream_t stream) {
  return launch_stage_impl<DialectCPP>(trace, level, t, apply_final_th, final_th, d_plan, h_plan, m, w, pix_cap, lds_max, stream);
}

// LDS bytes of k_stage with no pixel tile and the smallest chunk: the floor the host checks a model against
size_t stage_lds_bytes(int dim, int node_n, int leaf_n, int real_bytes) {
  return real_bytes == 4 ? (size_t)DenseLds<float, NodeF>(0, dim, node_n, leaf_n, 4, dense_acc(dim)).total
                         : (size_t)DenseLds<double, NodeD>(0, dim, node_n, leaf_n, 4, dense_acc(dim)).total;
}

// =============================================================================
// trace defaults: every window starts as "0 carts, mean shape"
// =============================================================================

template <typename Real>
__global__ void k_trace_fill(DevModelT<Real> m, WorkT<Real> w, unsigned n_windows) {
  const unsigned long long total = (unsigned long long)n_windows * m.dim;
  for (unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (unsigned long long)gridDim.x * blockDim.x) {
    const int d = (int)(idx % m.dim);
    w.tr_shape[idx] = m.mean_shape[d];
  }
}

template <typename Real>
hipError_t launch_trace_fill(const DevModelT<Real>& m, const WorkT<Real>& w, unsigned n_windows, hipStream_t stream) {
  hipLaunchKernelGGL(k_trace_fill<Real>, dim3(1024), dim3(256), 0, stream, m, w, n_windows);
  return hipGetLastError();
}
template hipError_t launch_trace_fill<float>(const DevModelT<float>&, const WorkT<float>&, unsigned, hipStream_t);
template hipError_t launch_trace_fill<double>(const DevModelT<double>&, const WorkT<double>&, unsigned, hipStream_t);

}  // namespace jda
