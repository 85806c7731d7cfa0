// HIP kernels of the JDA detect path for gfx950 (MI355X, wave64).
//
// Built with -ffp-contract=off -fno-gpu-flush-denormals-to-zero: every fp
// operation below must round exactly like the reference's scalar C/C++ (no FMA,
// IEEE division, denormals kept).
//
//   k_resize        bilinear pyramid image        reference c/jda.c:203-230
//   k_prep_stage0   stage-0 feature offsets per level (hoisted c/jda.c:370-389)
//   k_scan          stage-0 cascade walk, LDS tile, lane = window, survivors
//                   compacted by ballot/prefix-sum every chunk of carts
//                                                  reference c/jda.c:357-402
//   k_walk          generic walker (any stage; per-window shape; HBM/L2 pixels)
//                                                  c/jda.c:364-402, cascador.cpp:166-192
//   k_update        stage regression: W-row gather in cart order
//                                                  c/jda.c:404-411, btcart.cpp:407-424
//   k_pack          final detections -> contiguous rows
#include <hip/hip_runtime.h>

#include <climits>

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
};

struct DialectCPP {         // reference src/jda (Validate / CalcFeatureValue)
  using Real = double;
  using Node = NodeD;
  // data.cpp:40-47: fp64, round half away from zero
  static __device__ __forceinline__ int coord(double s, double o, int win) {
    const double v = (s + o) * (double)win;
    return to_int_x86(round(v));
  }
};

__device__ __forceinline__ int wave_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ unsigned long long lanes_below(int lane) {
  return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// level of a frame-local window index (uniform or per-lane; levels are few)
__device__ __forceinline__ int find_level(const DevPlan* plan, int wid) {
  int l = 0;
  const int n = plan->n_levels;
  for (int i = 1; i < n; i++)
    if (wid >= plan->lv[i].base) l = i;
  return l;
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

// =============================================================================
// stage-0 offset table
// =============================================================================

template <typename DL>
__global__ void k_prep_stage0(const DevPlan* __restrict__ plan, const typename DL::Node* __restrict__ nodes,
                              const typename DL::Real* __restrict__ mean_shape, int K, int node_n,
                              S0Node* __restrict__ table) {
  const int l = blockIdx.y;
  const DevLevel lv = plan->lv[l];
  if (lv.tile_class == kTileNone) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * node_n) return;
  const typename DL::Node nd = nodes[i];  // stage 0 occupies the first K*node_n nodes
  const int win = lv.win;
  int x1 = clamp_win(DL::coord(mean_shape[nd.lm1x2], nd.o1x, win), win);
  int y1 = clamp_win(DL::coord(mean_shape[nd.lm1x2 + 1], nd.o1y, win), win);
  int x2 = clamp_win(DL::coord(mean_shape[nd.lm2x2], nd.o2x, win), win);
  int y2 = clamp_win(DL::coord(mean_shape[nd.lm2x2 + 1], nd.o2y, win), win);
  S0Node o;
  o.offs = (uint32_t)(y1 * lv.pitch + x1) | ((uint32_t)(y2 * lv.pitch + x2) << 16);
  // feature is a difference of two bytes: thresholds beyond [-256,255] behave like the ends
  o.th = nd.th < -256 ? -256 : (nd.th > 255 ? 255 : nd.th);
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

int scan_chunk_max(int node_n, int leaf_n) {
  // LDS table chunk: at most 8 KiB of S0Node + leaf scores (sized for f64)
  int c = 8192 / (node_n * (int)sizeof(S0Node) + leaf_n * 8);
  if (c > 64) c = 64;
  if (c < 2) c = 2;
  return c & ~1;
}

namespace {

template <typename Real, bool TRACE>
struct ScanLds {
  // byte offsets inside dynamic LDS
  int pix, nodes, leaf, q_widx, q_score, q_hash, misc, total;
  __host__ __device__ ScanLds(int pix_budget, int chunk_max, int node_n, int leaf_n, int m_max) {
    int o = 0;
    pix = o; o += (pix_budget + 15) & ~15;
    nodes = o; o += chunk_max * node_n * (int)sizeof(S0Node); o = (o + 15) & ~15;
    leaf = o; o += chunk_max * leaf_n * (int)sizeof(Real); o = (o + 15) & ~15;
    q_score = o; o += 2 * m_max * (int)sizeof(Real);
    q_widx = o; o += 2 * m_max * 2; o = (o + 15) & ~15;
    q_hash = o; if (TRACE) o += 2 * m_max * 4;
    misc = o; o += 64;
    total = o;
  }
};

}  // namespace

size_t scan_lds_bytes(int pix_bytes, int node_n, int leaf_n, int real_bytes, bool trace, int tile_class) {
  const int cm = scan_chunk_max(node_n, leaf_n);
  const int mm = tile_class == kTileWide ? 512 : 64;
  if (real_bytes == 4) return trace ? ScanLds<float, true>(pix_bytes, cm, node_n, leaf_n, mm).total
                                    : ScanLds<float, false>(pix_bytes, cm, node_n, leaf_n, mm).total;
  return trace ? ScanLds<double, true>(pix_bytes, cm, node_n, leaf_n, mm).total
               : ScanLds<double, false>(pix_bytes, cm, node_n, leaf_n, mm).total;
}

// One cart of stage 0 for one window: D-1 dependent (node, 2 pixels) LDS reads.
template <int DEPTH>
__device__ __forceinline__ int scan_tree(const S0Node* __restrict__ tbl, const uint8_t* __restrict__ pix,
                                         int base, int depth_rt) {
  int node = 0;
  const int levels = DEPTH > 0 ? DEPTH - 1 : depth_rt - 1;
#pragma unroll
  for (int d = 0; d < levels; d++) {
    const S0Node r = tbl[node];
    const int a = pix[base + (int)(r.offs & 0xffffu)];
    const int b = pix[base + (int)(r.offs >> 16)];
    node = 2 * node + ((a - b <= r.th) ? 1 : 2);     // c/jda.c:391-393
  }
  return node;
}

template <typename Real, int BLOCK, int DEPTH, bool TRACE>
__global__ __launch_bounds__(BLOCK) void k_scan(const DevPlan* __restrict__ plan, DevModelT<Real> m,
                                                const S0Node* __restrict__ table, WorkT<Real> w,
                                                int level, int pix_budget, int chunk_max) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int M_MAX = BLOCK == 256 ? 512 : 64;
  const int node_n = m.node_n, leaf_n = m.leaf_n, K = m.K;
  const ScanLds<Real, TRACE> L(pix_budget, chunk_max, node_n, leaf_n, M_MAX);
  uint8_t* pix = lds + L.pix;
  S0Node* t_nodes = (S0Node*)(lds + L.nodes);
  Real* t_leaf = (Real*)(lds + L.leaf);
  Real* q_score = (Real*)(lds + L.q_score);
  uint16_t* q_widx = (uint16_t*)(lds + L.q_widx);
  unsigned* q_hash = (unsigned*)(lds + L.q_hash);
  int* misc = (int*)(lds + L.misc);   // [0],[1] queue counts; [2] global base

  const int tid = threadIdx.x;
  const int lane = tid & 63;

  // XCD-aware block -> (frame, tile): blocks b, b+8, b+16.. land on one XCD
  // (MI355X dispatches block b to XCD b % 8), so the 8 frames of a group each
  // stay inside one XCD's L2.
  const DevLevel lv = plan->lv[level];
  const int tiles_per_frame = lv.tiles_x * lv.tiles_y;
  const int b = blockIdx.x;
  const int group = b / (8 * tiles_per_frame);
  const int r = b - group * (8 * tiles_per_frame);
  const int frame = group * 8 + (r & 7);
  const int trel = r >> 3;
  if (frame >= w.n_frames) return;
  const int ty = trel / lv.tiles_x, tx = trel - ty * lv.tiles_x;
  const int wx0 = tx * lv.tw, wy0 = ty * lv.th;                 // first window of the tile
  const int twe = min(lv.tw, lv.nx - wx0), the = min(lv.th, lv.ny - wy0);
  const int x0 = wx0 * lv.step, y0 = wy0 * lv.step;             // tile origin in the frame
  const int pw = lv.win + (twe - 1) * lv.step, ph = lv.win + (the - 1) * lv.step;
  const uint8_t* img = w.frames + (size_t)frame * w.frame_stride;

  // ---- stage the pixel tile: coalesced dword rows when alignment allows ----
  const int W = plan->width;
  const bool al4 = ((W & 3) == 0) && ((w.frame_stride & 3) == 0) && ((((uintptr_t)w.frames) & 3) == 0);
  const int x0a = al4 ? (x0 & ~3) : x0;
  const int xshift = x0 - x0a;
  {
    constexpr int NW = BLOCK / 64;
    const int wv = tid >> 6;
    if (al4) {
      const int ndw = (xshift + pw + 3) >> 2;
      for (int rr = wv; rr < ph; rr += NW) {
        const uint32_t* g = (const uint32_t*)(img + (size_t)(y0 + rr) * W + x0a);
        uint32_t* d = (uint32_t*)(pix + rr * lv.pitch);
        for (int c = lane; c < ndw; c += 64) d[c] = g[c];
      }
    } else {
      for (int rr = wv; rr < ph; rr += NW) {
        const uint8_t* g = img + (size_t)(y0 + rr) * W + x0;
        uint8_t* d = pix + rr * lv.pitch;
        for (int c = lane; c < pw; c += 64) d[c] = g[c];
      }
    }
  }

  const int n_tile = lv.tw * lv.th;      // phase 0 enumerates the full tile; edge windows are filtered
  int n_items = n_tile;
  int cur = 0;
  unsigned my_carts = 0;
  const int gid0 = frame * plan->windows + lv.base;

  for (int c0 = 0; c0 < K;) {
    int len = c0 < 8 ? 8 : c0;
    if (len > chunk_max) len = chunk_max;
    const int c1 = min(K, c0 + len);
    // ---- stage this chunk's tables ----
    {
      const S0Node* src = table + lv.s0_table + c0 * node_n;
      for (int i = tid; i < (c1 - c0) * node_n; i += BLOCK) t_nodes[i] = src[i];
      const Real* ls = m.leaf + c0 * leaf_n;
      for (int i = tid; i < (c1 - c0) * leaf_n; i += BLOCK) t_leaf[i] = ls[i];
      if (tid == 0) misc[cur ^ 1] = 0;
    }
    __syncthreads();

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

      int k = c0;
      for (; k + 1 < c1; k += 2) {
        if (__ballot(alive) == 0ull) break;
        if (alive) {
          const S0Node* tn = t_nodes + (k - c0) * node_n;
          const int na = scan_tree<DEPTH>(tn, pix, base, m.D);
          const int nb = scan_tree<DEPTH>(tn + node_n, pix, base, m.D);
          const int la = na - node_n, lb = nb - node_n;
          Real s = score + t_leaf[(k - c0) * leaf_n + la];                // c/jda.c:396
          if (m.cnorm[k]) s = (s - m.cmean[k]) / m.cstd[k];               // c/jda.c:397
          if (TRACE) hash = fnv_step(hash, la);
          bool dead = s < m.cth[k];                                       // c/jda.c:399
          int kd = k;
          if (!dead) {
            s = s + t_leaf[(k + 1 - c0) * leaf_n + lb];
            if (m.cnorm[k + 1]) s = (s - m.cmean[k + 1]) / m.cstd[k + 1];
            if (TRACE) hash = fnv_step(hash, lb);
            dead = s < m.cth[k + 1];
            kd = k + 1;
          }
          score = s;
          if (dead) {
            alive = false;
            my_carts += kd + 1;
            if (TRACE) {
              const int gid = gid0 + (wy0 + wy) * lv.nx + wx0 + wx;
              w.tr_carts[gid] = kd + 1; w.tr_score[gid] = s; w.tr_hash[gid] = hash;
            }
          }
        }
      }
      if (k < c1 && k + 1 >= c1) {   // odd tail cart
        if (alive) {
          const int na = scan_tree<DEPTH>(t_nodes + (k - c0) * node_n, pix, base, m.D);
          const int la = na - node_n;
          Real s = score + t_leaf[(k - c0) * leaf_n + la];
          if (m.cnorm[k]) s = (s - m.cmean[k]) / m.cstd[k];
          if (TRACE) hash = fnv_step(hash, la);
          score = s;
          if (s < m.cth[k]) {
            alive = false;
            my_carts += k + 1;
            if (TRACE) {
              const int gid = gid0 + (wy0 + wy) * lv.nx + wx0 + wx;
              w.tr_carts[gid] = k + 1; w.tr_score[gid] = s; w.tr_hash[gid] = hash;
            }
          }
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
    __syncthreads();
    cur ^= 1;
    n_items = misc[cur];
    c0 = c1;
    if (n_items == 0) break;
  }

  // ---- survivors of all K carts of stage 0 -> global queue 0 ----
  if (n_items > 0) {
    if (tid == 0) misc[2] = (int)atomicAdd(&w.counters[kCntQueue0], (unsigned long long)n_items);
    __syncthreads();
    const unsigned gbase = (unsigned)misc[2];
    for (int i = tid; i < n_items; i += BLOCK) {
      const int widx = q_widx[cur * M_MAX + i];
      const int wy = widx / lv.tw, wx = widx - wy * lv.tw;
      const unsigned slot = gbase + i;
      if (slot < w.cap) {
        w.q_gid[0][slot] = (uint32_t)(gid0 + (wy0 + wy) * lv.nx + wx0 + wx);
        w.q_score[0][slot] = q_score[cur * M_MAX + i];
        w.q_src[0][slot] = 0;
        if (TRACE) w.q_hash[0][slot] = q_hash[cur * M_MAX + i];
      }
      my_carts += K;
    }
  }
  // ---- carts-evaluated counter (DetectionStatisic.cart_gothrough_n) ----
  unsigned v = my_carts;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane == 0 && v) {
    atomicAdd(&w.counters[kCntCarts], (unsigned long long)v);
    atomicAdd(&w.counters[kCntCartsScan], (unsigned long long)v);
  }
  if (tid == 0) atomicAdd(&w.counters[kCntWinScan], (unsigned long long)(twe * the));
}

template <typename Real, int BLOCK, bool TRACE>
static hipError_t launch_scan_depth(const DevPlan* d_plan, const DevPlan& h_plan, const DevModelT<Real>& m,
                                    const S0Node* table, const WorkT<Real>& w, int level, hipStream_t stream) {
  const DevLevel& lv = h_plan.lv[level];
  const int cm = scan_chunk_max(m.node_n, m.leaf_n);
  const int pix_bytes = lv.pitch * (lv.win + (lv.th - 1) * lv.step);
  const ScanLds<Real, TRACE> L(pix_bytes, cm, m.node_n, m.leaf_n, BLOCK == 256 ? 512 : 64);
  const int groups = (w.n_frames + 7) / 8;
  dim3 grid((unsigned)(groups * 8 * lv.tiles_x * lv.tiles_y)), block(BLOCK);
  auto go = [&](auto kern) {
    if (L.total > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    hipLaunchKernelGGL(kern, grid, block, L.total, stream, d_plan, m, table, w, level, pix_bytes, cm);
  };
  if (m.D == 4) go(k_scan<Real, BLOCK, 4, TRACE>);
  else if (m.D == 6) go(k_scan<Real, BLOCK, 6, TRACE>);
  else go(k_scan<Real, BLOCK, 0, TRACE>);
  return hipGetLastError();
}

template <typename Real>
hipError_t launch_scan(int level, bool trace, const DevPlan* d_plan, const DevPlan& h_plan,
                       const DevModelT<Real>& m, const S0Node* table, const WorkT<Real>& w,
                       hipStream_t stream) {
  const int tile_class = h_plan.lv[level].tile_class;
  if (tile_class == kTileNone || w.n_frames == 0) return hipSuccess;
  if (tile_class == kTileWide) {
    return trace ? launch_scan_depth<Real, 256, true>(d_plan, h_plan, m, table, w, level, stream)
                 : launch_scan_depth<Real, 256, false>(d_plan, h_plan, m, table, w, level, stream);
  }
  return trace ? launch_scan_depth<Real, 64, true>(d_plan, h_plan, m, table, w, level, stream)
               : launch_scan_depth<Real, 64, false>(d_plan, h_plan, m, table, w, level, stream);
}

template hipError_t launch_scan<float>(int, bool, const DevPlan*, const DevPlan&, const DevModelT<float>&,
                                       const S0Node*, const WorkT<float>&, hipStream_t);
template hipError_t launch_scan<double>(int, bool, const DevPlan*, const DevPlan&, const DevModelT<double>&,
                                        const S0Node*, const WorkT<double>&, hipStream_t);

// =============================================================================
// queue of windows the scan does not cover
// =============================================================================

template <typename Real>
__global__ void k_enqueue_generic(const DevPlan* __restrict__ plan, WorkT<Real> w, int per_frame, int all_levels) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)per_frame * w.n_frames;
  if (idx == 0) w.counters[kCntGeneric] = (unsigned long long)total;
  if (idx >= total) return;
  const int frame = (int)(idx / per_frame);
  int r = (int)(idx - (long long)frame * per_frame);
  int wid = -1;
  for (int i = 0; i < plan->n_levels; i++) {
    const DevLevel* c = &plan->lv[i];
    if (!all_levels && c->tile_class != kTileNone) continue;
    const int cnt = c->nx * c->ny;
    if (wid < 0 && r < cnt) wid = c->base + r;
    r -= cnt;
  }
  w.qg_gid[idx] = (uint32_t)(frame * plan->windows + wid);
}

template <typename Real>
hipError_t launch_enqueue_generic(const DevPlan* d_plan, const DevPlan& h_plan, bool all_levels,
                                  const WorkT<Real>& w, hipStream_t stream) {
  long long per_frame = 0;
  for (int i = 0; i < h_plan.n_levels; i++)
    if (all_levels || h_plan.lv[i].tile_class == kTileNone) per_frame += (long long)h_plan.lv[i].nx * h_plan.lv[i].ny;
  const long long total = per_frame * w.n_frames;
  if (total == 0) return hipSuccess;
  dim3 block(256), grid((unsigned)((total + 255) / 256));
  hipLaunchKernelGGL(k_enqueue_generic<Real>, grid, block, 0, stream, d_plan, w, (int)per_frame, all_levels ? 1 : 0);
  return hipGetLastError();
}
template hipError_t launch_enqueue_generic<float>(const DevPlan*, const DevPlan&, bool, const WorkT<float>&, hipStream_t);
template hipError_t launch_enqueue_generic<double>(const DevPlan*, const DevPlan&, bool, const WorkT<double>&, hipStream_t);

// =============================================================================
// generic walker: one stage, lane = window, per-window shape in LDS
// =============================================================================

namespace {

// Where a window reads its pixels for one feature scale.
struct View {
  const uint8_t* img; int w, h, ox, oy;
};

template <typename DL>
__device__ __forceinline__ int node_feature(const typename DL::Node& nd, const typename DL::Real* sh, int sh_stride,
                                            int sh_lane, int win, const View& v0, const View& v1, const View& v2,
                                            bool multi) {
  using Real = typename DL::Real;
  const Real s1x = sh[nd.lm1x2 * sh_stride + sh_lane], s1y = sh[(nd.lm1x2 + 1) * sh_stride + sh_lane];
  const Real s2x = sh[nd.lm2x2 * sh_stride + sh_lane], s2y = sh[(nd.lm2x2 + 1) * sh_stride + sh_lane];
  const int x1 = clamp_win(DL::coord(s1x, nd.o1x, win), win);
  const int y1 = clamp_win(DL::coord(s1y, nd.o1y, win), win);
  const int x2 = clamp_win(DL::coord(s2x, nd.o2x, win), win);
  const int y2 = clamp_win(DL::coord(s2y, nd.o2y, win), win);
  if (!multi) {
    const int a = v0.img[(size_t)(v0.oy + y1) * v0.w + v0.ox + x1];
    const int b = v0.img[(size_t)(v0.oy + y2) * v0.w + v0.ox + x2];
    return a - b;
  }
  // scale != 0: the reference indexes the half/quarter image with full-window
  // coordinates (c/jda.c:347-354) and may leave it; reads are clamped to the image.
  const View& v = nd.scale == 0 ? v0 : (nd.scale == 1 ? v1 : v2);
  const int gx1 = min(v.ox + x1, v.w - 1), gy1 = min(v.oy + y1, v.h - 1);
  const int gx2 = min(v.ox + x2, v.w - 1), gy2 = min(v.oy + y2, v.h - 1);
  const int a = v.img[(size_t)gy1 * v.w + gx1];
  const int b = v.img[(size_t)gy2 * v.w + gx2];
  return a - b;
}

template <typename Real>
__device__ __forceinline__ void decode_window(const DevPlan* plan, const WorkT<Real>& w, uint32_t gid, float inv_sqrt2,
                                              int* win, View* v0, View* v1, View* v2, bool multi) {
  const int frame = (int)(gid / (uint32_t)plan->windows);
  const int wid = (int)(gid - (uint32_t)frame * (uint32_t)plan->windows);
  const int l = find_level(plan, wid);
  const DevLevel* lv = &plan->lv[l];
  const int rel = wid - lv->base;
  const int iy = rel / lv->nx, ix = rel - iy * lv->nx;
  const int x = ix * lv->step, y = iy * lv->step;
  *win = lv->win;
  v0->img = w.frames + (size_t)frame * w.frame_stride; v0->w = plan->width; v0->h = plan->height; v0->ox = x; v0->oy = y;
  if (multi) {
    v1->img = w.half + (size_t)frame * w.half_stride; v1->w = w.hw; v1->h = w.hh;
    v1->ox = (int)((float)x * inv_sqrt2); v1->oy = (int)((float)y * inv_sqrt2);      // c/jda.c:345-346
    v2->img = w.quarter + (size_t)frame * w.quarter_stride; v2->w = w.qw; v2->h = w.qh;
    v2->ox = x / 2; v2->oy = y / 2;                                                    // c/jda.c:351-352
  }
}

}  // namespace

template <typename DL, bool TRACE>
__global__ __launch_bounds__(64) void k_walk(const DevPlan* __restrict__ plan, DevModelT<typename DL::Real> m,
                                             WorkT<typename DL::Real> w, int t, int multi_i, float inv_sqrt2) {
  using Real = typename DL::Real;
  using Node = typename DL::Node;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  Real* sh = (Real*)lds;  // [dim][64]
  const bool multi = multi_i != 0;
  const int lane = threadIdx.x;
  const int K = m.K, node_n = m.node_n, leaf_n = m.leaf_n, dim = m.dim;
  const int pin = (t + 1) & 1, pout = t & 1;   // parity of queue t-1 / queue t
  const unsigned n_in = (unsigned)(t == 0 ? w.counters[kCntGeneric] : w.counters[kCntQueue0 + t - 1]);
  const uint32_t* in_gid = t == 0 ? w.qg_gid : w.q_gid[pin];
  const Node* nodes = (const Node*)m.nodes + (size_t)t * K * node_n;
  const Real* leaf_tab = m.leaf + (size_t)t * K * leaf_n;
  const Real* cth = m.cth + (size_t)t * K;
  const Real* cmean = m.cmean + (size_t)t * K;
  const Real* cstd = m.cstd + (size_t)t * K;
  const uint8_t* cnorm = m.cnorm + (size_t)t * K;
  unsigned my_carts = 0;

  for (unsigned s0 = blockIdx.x * 64u; s0 < n_in; s0 += gridDim.x * 64u) {
    const unsigned slot = s0 + lane;
    bool alive = slot < n_in;
    uint32_t gid = 0;
    Real score = 0;
    unsigned hash = kFnvSeed;
    int win = 24;
    View v0{}, v1{}, v2{};
    if (alive) {
      gid = in_gid[slot];
      if (t > 0) { score = w.q_score[pin][slot]; if (TRACE) hash = w.q_hash[pin][slot]; }
      decode_window<Real>(plan, w, gid, inv_sqrt2, &win, &v0, &v1, &v2, multi);
    }
    __syncthreads();   // previous iteration's readers are done with sh
    if (alive) {
      if (t == 0) {
        for (int d = 0; d < dim; d++) sh[d * 64 + lane] = m.mean_shape[d];
      } else {
        const Real* src = w.shape[pin] + (size_t)slot * dim;
        for (int d = 0; d < dim; d++) sh[d * 64 + lane] = src[d];
      }
    }
    __syncthreads();

    for (int k = 0; k < K; k++) {
      if (__ballot(alive) == 0ull) break;
      if (alive) {
        int node = 0;
        for (int d = 0; d < m.D - 1; d++) {
          const Node nd = nodes[(size_t)k * node_n + node];
          const int feat = node_feature<DL>(nd, sh, 64, lane, win, v0, v1, v2, multi);
          node = 2 * node + (feat <= nd.th ? 1 : 2);
        }
        const int lf = node - node_n;
        Real s = score + leaf_tab[(size_t)k * leaf_n + lf];
        if (cnorm[k]) s = (s - cmean[k]) / cstd[k];
        score = s;
        if (TRACE) hash = fnv_step(hash, lf);
        if (s < cth[k]) {
          alive = false;
          my_carts += k + 1;
          if (TRACE) {
            w.tr_carts[gid] = t * K + k + 1; w.tr_score[gid] = s; w.tr_hash[gid] = hash;
            if (t > 0)
              for (int d = 0; d < dim; d++) w.tr_shape[(size_t)gid * dim + d] = sh[d * 64 + lane];
          }
        }
      }
    }
    if (alive) my_carts += K;
    // survivors -> queue t (one atomic per wave, order inside the wave kept)
    const unsigned long long mask = __ballot(alive);
    if (mask) {
      unsigned long long wbase = 0;
      if (lane == 0) wbase = atomicAdd(&w.counters[kCntQueue0 + t], (unsigned long long)__popcll(mask));
      wbase = __shfl(wbase, 0);
      if (alive) {
        const unsigned o = (unsigned)wbase + __popcll(mask & lanes_below(lane));
        if (o < w.cap) {
          w.q_gid[pout][o] = gid; w.q_score[pout][o] = score; w.q_src[pout][o] = slot;
          if (TRACE) w.q_hash[pout][o] = hash;
        }
      }
    }
  }
  unsigned v = my_carts;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane == 0 && v) atomicAdd(&w.counters[kCntCarts], (unsigned long long)v);
}

template <typename Real>
hipError_t launch_walk(int dialect, bool trace, int t, const DevPlan* d_plan, const DevModelT<Real>& m,
                       const WorkT<Real>& w, hipStream_t stream);

namespace {
template <typename DL>
hipError_t launch_walk_impl(bool trace, int t, const DevPlan* d_plan, const DevModelT<typename DL::Real>& m,
                            const WorkT<typename DL::Real>& w, hipStream_t stream) {
  const size_t lds = (size_t)m.dim * 64 * sizeof(typename DL::Real);
  const int multi = (w.half != nullptr) ? 1 : 0;
  const float r = 1.f / sqrtf(2.f);
  unsigned blocks = (w.cap + 63) / 64;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks == 0) blocks = 1;
  auto go = [&](auto kern) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, stream, d_plan, m, w, t, multi, r);
  };
  if (trace) go(k_walk<DL, true>); else go(k_walk<DL, false>);
  return hipGetLastError();
}
}  // namespace

template <>
hipError_t launch_walk<float>(int dialect, bool trace, int t, const DevPlan* d_plan, const DevModelT<float>& m,
                              const WorkT<float>& w, hipStream_t stream) {
  (void)dialect;
  return launch_walk_impl<DialectC>(trace, t, d_plan, m, w, stream);
}
template <>
hipError_t launch_walk<double>(int dialect, bool trace, int t, const DevPlan* d_plan, const DevModelT<double>& m,
                               const WorkT<double>& w, hipStream_t stream) {
  (void)dialect;
  return launch_walk_impl<DialectCPP>(trace, t, d_plan, m, w, stream);
}

// =============================================================================
// stage regression (+ final threshold)
// =============================================================================

template <typename DL, bool TRACE>
__global__ __launch_bounds__(256) void k_update(const DevPlan* __restrict__ plan, DevModelT<typename DL::Real> m,
                                                WorkT<typename DL::Real> w, int t, int multi_i, float inv_sqrt2,
                                                int is_last, int apply_th, typename DL::Real final_th,
                                                uint32_t* __restrict__ out_slot) {
  using Real = typename DL::Real;
  using Node = typename DL::Node;
  constexpr bool kCpp = sizeof(Real) == 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int K = m.K, node_n = m.node_n, leaf_n = m.leaf_n, dim = m.dim;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int dim_pad = (dim + 1) & ~1;
  Real* sh = (Real*)lds + wv * dim_pad;                                     // [4][dim]
  uint16_t* lbf = (uint16_t*)(lds + 4 * dim_pad * sizeof(Real)) + wv * ((K + 7) & ~7);  // [4][K]
  const bool multi = multi_i != 0;
  const int pcur = t & 1, pprev = (t + 1) & 1;
  const unsigned n = (unsigned)w.counters[kCntQueue0 + t];
  const Node* nodes = (const Node*)m.nodes + (size_t)t * K * node_n;
  const Real* wt = m.w + (size_t)t * K * leaf_n * dim;

  for (unsigned s0 = blockIdx.x * 4u; s0 < n; s0 += gridDim.x * 4u) {
    const unsigned slot = s0 + wv;
    const bool has = slot < n;     // wave-uniform
    uint32_t gid = 0;
    int win = 24;
    View v0{}, v1{}, v2{};
    __syncthreads();
    if (has) {
      gid = w.q_gid[pcur][slot];
      decode_window<Real>(plan, w, gid, inv_sqrt2, &win, &v0, &v1, &v2, multi);
      const Real* src = t == 0 ? m.mean_shape : w.shape[pprev] + (size_t)w.q_src[pcur][slot] * dim;
      for (int d = lane; d < dim; d += 64) sh[d] = src[d];
    }
    __syncthreads();
    if (has) {
      // leaves of this stage, one cart per lane (shape is fixed during a stage)
      for (int k = lane; k < K; k += 64) {
        int node = 0;
        for (int d = 0; d < m.D - 1; d++) {
          const Node nd = nodes[(size_t)k * node_n + node];
          const int feat = node_feature<DL>(nd, sh, 1, 0, win, v0, v1, v2, multi);
          node = 2 * node + (feat <= nd.th ? 1 : 2);
        }
        lbf[k] = (uint16_t)(node - node_n);
      }
    }
    __syncthreads();
    if (has) {
      const Real score = w.q_score[pcur][slot];
      const bool emit = is_last && !(apply_th && score < final_th);   // c/jda.c:414
      for (int d = lane; d < dim; d += 64) {
        // rows added strictly in cart order (c/jda.c:404-411). Dialect CPP sums
        // the delta from zero and adds it once (btcart.cpp:407-424).
        Real acc = kCpp ? (Real)0 : sh[d];
        const Real* col = wt + d;
        int k = 0;
        for (; k + 8 <= K; k += 8) {
          Real r[8];
#pragma unroll
          for (int u = 0; u < 8; u++) r[u] = col[(size_t)((k + u) * leaf_n + lbf[k + u]) * dim];
#pragma unroll
          for (int u = 0; u < 8; u++) acc = acc + r[u];
        }
        for (; k < K; k++) acc = acc + col[(size_t)(k * leaf_n + lbf[k]) * dim];
        if (kCpp) {
          // identity STParameter::Apply (data.hpp:42-45) on (dx,dy): 1*(1*x+0*y) / 1*(0*x+1*y)
          const Real other = __shfl_xor(acc, 1);
          const Real zero = (Real)0, one = (Real)1;
          acc = (d & 1) ? one * (zero * other + one * acc) : one * (one * acc + zero * other);
          acc = sh[d] + acc;
        }
        w.shape[pcur][(size_t)slot * dim + d] = acc;
        if (TRACE && is_last) w.tr_shape[(size_t)gid * dim + d] = acc;
      }
      if (lane == 0) {
        if (TRACE && is_last) { w.tr_carts[gid] = m.T * K; w.tr_score[gid] = score; w.tr_hash[gid] = w.q_hash[pcur][slot]; }
        if (emit) {
          const unsigned o = (unsigned)atomicAdd(&w.counters[kCntOut], 1ull);
          if (o < w.cap) out_slot[o] = slot;
        }
      }
    }
  }
}

namespace {
template <typename DL>
hipError_t launch_update_impl(bool trace, int t, bool apply_th, typename DL::Real th, const DevPlan* d_plan,
                              const DevModelT<typename DL::Real>& m, const WorkT<typename DL::Real>& w,
                              hipStream_t stream) {
  using Real = typename DL::Real;
  const int dim_pad = (m.dim + 1) & ~1;
  const size_t lds = 4 * (size_t)dim_pad * sizeof(Real) + 4 * (size_t)((m.K + 7) & ~7) * 2;
  const int multi = (w.half != nullptr) ? 1 : 0;
  const float r = 1.f / sqrtf(2.f);
  unsigned blocks = (w.cap + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks == 0) blocks = 1;
  const int is_last = (t == m.T - 1) ? 1 : 0;
  auto go = [&](auto kern) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, stream, d_plan, m, w, t, multi, r, is_last,
                       apply_th ? 1 : 0, th, w.out_slot);
  };
  if (trace) go(k_update<DL, true>); else go(k_update<DL, false>);
  return hipGetLastError();
}
}  // namespace

template <>
hipError_t launch_update<float>(int dialect, bool trace, int t, bool apply_final_th, float final_th,
                                const DevPlan* d_plan, const DevModelT<float>& m, const WorkT<float>& w,
                                hipStream_t stream) {
  (void)dialect;
  return launch_update_impl<DialectC>(trace, t, apply_final_th, final_th, d_plan, m, w, stream);
}
template <>
hipError_t launch_update<double>(int dialect, bool trace, int t, bool apply_final_th, double final_th,
                                 const DevPlan* d_plan, const DevModelT<double>& m, const WorkT<double>& w,
                                 hipStream_t stream) {
  (void)dialect;
  return launch_update_impl<DialectCPP>(trace, t, apply_final_th, final_th, d_plan, m, w, stream);
}

// =============================================================================
// final detections -> contiguous rows (into the buffers of the other parity)
// =============================================================================

template <typename Real>
__global__ void k_pack(WorkT<Real> w, int T, int dim) {
  const int pl = (T - 1) & 1, po = T & 1;
  const unsigned n = (unsigned)w.counters[kCntOut];
  const unsigned long long total = (unsigned long long)n * dim;
  for (unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned i = (unsigned)(idx / dim);
    const int d = (int)(idx - (unsigned long long)i * dim);
    const unsigned s = w.out_slot[i];  // slot in queue T-1
    w.shape[po][idx] = w.shape[pl][(size_t)s * dim + d];
    if (d == 0) { w.q_gid[po][i] = w.q_gid[pl][s]; w.q_score[po][i] = w.q_score[pl][s]; }
  }
}

template <typename Real>
hipError_t launch_pack(const WorkT<Real>& w, int T, int dim, hipStream_t stream) {
  hipLaunchKernelGGL(k_pack<Real>, dim3(1024), dim3(256), 0, stream, w, T, dim);
  return hipGetLastError();
}
template hipError_t launch_pack<float>(const WorkT<float>&, int, int, hipStream_t);
template hipError_t launch_pack<double>(const WorkT<double>&, int, int, hipStream_t);

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
