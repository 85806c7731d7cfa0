// k_stage: dense mode -- one whole stage for a 16x16 tile of windows per workgroup, lane = window,
// tables and weight rows of a chunk of carts shared in LDS (same references as k_finish).
#include "kernels_common.h"

namespace jda {

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
#ifdef JDA_BOUNDS_CHECK
  const Bc bc_fr((long long)(uintptr_t)w.bc_lo, (long long)(uintptr_t)w.bc_hi);
#else
  const Bc bc_fr;
#endif
  if (GLB) { pix = img + (size_t)y0 * W + x0; ppitch = W; }
  else xshift = load_tile<BLOCK>(lds + L.pix, w.frames, w.frame_stride, img, W, x0, y0, pw, ph, pitch, tid, bc_fr, Bc(0, ((long long)pitch * ph + 15) & ~15ll));
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
#ifdef JDA_BOUNDS_CHECK
          if (GLB) { JDA_BC_ADDR(bc_fr, pix + (__umul24((unsigned)y1, (unsigned)ppitch) + (unsigned)x1 + (unsigned)base), 1, kBcStagePix); JDA_BC_ADDR(bc_fr, pix + (__umul24((unsigned)y2, (unsigned)ppitch) + (unsigned)x2 + (unsigned)base), 1, kBcStagePix); }
          else if (valid) { JDA_BC(Bc(0, (long long)pitch * ph), __umul24((unsigned)y1, (unsigned)ppitch) + (unsigned)x1 + (unsigned)base, 1, kBcStagePix); JDA_BC(Bc(0, (long long)pitch * ph), __umul24((unsigned)y2, (unsigned)ppitch) + (unsigned)x2 + (unsigned)base, 1, kBcStagePix); }
#endif
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
      if (o < w.cap_m) {
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
  const int chunk = dense_chunk<Real, Node>(pix_bytes, m.dim, m.node_n, m.leaf_n, m.K, lds_max);
  if (chunk == 0) return hipErrorInvalidValue;
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


JDA_BC_READER(k_stage)

}  // namespace jda
