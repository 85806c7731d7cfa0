// k_finish: wave = window, for every survivor of the scan: lanes = carts for a stage's tree walks
// (c/jda.c:366-400, cart.cpp:392-404), the score recurrence replayed in cart order (c/jda.c:395-399),
// then lanes = shape coordinates for the regression gather in cart order (c/jda.c:404-411,
// btcart.cpp:407-424), final cut (c/jda.c:414) and emit.
#include "kernels_common.h"

namespace jda {

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
// TILE: the window's own pixels are in LDS (tile, row pitch tpitch), see load_window_tile.
template <typename DL, bool MULTI, bool ST, bool TILE = false>
__device__ __forceinline__ int node_feature(typename DL::Node nd, const typename DL::Real* sh, int win,
                                            const View& v0, const View& v1, const View& v2,
                                            const Stp<typename DL::Real>& stp, bool apply_st,
                                            const uint8_t* tile = nullptr, int tpitch = 0) {
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
    if (TILE) {
      const int a = tile[__umul24((unsigned)y1, (unsigned)tpitch) + (unsigned)x1];
      const int b = tile[__umul24((unsigned)y2, (unsigned)tpitch) + (unsigned)x2];
      return a - b;
    }
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
  v0->img = w.img_off != nullptr ? w.frames + w.img_off[frame] : w.frames + (size_t)frame * w.frame_stride;   // ragged batch: per-image offset
  v0->w = plan->width; v0->h = plan->height; v0->ox = x; v0->oy = y;
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

// The window's win x win pixels -> LDS, by one wave: tile[y * tpitch + x], tpitch = win rounded up to 4.
// A finishing window reads 2 random pixels per split node, 6*K per stage: from the frame each 64-lane byte
// load touches up to 64 cache lines (44 texture-addresser clocks, lds_bench) and drags 128-byte lines through
// L1; from LDS it is one ds_read_u8 (8 clocks at random addresses).  Rows are fetched as aligned dwords and
// shifted into place (the window's first column is at any byte); a dword is only loaded when it holds at
// least one byte of the row, so no load leaves the frame's last page.
__device__ __forceinline__ void load_window_tile(const uint8_t* __restrict__ wbase, int W, int win, uint8_t* tile,
                                                 int tpitch, int lane) {
  const int dpr = tpitch >> 2;
  const float inv = 1.0f / (float)dpr;
  const int total = win * dpr;
  uint32_t* t32 = (uint32_t*)tile;
  for (int idx = lane; idx < total; idx += 64) {
    const int y = (int)(((float)idx + 0.5f) * inv);      // exact: idx < 2^16, see walk_carts_s0
    const int j = idx - y * dpr;
    const uint8_t* p = wbase + (size_t)y * W + 4 * j;
    const unsigned sft = (unsigned)((uintptr_t)p & 3u);
    const uint32_t* q = (const uint32_t*)(p - sft);
    const uint32_t lo = q[0];
    uint32_t hi = 0;
    if ((int)sft + min(4, win - 4 * j) > 4) hi = q[1];
    t32[idx] = __builtin_amdgcn_alignbyte(hi, lo, sft);
  }
}

}  // namespace

// Tree walks of G carts (k[0..G)) of one stage for the window whose shape is sh[],
// in lockstep: per tree level the G node records are fetched together, then the
// 2G pixels, so the memory round trips of the G walks overlap.  -> leaf indices.
template <typename DL, int G, bool MULTI, bool ST, bool TILE = false>
__device__ __forceinline__ void walk_carts(const NodeOff<typename DL::Real>* __restrict__ stage_off,
                                           const uint2* __restrict__ stage_meta, int K, const int* k,
                                           int depth, int node_n, const typename DL::Real* sh, int win,
                                           const View& v0, const View& v1, const View& v2,
                                           const Stp<typename DL::Real>& stp, bool apply_st, int* leaf,
                                           const uint8_t* tile = nullptr, int tpitch = 0) {
  int node[G];
#pragma unroll
  for (int g = 0; g < G; g++) node[g] = 0;
  for (int d = 0; d < depth - 1; d++) {
    // the level's records of the wave's 64 carts are consecutive (kernels.h: lm_index)
    const unsigned first = (1u << d) - 1u, lvl = (unsigned)K * first - first;
    typename DL::Node nd[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
      const unsigned o = lvl + ((unsigned)k[g] << d) + (unsigned)node[g];
      const NodeOff<typename DL::Real> f = stage_off[o];
      const uint2 mt = stage_meta[o];
      nd[g].o1x = f.o1x; nd[g].o1y = f.o1y; nd[g].o2x = f.o2x; nd[g].o2y = f.o2y;
      nd[g].lm1x2 = (int)(mt.x & 0x7fffu); nd[g].lm2x2 = (int)((mt.x >> 15) & 0x7fffu); nd[g].scale = (int)(mt.x >> 30);
      nd[g].th = (int)mt.y;
    }
    int feat[G];
#pragma unroll
    for (int g = 0; g < G; g++) feat[g] = node_feature<DL, MULTI, ST, TILE>(nd[g], sh, win, v0, v1, v2, stp, apply_st, tile, tpitch);
#pragma unroll
    for (int g = 0; g < G; g++) node[g] = 2 * node[g] + (feat[g] <= nd[g].th ? 1 : 2);   // c/jda.c:392-393
  }
#pragma unroll
  for (int g = 0; g < G; g++) leaf[g] = node[g] - node_n;
}

// Stage-0 walks from the level-major table k_prep_stage0 writes for k_finish (S0Node, one 8-byte record per node:
// both pixels as (x, y) inside the window, 11 bits each, and the clamped threshold): one record load instead of two,
// no coordinate arithmetic.  pix/pitch: the window's origin in the frame with the frame's width, or the window's own
// copy in LDS (TILE) with its pitch.
template <int G, bool TILE>
__device__ __forceinline__ void walk_carts_s0(const S0Node* __restrict__ tbl, int K, const int* k, int depth, int node_n,
                                              const uint8_t* __restrict__ pix, int pitch, int* leaf) {
  int node[G];
#pragma unroll
  for (int g = 0; g < G; g++) node[g] = 0;
  for (int d = 0; d < depth - 1; d++) {
    const unsigned first = (1u << d) - 1u, lvl = (unsigned)K * first - first;     // level-major table, lm_index
    S0Node r[G];
#pragma unroll
    for (int g = 0; g < G; g++) r[g] = tbl[lvl + ((unsigned)k[g] << d) + (unsigned)node[g]];
    int pa[G], pb[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
      const unsigned p1 = r[g].lo & 0x3fffffu, p2 = __builtin_amdgcn_alignbit(r[g].hi, r[g].lo, 22) & 0x3fffffu;
      pa[g] = pix[__umul24(p1 >> 11, (unsigned)pitch) + (p1 & 0x7ffu)];
      pb[g] = pix[__umul24(p2 >> 11, (unsigned)pitch) + (p2 & 0x7ffu)];
    }
#pragma unroll
    for (int g = 0; g < G; g++) node[g] = 2 * node[g] + (pa[g] - pb[g] <= (int)((r[g].hi >> 12) & 0x3ffu) - 256 ? 1 : 2);   // c/jda.c:391-393
  }
#pragma unroll
  for (int g = 0; g < G; g++) leaf[g] = node[g] - node_n;
}

// Stages [t_begin, t_end) for every window of the input queue.  Windows that are
// still alive after stage t_end-1 go to the mid queue (t_end < T) or, after the
// final threshold, to the detection list (t_end == T).
// (One-wave workgroups: the hardware keeps at most 16 workgroups on a CU, so a CU works on 16 windows at a
// time.  Workgroups of 2-4 independent waves lift that to the register limit, measured: no gain -- the launches
// are bound by the CU's texture addresser / L1 (TA_BUSY 70-75 % on average, 95 % on the busiest CU with 16
// windows), not by the number of windows in flight.)
template <typename DL, bool TRACE, int kG, bool MULTI, bool ST>
__global__ __launch_bounds__(64) void k_finish(const DevPlan* __restrict__ plan, DevModelT<typename DL::Real> m,
                                               WorkT<typename DL::Real> w, int multi_i, float inv_sqrt2,
                                               int t_begin, int t_end, int apply_th, typename DL::Real final_th,
                                               const S0Node* __restrict__ s0_table, int tile_win) {
  using Real = typename DL::Real;
  constexpr bool kCpp = sizeof(Real) == 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x;
  const int T = m.T, K = m.K, node_n = m.node_n, leaf_n = m.leaf_n, dim = m.dim;
  const int dim_pad = (dim + 1) & ~1;
  Real* sh = (Real*)lds;                                     // current shape        [dim_pad] (the regression updates it
                                                             // in place: coordinate d is read and written by one lane only)
  uint32_t* lbf = (uint32_t*)(sh + dim_pad);                 // W row (in elements) chosen by every cart [K] (kept as the
                                                             // finished offset: 16-bit leaf indices and the multiply in the
                                                             // regression loop cost 37 us per step)
  int* stage_cnt = (int*)(lbf + ((K + 3) & ~3));             // per-block stage counters
  Real* st_tmp = (Real*)(stage_cnt + kMaxStages);            // similarity-transform scratch [2*dim_pad + 8] (ST only)
  // the window's pixels (windows up to tile_win pixels, single-scale models), behind the scratch, 16-byte aligned
  uint8_t* tile = lds + (((size_t)((unsigned char*)(st_tmp + (ST ? 2 * dim_pad + 8 : 0)) - lds) + 15) & ~(size_t)15);
  constexpr bool multi = MULTI;   // split nodes read the half/quarter images too
  (void)multi_i;
  if (lane < kMaxStages) stage_cnt[lane] = 0;
  const bool from_scan = t_begin == 0;
  const unsigned n = (unsigned)min(w.counters[from_scan ? kCntTail : kCntMid], (unsigned long long)w.cap);
  unsigned long long carts_acc = 0;

#ifdef JDA_SCAN_TIMING
  unsigned long long stamps[15];
  int n_stamp = 0;
  int dbg_win = 0;
#define JDA_FSTAMP() do { if (n_stamp < 15) stamps[n_stamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define JDA_FSTAMP() do { } while (0)
#endif
  for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
    JDA_FSTAMP();
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
    const S0Node* s0_tbl = nullptr;
    if (!MULTI && s0_table != nullptr && t_begin == 0) {
      const bool hit = lane < plan->n_levels && plan->lv[lane].win == win;
      const unsigned long long mh = __ballot(hit);
      if (mh) {
        const DevLevel lv = plan->lv[__ffsll((long long)mh) - 1];
        if (lv.tiled) s0_tbl = s0_table + lv.s0_table;
      }
    }
    const uint8_t* wbase = v0.img + (size_t)v0.oy * v0.w + v0.ox;
#ifdef JDA_SCAN_TIMING
    dbg_win = win;
#endif
    const bool use_tile = !MULTI && win <= tile_win;
    const int tpitch = (win + 3) & ~3;
    __syncthreads();                       // previous window's readers are done with sh (and the tile)
    if (!MULTI && use_tile) load_window_tile(wbase, v0.w, win, tile, tpitch, lane);
    {
      const Real* src = from_scan ? m.mean_shape : w.m_shape + (size_t)i * dim;
      for (int d = lane; d < dim; d += 64) sh[d] = src[d];
    }
    __syncthreads();
    JDA_FSTAMP();

    bool alive = true;
    int carts_n = 0;
    for (int t = t_begin; t < t_end; t++) {
      const NodeOff<Real>* n_off = (const NodeOff<Real>*)m.lm_off + (size_t)t * K * node_n;
      const uint2* n_meta = m.lm_meta + (size_t)t * K * node_n;
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
        if (t == 0 && s0_tbl && use_tile) walk_carts_s0<kG, true>(s0_tbl, K, kk, m.D, node_n, tile, tpitch, lf);
        else if (t == 0 && s0_tbl) walk_carts_s0<kG, false>(s0_tbl, K, kk, m.D, node_n, wbase, v0.w, lf);
        else if (!MULTI && use_tile) walk_carts<DL, kG, MULTI, ST, true>(n_off, n_meta, K, kk, m.D, node_n, sh, win, v0, v1, v2, stp, apply_st, lf, tile, tpitch);
        else walk_carts<DL, kG, MULTI, ST>(n_off, n_meta, K, kk, m.D, node_n, sh, win, v0, v1, v2, stp, apply_st, lf);
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
        if (t == t_begin) JDA_FSTAMP();
      }
      if (t != t_begin) JDA_FSTAMP();
      if (!alive) break;
      // leaves of the carts k_scan already scored (needed only now that the stage is passed)
      for (int k0 = 0; k0 < k_first; k0 += 128) {
        int kk[2], lf[2];
        kk[0] = min(k0 + lane, k_first - 1); kk[1] = min(k0 + 64 + lane, k_first - 1);
        if (t == 0 && s0_tbl && use_tile) walk_carts_s0<2, true>(s0_tbl, K, kk, m.D, node_n, tile, tpitch, lf);
        else if (t == 0 && s0_tbl) walk_carts_s0<2, false>(s0_tbl, K, kk, m.D, node_n, wbase, v0.w, lf);
        else if (!MULTI && use_tile) walk_carts<DL, 2, MULTI, ST, true>(n_off, n_meta, K, kk, m.D, node_n, sh, win, v0, v1, v2, stp, apply_st, lf, tile, tpitch);
        else walk_carts<DL, 2, MULTI, ST>(n_off, n_meta, K, kk, m.D, node_n, sh, win, v0, v1, v2, stp, apply_st, lf);
        if (k0 + lane < k_first) lbf[k0 + lane] = (uint32_t)((k0 + lane) * leaf_n + lf[0]) * (uint32_t)dim;
        if (k0 + 64 + lane < k_first) lbf[k0 + 64 + lane] = (uint32_t)((k0 + 64 + lane) * leaf_n + lf[1]) * (uint32_t)dim;
      }
      __syncthreads();
      if (t == t_begin) JDA_FSTAMP();
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
        sh[d] = acc;
      }
      __syncthreads();
      JDA_FSTAMP();
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
#ifdef JDA_SCAN_TIMING
  JDA_FSTAMP();
  if (lane == 0 && w.dbg && blockIdx.x < 65536 && t_begin > 0 && blockIdx.x < n) {
    unsigned long long* o = w.dbg + (size_t)blockIdx.x * 32;
    o[0] = (unsigned long long)n_stamp | (0x7777ull << 32);
    for (int i = 0; i < n_stamp; i++) o[1 + i] = stamps[i];
    o[16] = (unsigned long long)dbg_win;
    o[17] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 32);   // HW_ID, XCC_ID
  }
#endif
  __syncthreads();
  if (lane < T && stage_cnt[lane]) atomicAdd(shard_counter(w.counters, kCntStage0 + lane), (unsigned long long)stage_cnt[lane]);
  if (lane == 0 && carts_acc) atomicAdd(shard_counter(w.counters, kCntCarts), carts_acc);
}

namespace {
template <typename DL>
hipError_t launch_finish_impl(bool trace, int t_begin, int t_end, bool apply_th, typename DL::Real th,
                              const DevPlan* d_plan, const DevModelT<typename DL::Real>& m,
                              const WorkT<typename DL::Real>& w, int groups, long long n_hint, const S0Node* s0_table,
                              int tile_win, hipStream_t stream) {
  using Real = typename DL::Real;
  const int dim_pad = (m.dim + 1) & ~1;
  const bool st = sizeof(Real) == 8 && m.similarity != 0;
  const int multi = (w.half != nullptr) ? 1 : 0;
  if (multi) tile_win = 0;
  const size_t base = (size_t)dim_pad * sizeof(Real) + (size_t)((m.K + 3) & ~3) * 4 + kMaxStages * sizeof(int) +
                      (st ? (2 * (size_t)dim_pad + 8) * sizeof(Real) : 0);
  // tile_win < 0: the largest window tile that keeps 16 workgroups on a CU.  Measured on MI355X with one-wave
  // workgroups (profiles/r02_finish_experiments.txt): up to 7,680 bytes of LDS per workgroup the launch time does
  // not depend on the LDS size, 7,712 bytes cost +17 %, 7,856 +23 % (16 x 7.5 KB = 120 KB) -- the launch is bound
  // by the texture addresser but still needs its 16 windows per CU.
  if (tile_win < 0) {
    tile_win = 0;
    for (int tw = 16; tw <= 255; tw++)
      if (base + (size_t)tw * ((tw + 3) & ~3) + 16 <= kFinishLdsPerGroup) tile_win = tw;
  }
  const size_t tile_bytes = tile_win > 0 ? (size_t)tile_win * ((tile_win + 3) & ~3) + 16 : 0;
  const size_t lds = base + tile_bytes;
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
                       apply_th ? 1 : 0, th, s0_table, tile_win);
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
                                int groups, long long n_hint, const S0Node* s0_table, int tile_win, hipStream_t stream) {
  return launch_finish_impl<DialectC>(trace, t_begin, t_end, apply_final_th, final_th, d_plan, m, w, groups, n_hint, s0_table, tile_win, stream);
}
template <>
hipError_t launch_finish<double>(bool trace, int t_begin, int t_end, bool apply_final_th, double final_th,
                                 const DevPlan* d_plan, const DevModelT<double>& m, const WorkT<double>& w,
                                 int groups, long long n_hint, const S0Node* s0_table, int tile_win, hipStream_t stream) {
  return launch_finish_impl<DialectCPP>(trace, t_begin, t_end, apply_final_th, final_th, d_plan, m, w, groups, n_hint, s0_table, tile_win, stream);
}


}  // namespace jda
