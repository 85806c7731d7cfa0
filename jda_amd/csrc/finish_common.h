// Device helpers of the finishing kernels (k_finish.hip: wave = window; k_wide.hip: workgroup = window): window
// views, split-node features, tree walks of a stage's carts, similarity transform (dialect CPP).
#pragma once
#include "kernels_common.h"

namespace jda {

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
  Bc bc;    // bounds-check build: the device range the image's pixels lie in (empty otherwise)
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
      JDA_BC(Bc(0, (long long)tpitch * win), __umul24((unsigned)y1, (unsigned)tpitch) + (unsigned)x1, 1, kBcFinishTile);
      JDA_BC(Bc(0, (long long)tpitch * win), __umul24((unsigned)y2, (unsigned)tpitch) + (unsigned)x2, 1, kBcFinishTile);
      const int a = tile[__umul24((unsigned)y1, (unsigned)tpitch) + (unsigned)x1];
      const int b = tile[__umul24((unsigned)y2, (unsigned)tpitch) + (unsigned)x2];
      return a - b;
    }
    // rows and widths are below 2^16: 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate)
    JDA_BC_ADDR(v0.bc, v0.img + __umul24((unsigned)(v0.oy + y1), (unsigned)v0.w) + (unsigned)(v0.ox + x1), 1, kBcFinishPix);
    JDA_BC_ADDR(v0.bc, v0.img + __umul24((unsigned)(v0.oy + y2), (unsigned)v0.w) + (unsigned)(v0.ox + x2), 1, kBcFinishPix);
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
#ifdef JDA_BOUNDS_CHECK
  v0->bc = Bc((long long)(uintptr_t)w.bc_lo, (long long)(uintptr_t)w.bc_hi);
#endif
  if (multi && w.patch_hs > 0) {
    // method 0 (cascador.cpp:243-245): the window's own half_size / quarter_size patches, resized from its ROI
    const DevLevel& lv = plan->lv[0];
    const size_t wi = (size_t)(y / lv.step) * (size_t)lv.nx + (size_t)(x / lv.step);
    const int hs = w.patch_hs, qs = w.patch_qs;
    v1->img = w.half + (size_t)frame * w.half_stride + wi * (size_t)(hs * hs); v1->w = hs; v1->h = hs; v1->ox = 0; v1->oy = 0; v1->pw = hs;
    v2->img = w.quarter + (size_t)frame * w.quarter_stride + wi * (size_t)(qs * qs); v2->w = qs; v2->h = qs; v2->ox = 0; v2->oy = 0; v2->pw = qs;
  } else if (multi) {
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

// The window's win x win pixels -> LDS, by one wave (or nthreads threads): tile[y * tpitch + x], tpitch = win rounded up to 4.
// A finishing window reads 2 random pixels per split node, 6*K per stage: from the frame each 64-lane byte
// load touches up to 64 cache lines (44 texture-addresser clocks, lds_bench) and drags 128-byte lines through
// L1; from LDS it is one ds_read_u8 (8 clocks at random addresses).  Rows are fetched as aligned dwords and
// shifted into place (the window's first column is at any byte); a dword is only loaded when it holds at
// least one byte of the row, so no load leaves the frame's last page.
__device__ __forceinline__ void load_window_tile(const uint8_t* __restrict__ wbase, int W, int win, uint8_t* tile,
                                                 int tpitch, int lane, int nthreads = 64, const Bc& bc = Bc()) {
  const int dpr = tpitch >> 2;
  const float inv = 1.0f / (float)dpr;
  const int total = win * dpr;
  uint32_t* t32 = (uint32_t*)tile;
  for (int idx = lane; idx < total; idx += nthreads) {
    const int y = (int)(((float)idx + 0.5f) * inv);      // exact: idx < 2^16, see walk_carts_s0
    const int j = idx - y * dpr;
    const uint8_t* p = wbase + (size_t)y * W + 4 * j;
    const unsigned sft = (unsigned)((uintptr_t)p & 3u);
    const uint32_t* q = (const uint32_t*)(p - sft);
#ifdef JDA_BOUNDS_CHECK
    // (aligned dwords that hold at least one byte of the row: the first may start up to 3 bytes in front of it, the last
    // end up to 3 bytes behind it -- inside the same aligned word as a pixel of the frame, never on another page; what is
    // checked is that every dword read OVERLAPS the frames' range)
    if ((long long)(uintptr_t)q + 4 <= bc.lo || (long long)(uintptr_t)q >= bc.hi) jda_bc_fail(kBcWindowTileLoad, __LINE__);
    if ((int)sft + min(4, win - 4 * j) > 4 && ((long long)(uintptr_t)(q + 1) + 4 <= bc.lo || (long long)(uintptr_t)(q + 1) >= bc.hi)) jda_bc_fail(kBcWindowTileLoad, __LINE__);
    JDA_BC(Bc(0, (long long)tpitch * win), (long long)idx * 4, 4, kBcWindowTileLoad);
#endif
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
                                           const uint8_t* tile = nullptr, int tpitch = 0,
                                           const typename DL::Node* __restrict__ stage_deep = nullptr, int split = 1 << 20) {
  int node[G];
#pragma unroll
  for (int g = 0; g < G; g++) node[g] = 0;
  const int levels = depth - 1;
  const int shallow = min(levels, split);
  for (int d = 0; d < shallow; d++) {
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
  // the last levels of a deep tree (a loop of its own: no iteration for the shipped depth, and the loop above keeps the
  // code it had): whole records, grouped under the path's ancestor (kernels.h: lm_deep_index)
  for (int d = shallow; d < levels; d++) {
    typename DL::Node nd[G];
#pragma unroll
    for (int g = 0; g < G; g++)
      nd[g] = stage_deep[lm_deep_index((unsigned)k[g], (unsigned)d, (unsigned)node[g], (unsigned)levels, (unsigned)split)];
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
                                              const uint8_t* __restrict__ pix, int pitch, int* leaf, const Bc& bc = Bc()) {
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
#ifdef JDA_BOUNDS_CHECK
      // bc: TILE -- indices inside the window's LDS copy; else addresses inside the frames
      if (TILE) { JDA_BC(bc, __umul24(p1 >> 11, (unsigned)pitch) + (p1 & 0x7ffu), 1, kBcFinishTile); JDA_BC(bc, __umul24(p2 >> 11, (unsigned)pitch) + (p2 & 0x7ffu), 1, kBcFinishTile); }
      else { JDA_BC_ADDR(bc, pix + __umul24(p1 >> 11, (unsigned)pitch) + (p1 & 0x7ffu), 1, kBcFinishPix); JDA_BC_ADDR(bc, pix + __umul24(p2 >> 11, (unsigned)pitch) + (p2 & 0x7ffu), 1, kBcFinishPix); }
#endif
      pa[g] = pix[__umul24(p1 >> 11, (unsigned)pitch) + (p1 & 0x7ffu)];
      pb[g] = pix[__umul24(p2 >> 11, (unsigned)pitch) + (p2 & 0x7ffu)];
    }
#pragma unroll
    for (int g = 0; g < G; g++) node[g] = 2 * node[g] + (pa[g] - pb[g] <= (int)((r[g].hi >> 12) & 0x3ffu) - 256 ? 1 : 2);   // c/jda.c:391-393
  }
#pragma unroll
  for (int g = 0; g < G; g++) leaf[g] = node[g] - node_n;
}


}  // namespace jda
