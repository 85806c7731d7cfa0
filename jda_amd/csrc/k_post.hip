// k_post: the per-frame post-processing of a dialect-C batch on the device -- detections of a pass back into scan
// order, the reference's score ordering (exchange sort under a strict `<`, c/jda.c:256-264), greedy NMS
// (c/jda.c:267-284), survivors in scan order (c/jda.c:295-301) and relocation of their landmarks (c/jda.c:303-313) --
// one workgroup per frame, results written straight into mapped pinned host memory.  Same integer and float
// arithmetic in the same order as post.cpp (the host form, which stays the reference for everything this kernel
// declines: a frame with more than kPostMaxDets detections, ties or NaN among more than kPostLiteralMax scores,
// more rows than the host reserved): bit-identical results, tests/test_device_post.py.
#include "kernels_common.h"

namespace jda {

namespace {

constexpr int kPostMaxDets = 1024;      // detections of one frame the kernel takes
constexpr int kPostLiteralMax = 256;    // ... and of those, how many it orders by the literal exchange sort (ties, NaN)

}  // namespace

// rag_gid / rag_img (ragged pass, else null): first gid of every image of the pass (n + 1 entries; an image's gids are
// consecutive: level, y, x) and the images' sizes -- its window grid per level is derived like post_ragged does.
__global__ __launch_bounds__(256) void k_post(const DevPlan* __restrict__ plan, WorkT<float> w, int dim, int do_nms, float overlap,
                                               PostOut o, const uint32_t* __restrict__ rag_gid, const RagImg* __restrict__ rag_img) {
  __shared__ unsigned long long key[kPostMaxDets];      // gid << 32 | position in the pass's detection list
  __shared__ float sc[kPostMaxDets];
  __shared__ int bx[kPostMaxDets], by[kPostMaxDets], bs[kPostMaxDets];
  __shared__ short order[kPostMaxDets];
  __shared__ unsigned char keep[kPostMaxDets];
  __shared__ int s_n, s_over, s_lit, s_kept, s_base;
  const int frame = (int)blockIdx.x, tid = (int)threadIdx.x;
  const unsigned n_out = (unsigned)min(w.counters[kCntOut], (unsigned long long)w.cap_m);
  const unsigned wpf = (unsigned)plan->windows;
  const unsigned g_lo = rag_gid ? rag_gid[frame] : 0u, g_hi = rag_gid ? rag_gid[frame + 1] : 0u;
  if (tid == 0) { s_n = 0; s_over = 0; s_lit = 0; s_kept = 0; s_base = 0; }
  __syncthreads();
  // ---- this frame's detections out of the pass's list (a few thousand entries, read by every workgroup from L2) ----
  for (unsigned i = (unsigned)tid; i < n_out; i += 256u) {
    const unsigned g = w.out_gid[i];
    if (rag_gid ? (g >= g_lo && g < g_hi) : (g / wpf == (unsigned)frame)) {
      const int p = atomicAdd(&s_n, 1);
      if (p < kPostMaxDets) key[p] = ((unsigned long long)g << 32) | (unsigned long long)i;
      else s_over = 1;
    }
  }
  __syncthreads();
  if (s_over) {                           // too many for one workgroup: the host takes this pass
    if (tid == 0) { o.n[frame] = -1; o.flag[0] = 1; }
    return;
  }
  const int n = s_n;
  // ---- back into scan order: bitonic sort of the keys (gids are unique) ----
  int np = 1;
  while (np < n) np <<= 1;
  for (int i = n + tid; i < np; i += 256) key[i] = ~0ull;
  __syncthreads();
  for (int k = 2; k <= np; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = key[i], b = key[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { key[i] = b; key[l] = a; }
        }
      }
      __syncthreads();
    }
  // ---- window and score of every detection (post_host.cpp: locate) ----
  for (int i = tid; i < n; i += 256) {
    const unsigned g = (unsigned)(key[i] >> 32), idx = (unsigned)(key[i] & 0xffffffffu);
    if (rag_gid) {
      // the image's own grids, level after level (ragged.cpp: post_ragged)
      const int W = rag_img[frame].w, H = rag_img[frame].h;
      unsigned lbase = g_lo;
      int l = 0, nx = 1;
      for (;; l++) {
        const DevLevel lv = plan->lv[l];
        nx = (W - lv.win) / lv.step + 1;
        const unsigned cntl = (unsigned)nx * (unsigned)((H - lv.win) / lv.step + 1);
        if (g < lbase + cntl || l + 1 >= plan->n_levels) break;
        lbase += cntl;
      }
      const DevLevel lv = plan->lv[l];
      const unsigned rel = g - lbase;
      bx[i] = (int)(rel % (unsigned)nx) * lv.step; by[i] = (int)(rel / (unsigned)nx) * lv.step; bs[i] = lv.win;
    } else {
      const int wid = (int)(g - (unsigned)frame * wpf);
      int l = 0;
      for (int q = 1; q < plan->n_levels; q++)
        if (wid >= plan->lv[q].base) l = q;
      const DevLevel lv = plan->lv[l];
      const int rel = wid - lv.base;
      by[i] = (rel / lv.nx) * lv.step; bx[i] = (rel % lv.nx) * lv.step; bs[i] = lv.win;
    }
    sc[i] = w.out_score[idx];
    keep[i] = 1;
  }
  __syncthreads();
  if (do_nms && n > 1) {
    // ---- the reference's order: with distinct scores its exchange sort yields THE descending order (rank = scores
    //      above); ties or NaN make its permutation depend on the swaps, replayed literally by one thread ----
    for (int i = tid; i < n; i += 256) {
      const float s = sc[i];
      if (s != s) s_lit = 1;
      int r = 0;
      for (int j = 0; j < n; j++) r += (sc[j] > s || (sc[j] == s && j < i)) ? 1 : 0;
      if (r < n) order[r] = (short)i;
    }
    __syncthreads();
    if (!s_lit)
      for (int i = tid; i + 1 < n; i += 256)
        if (sc[order[i]] == sc[order[i + 1]]) s_lit = 1;
    __syncthreads();
    if (s_lit) {
      if (n > kPostLiteralMax) {
        if (tid == 0) { o.n[frame] = -1; o.flag[0] = 1; }
        return;
      }
      if (tid == 0) {
        for (int i = 0; i < n; i++) order[i] = (short)i;
        for (int i = 0; i + 1 < n; i++)
          for (int j = i + 1; j < n; j++)
            if (sc[order[i]] < sc[order[j]]) { const short t = order[i]; order[i] = order[j]; order[j] = t; }      // c/jda.c:256-264
      }
      __syncthreads();
    }
    // ---- greedy suppression in that order (c/jda.c:267-284) ----
    for (int i = 0; i + 1 < n; i++) {
      const int a = order[i];
      if (keep[a]) {
        const int ax = bx[a], ay = by[a], as = bs[a];
        const int area_a = as * as;
        for (int j = i + 1 + tid; j < n; j += 256) {
          const int b = order[j];
          if (!keep[b]) continue;
          const int qx = bx[b], qy = by[b], qs = bs[b];
          const int ix0 = max(ax, qx), iy0 = max(ay, qy);
          const int ix1 = min(ax + as, qx + qs), iy1 = min(ay + as, qy + qs);
          const int iw = max(0, ix1 - ix0), ih = max(0, iy1 - iy0);
          const float ov = (float)(iw * ih) / (float)(area_a + qs * qs - iw * ih);
          if (ov > overlap) keep[b] = 0;
        }
      }
      __syncthreads();
    }
  }
  // ---- survivors in scan order (c/jda.c:295-301): position = kept detections before it ----
  for (int i = tid; i < n; i += 256) {
    if (!keep[i]) continue;
    int pos = 0;
    for (int j = 0; j < i; j++) pos += keep[j];
    order[pos] = (short)i;                // (the score order is not needed any more)
    atomicAdd(&s_kept, 1);
  }
  __syncthreads();
  const int kept = s_kept;
  if (tid == 0) {
    unsigned base = 0;
    if (kept > 0) base = (unsigned)atomicAdd(o.cursor, (unsigned long long)kept);
    if (base + (unsigned)kept > o.cap_rows) { o.n[frame] = -1; o.flag[0] = 1; s_base = -1; }
    else { o.n[frame] = kept; o.first[frame] = (int)base; s_base = (int)base; }
  }
  __syncthreads();
  const int base = s_base;
  if (base < 0) return;
  for (int p = tid; p < kept; p += 256) {
    const int i = order[p];
    o.bb[3 * (base + p)] = bx[i]; o.bb[3 * (base + p) + 1] = by[i]; o.bb[3 * (base + p) + 2] = bs[i];
    o.score[base + p] = sc[i];
  }
  // landmarks relocated into the frame (post.cpp: relocate_dialect_c; two roundings, no FMA)
  for (int t = tid; t < kept * dim; t += 256) {
    const int p = t / dim, d = t - p * dim;
    const int i = order[p];
    const unsigned idx = (unsigned)(key[i] & 0xffffffffu);
    const float v = w.out_shape[(size_t)idx * dim + d];
    const float fs = (float)bs[i], fo = (float)((d & 1) ? by[i] : bx[i]);
    const float pv = v * fs;
    o.shape[(size_t)(base + p) * dim + d] = pv + fo;
  }
}

hipError_t launch_post(const DevPlan* d_plan, const WorkT<float>& w, int dim, int n_frames, bool do_nms, float overlap,
                       const PostOut& o, hipStream_t stream, const uint32_t* rag_gid, const RagImg* rag_img) {
  if (n_frames <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_post, dim3((unsigned)n_frames), dim3(256), 0, stream, d_plan, w, dim, do_nms ? 1 : 0, overlap, o, rag_gid, rag_img);
  return hipGetLastError();
}

}  // namespace jda
