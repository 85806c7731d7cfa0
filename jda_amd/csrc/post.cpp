// NMS + relocation on the host. Compiled with -ffp-contract=off.
#include "post.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace jda {

std::vector<int> nms_dialect_c(const int* bb, const float* scores, int n, float overlap) {
  std::vector<int> out;
  nms_dialect_c_into(bb, scores, n, overlap, &out);
  return out;
}

void nms_dialect_c_into(const int* bb, const float* scores, int n, float overlap, std::vector<int>* keep_out) {
  // scratch reused across calls on this thread (the per-frame lists are short and many)
  static thread_local std::vector<int> order;
  static thread_local std::vector<char> keep;
  order.resize(n);
  std::iota(order.begin(), order.end(), 0);
  // The reference orders candidates with an exchange sort under a strict `<`
  // (c/jda.c:256-264). When all scores are distinct its result is THE
  // descending order, which any stable sort finds. With ties (or NaN) the
  // exchange sort's permutation is replayed literally.
  bool literal = false;
  for (int i = 0; i < n && !literal; i++) literal = std::isnan(scores[i]);
  if (!literal) {
    if (n <= 64) {                      // insertion sort: stable, no allocation
      for (int i = 1; i < n; i++) {
        const int v = order[i];
        int j = i - 1;
        while (j >= 0 && scores[order[j]] < scores[v]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = v;
      }
    } else {
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] > scores[b]; });
    }
    for (int i = 0; i + 1 < n && !literal; i++) literal = scores[order[i]] == scores[order[i + 1]];
  }
  if (literal) {
    std::iota(order.begin(), order.end(), 0);
    for (int i = 0; i + 1 < n; i++)
      for (int j = i + 1; j < n; j++)
        if (scores[order[i]] < scores[order[j]]) std::swap(order[i], order[j]);
  }

  keep.assign(n, 1);
  for (int i = 0; i + 1 < n; i++) {                       // c/jda.c:267-284
    const int a = order[i];
    if (!keep[a]) continue;
    const int ax = bb[3 * a], ay = bb[3 * a + 1], as = bb[3 * a + 2];
    const int area_a = as * as;
    for (int j = i + 1; j < n; j++) {
      const int b = order[j];
      if (!keep[b]) continue;
      const int bx = bb[3 * b], by = bb[3 * b + 1], bs = bb[3 * b + 2];
      const int ix0 = std::max(ax, bx), iy0 = std::max(ay, by);
      const int ix1 = std::min(ax + as, bx + bs), iy1 = std::min(ay + as, by + bs);
      const int iw = std::max(0, ix1 - ix0), ih = std::max(0, iy1 - iy0);
      const float ov = (float)(iw * ih) / (float)(area_a + bs * bs - iw * ih);
      if (ov > overlap) keep[b] = 0;
    }
  }
  std::vector<int>& out = *keep_out;
  out.clear();
  for (int i = 0; i < n; i++)                             // c/jda.c:295-301: scan order
    if (keep[i]) out.push_back(i);
}

void relocate_dialect_c(float* shape, int landmark_n, int x, int y, int size) {
  const float fs = (float)size, fx = (float)x, fy = (float)y;
  for (int j = 0; j < landmark_n; j++) {
    const float px = shape[2 * j] * fs;
    const float py = shape[2 * j + 1] * fs;
    shape[2 * j] = px + fx;
    shape[2 * j + 1] = py + fy;
  }
}

std::vector<int> nms_dialect_cpp(const int* r, const double* scores, int n, double overlap) {
  // std::multimap<double,int> iterates by ascending key, equal keys in
  // insertion order (cascador.cpp:394-397): that is a stable ascending sort.
  std::vector<int> asc(n);
  std::iota(asc.begin(), asc.end(), 0);
  std::stable_sort(asc.begin(), asc.end(), [&](int a, int b) { return scores[a] < scores[b]; });
  std::vector<char> alive(n, 1);
  std::vector<int> picked;
  int hi = n - 1;  // position of the greatest live key
  while (true) {
    while (hi >= 0 && !alive[hi]) hi--;
    if (hi < 0) break;
    const int last = asc[hi];                              // map.rbegin()
    picked.push_back(last);
    const double la = (double)(r[4 * last + 2] * r[4 * last + 3]);
    bool erased_last = false;
    for (int p = 0; p <= hi; p++) {                        // cascador.cpp:405-422
      if (!alive[p]) continue;
      const int idx = asc[p];
      const double x1 = std::max(r[4 * idx], r[4 * last]);
      const double y1 = std::max(r[4 * idx + 1], r[4 * last + 1]);
      const double x2 = std::min(r[4 * idx] + r[4 * idx + 2], r[4 * last] + r[4 * last + 2]);
      const double y2 = std::min(r[4 * idx + 1] + r[4 * idx + 3], r[4 * last + 1] + r[4 * last + 3]);
      const double w = std::max(0., x2 - x1), h = std::max(0., y2 - y1);
      const double ia = (double)(r[4 * idx + 2] * r[4 * idx + 3]);
      const double wh = w * h;
      const double ov = wh / (ia + la - wh);
      if (ov > overlap) { alive[p] = 0; if (p == hi) erased_last = true; }
    }
    if (!erased_last) {
      // A box that does not overlap itself above the threshold (overlap >= 1
      // or an empty rect) is never erased by the reference, which then spins
      // forever; drop it so the loop ends and say nothing more about it.
      alive[hi] = 0;
    }
  }
  return picked;
}

void relocate_dialect_cpp(double* shape, int landmark_n, int x, int y, int w, int h) {
  for (int j = 0; j < landmark_n; j++) {
    const double px = shape[2 * j] * (double)w;
    const double py = shape[2 * j + 1] * (double)h;
    shape[2 * j] = (double)x + px;
    shape[2 * j + 1] = (double)y + py;
  }
}

}  // namespace jda
