// Scan-plan construction. Compiled with -ffp-contract=off: the level growth
// is float (dialect C) / double (dialect CPP) arithmetic followed by a
// truncating cast, and must round exactly like the reference build.
#include "plan.h"

#include <algorithm>

namespace jda {

namespace {

void push_level(ScanPlan* p, int win, int step) {
  Level lv;
  lv.win = win;
  lv.step = step;
  lv.nx = (p->width - win) / step + 1;
  lv.ny = (p->height - win) / step + 1;
  lv.base = p->windows;
  p->windows += (long long)lv.nx * lv.ny;
  p->levels.push_back(lv);
}

}  // namespace

bool plan_dialect_c(int width, int height, float scale, int min_size, int max_size,
                    ScanPlan* plan, std::string* err) {
  ScanPlan p;
  p.width = width; p.height = height;
  if (width <= 0 || height <= 0) { if (err) *err = "frame has no pixels"; return false; }
  min_size = std::max(min_size, 24);                       // c/jda.c:459
  if (max_size <= 0) max_size = std::min(width, height);   // c/jda.c:460
  max_size = std::min(max_size, std::min(width, height));  // c/jda.c:321-322
  int win = 24;                                            // c/jda.c:320
  auto grow = [&](int w) { return (int)((float)w * scale); };
  if (!(grow(24) > 24)) {
    // the reference's `win_size *= scale` loops forever here
    if (err) *err = "scale does not grow a 24-pixel window; the reference scan would not terminate";
    return false;
  }
  while (win < min_size) win = grow(win);                  // c/jda.c:331
  for (; win <= max_size; win = grow(win)) {               // c/jda.c:332
    const int step = (int)((float)win * 0.1f);             // c/jda.c:333
    push_level(&p, win, step);
  }
  *plan = std::move(p);
  return true;
}

bool plan_dialect_cpp(int width, int height, int minimum_size, int step, double factor,
                      ScanPlan* plan, std::string* err) {
  ScanPlan p;
  p.width = width; p.height = height;
  if (width <= 0 || height <= 0) { if (err) *err = "frame has no pixels"; return false; }
  if (minimum_size < 1 || step < 1) { if (err) *err = "minimum_size and step must be positive"; return false; }
  if (!((int)(minimum_size * factor) > minimum_size)) {
    if (err) *err = "factor does not grow the window; the reference scan would not terminate";
    return false;
  }
  int win = minimum_size;                                  // cascador.cpp:314
  while (win <= width && win <= height) {                  // cascador.cpp:333
    push_level(&p, win, step);
    win = (int)(win * factor);                             // cascador.cpp:369
  }
  *plan = std::move(p);
  return true;
}

bool plan_single_level(int width, int height, int win, int step, ScanPlan* plan, std::string* err) {
  ScanPlan p;
  p.width = width; p.height = height;
  if (width <= 0 || height <= 0 || win < 1 || step < 1 || win > width || win > height) {
    if (err) *err = "level image smaller than the window, or bad window/step";
    return false;
  }
  push_level(&p, win, step);                               // cascador.cpp:221-226,260
  *plan = std::move(p);
  return true;
}

}  // namespace jda
