// Window enumeration ("scan plan") of one frame size: which window sizes are
// scanned, with which step, and where each level's windows start in the
// per-frame scan order (level, y, x).
#pragma once
#include <string>
#include <vector>

namespace jda {

struct Level {
  int win;        // window side in pixels
  int step;       // x/y stride in pixels
  int nx, ny;     // windows per row / column
  long long base; // index of this level's first window within the frame
};

struct ScanPlan {
  int width = 0, height = 0;
  std::vector<Level> levels;
  long long windows = 0;  // per frame
};

// Dialect C: reference c/jda.c:459-460 (argument fix-ups) and 318-339
// (24-pixel seed grown by float multiplication, step = 10 % of the window).
// Fails (false + err) where the reference would never terminate (a scale
// that does not grow the window) or the frame is degenerate.
bool plan_dialect_c(int width, int height, float scale, int min_size, int max_size,
                    ScanPlan* plan, std::string* err);

// Dialect CPP: reference src/jda/cascador.cpp:310-376 (fddb.method = 1):
// start at minimum_size, fixed pixel step, win = int(win*factor) in double.
bool plan_dialect_cpp(int width, int height, int minimum_size, int step, double factor,
                      ScanPlan* plan, std::string* err);

// One pyramid level of dialect CPP's method 0 (reference src/jda/cascador.cpp:216-262,
// detectSingleScale): a fixed win x win window slid with a pixel step over a level image.
bool plan_single_level(int width, int height, int win, int step, ScanPlan* plan, std::string* err);

}  // namespace jda
