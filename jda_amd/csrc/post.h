// Host post-processing of the survivors of one frame: non-maximum suppression
// and relocation of window-normalised shapes to absolute pixel coordinates.
// Both are order-dependent in the reference and are reproduced exactly.
#pragma once
#include <vector>

namespace jda {

// Dialect C NMS (reference c/jda.c:237-316). Input: n boxes (x,y,size) in scan
// order with scores. Returns the indices that survive, in scan order.
std::vector<int> nms_dialect_c(const int* bboxes3, const float* scores, int n, float overlap);
// The same into a caller-owned vector (no allocation once it has grown): the per-frame loop of a batch.
void nms_dialect_c_into(const int* bboxes3, const float* scores, int n, float overlap, std::vector<int>* keep_out);

// Dialect C relocation (reference c/jda.c:465-474): shape = shape*size + origin,
// a float multiply followed by a float add (never fused).
void relocate_dialect_c(float* shape, int landmark_n, int x, int y, int size);

// Dialect CPP NMS (reference src/jda/cascador.cpp:387-429): multimap on score,
// repeatedly pick the maximum and erase everything overlapping it. Returns the
// picked indices in descending-score order. rects4 is (x,y,w,h).
std::vector<int> nms_dialect_cpp(const int* rects4, const double* scores, int n, double overlap);

// Dialect CPP relocation (reference src/jda/cascador.cpp:462-474).
void relocate_dialect_cpp(double* shape, int landmark_n, int x, int y, int w, int h);

}  // namespace jda
