// libjda.so, host side: what the entry points (abi.cpp) and the translation units behind them share.
#pragma once
#include "run.h"

namespace jda {

struct WinRef { int frame, x, y, win; };

// post_host.cpp
WinRef locate(const ScanPlan& sp, uint32_t gid);
void parallel_for(int n, const std::function<void(int)>& fn, bool small_job = false);
void fill_stats(jdaStats* st, const RunStats& rs, long long patch_n, int T, int K, double host_ms);
jdaResult empty_result(int landmark_n);
double post_c(Cascador* c, const ScanPlan& sp, const RawDets<float>& dets, int n, const jdaDetectOptions* opt,
              jdaResult* out);

// detect.cpp
bool plan_c_call(Cascador* c, size_t stride, int width, int height, float scale, int min_size, int max_size,
                 ScanPlan* sp, PlanEntry** pe);
bool cpp_model_complete(const Cascador* c);
bool begin_device(Cascador* c);
bool stage_frames(Lane* ln, const unsigned char* const* frames, int n, size_t fbytes, size_t* stride, bool defer = false);
int detect_c_device(Cascador* c, const uint8_t* d_frames, size_t stride, int n, int width, int height,
                    float scale, int min_size, int max_size, float th, const jdaDetectOptions* opt,
                    jdaResult* out, const unsigned char* const* host_frames = nullptr);

// detect_cpp.cpp: dialect CPP, method 1 (cascador.cpp:310-376,431-477) on a uniform batch; frames on the device
// (d_frames) or, with host_frames set, in host memory
struct CppCall { int minimum_size, step; double factor, overlap; int nms; };
int detect_cpp_device(Cascador* c, const uint8_t* d_frames, size_t stride, int n, int width, int height, const CppCall& call,
                      jdaStats* stats, jdaResultD* out, const unsigned char* const* host_frames = nullptr);
// NMS (cascador.cpp:387-429), relocation (462-474) and the jdaResultD of one image from its n candidates in scan order:
// rects (x, y, w, h), scores, window-normalised shapes of `dim` doubles each
void emit_cpp_result(const int* rects4, const double* scores, const double* shapes, int n, int L, double overlap, bool nms,
                     jdaResultD* out);
jdaResultD empty_result_d(int landmark_n);

// tickets.cpp
int submit_c_device(Cascador* c, const uint8_t* d_frames, size_t stride, int n, int width, int height,
                    float scale, int min_size, int max_size, float th, const jdaDetectOptions* opt,
                    const unsigned char* const* host_frames = nullptr);
int wait_c_device(Cascador* c, int slot, jdaStats* stats, jdaResult* out);

// ragged.cpp: a list of differently sized images as one job, either dialect
int detect_ragged(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                  const int* widths, const int* heights, int n, float scale, int min_size, int max_size, float th,
                  const jdaDetectOptions* opt, jdaResult* out);
// Detection rows of a job, grown with realloc and handed to the caller as they are (jdaRowsRelease / jdaRowsDRelease = free).
template <typename T>
struct RowsOut {
  T* p = nullptr; size_t n = 0, cap = 0;          // n, cap in elements
  T* grow(size_t add) {                            // room for `add` more elements; returns where they start
    if (n + add > cap) {
      size_t nc = std::max<size_t>(std::max<size_t>(cap * 2, n + add), 1024);
      T* q = (T*)std::realloc(p, nc * sizeof(T));
      if (!q) throw std::bad_alloc();
      p = q; cap = nc;
    }
    T* at = p + n; n += add;
    return at;
  }
  T* release() { T* q = p ? p : (T*)std::malloc(sizeof(T)); p = nullptr; n = cap = 0; return q; }   // (never NULL on success)
  ~RowsOut() { std::free(p); }
  RowsOut() = default; RowsOut(const RowsOut&) = delete; RowsOut& operator=(const RowsOut&) = delete;
};
int detect_ragged_rows(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                       const int* widths, const int* heights, int n, float scale, int min_size, int max_size, float th,
                       const jdaDetectOptions* opt, int frame_offset, RowsOut<float>* rows);
int detect_ragged_cpp_rows(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                           const int* widths, const int* heights, int n, const CppCall& call, jdaStats* stats, int frame_offset,
                           RowsOut<double>* rows);
int detect_ragged_cpp(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                      const int* widths, const int* heights, int n, const CppCall& call, jdaStats* stats, jdaResultD* out);

}  // namespace jda
