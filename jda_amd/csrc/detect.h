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

// tickets.cpp
int submit_c_device(Cascador* c, const uint8_t* d_frames, size_t stride, int n, int width, int height,
                    float scale, int min_size, int max_size, float th, const jdaDetectOptions* opt,
                    const unsigned char* const* host_frames = nullptr);
int wait_c_device(Cascador* c, int slot, jdaStats* stats, jdaResult* out);

// ragged.cpp
int detect_ragged(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                  const int* widths, const int* heights, int n, float scale, int min_size, int max_size, float th,
                  const jdaDetectOptions* opt, jdaResult* out);

}  // namespace jda
