// libjda.so: the C ABI of include/jda.h on top of the HIP kernels (abi.cpp); this file: the dialect-C batch entry.
//
// Boundary of the device work: the host enumerates pyramid
// levels and does NMS + relocation; the device does resize, cascade walk,
// stage regression and compaction.  There is no CPU fallback for the cascade:
// without a usable HIP device every detect entry fails loudly.
#include "detect.h"

namespace jda {

// Validation + plan of a dialect-C call (shared by the synchronous and the submit/wait entries).  Takes c->mu for
// the shared parts (device, model, plan cache); the plan comes back pinned.
bool plan_c_call(Cascador* c, size_t stride, int width, int height, float scale, int min_size, int max_size,
                        ScanPlan* sp, PlanEntry** pe) {
  std::string err;
  if (!plan_dialect_c(width, height, scale, min_size, max_size, sp, &err)) { fail(err); return false; }
  if (stride < (size_t)width * height) { fail("frame_stride smaller than a frame"); return false; }
  std::unique_lock<std::mutex> lk(c->mu);
  if (!ensure_device(c) || !upload_model<float>(c)) return false;
  unsigned sb; std::memcpy(&sb, &scale, 4);
  PlanKey key{width, height, JDA_DIALECT_C, (int)sb, std::max(min_size, 24), max_size <= 0 ? -1 : max_size, 0ull};
  return get_plan(c, lk, key, *sp, JDA_DIALECT_C, pe);
}

// Dialect CPP walks stages [0, current_stage_idx) and then carts [0, current_cart_idx] of the next one without that
// stage's regression (cascador.cpp:178,199-209): header ints 5, 6 of the model file (cascador.cpp:93-104).  A complete
// model carries (T, -1); the float files of the C library carry (T+1, -1) (c/jda.c:662-665), which the reference's own
// C++ loader would walk out of bounds with -- both mean "every stage" here.  A training snapshot (fewer stages, or a stage
// cut at a cart) runs like Validate runs it: the fp64 tables are padded with pass-through carts and zero weight rows
// (model_dev.cpp; r06 -- until then the dialect-CPP entries refused snapshots); with the similarity transform on, the stage in
// training walks with the PREVIOUS stage's parameter (k_finish keeps it, DevModelT::similarity).  Refused: a status the
// reference's loader asserts against (cascador.cpp:138-141).  Dialect C ignores the header like c/jda.c:499-505 does.
bool cpp_model_complete(const Cascador* c) {
  const HostModel& h = c->hm;
  if ((h.hdr_stage == h.T || h.hdr_stage == h.T + 1) && h.hdr_cart == -1) return true;
  const std::string status = "header says stage " + std::to_string(h.hdr_stage) + ", cart " + std::to_string(h.hdr_cart) + " of T=" + std::to_string(h.T) +
                             ", K=" + std::to_string(h.K);
  if (h.hdr_stage >= 0 && h.hdr_stage < h.T && h.hdr_cart >= -1 && h.hdr_cart < h.K) return true;
  fail("partial model with an impossible training status (" + status + "): refused by the dialect-CPP entries");
  return false;
}

// ... without a plan (entries that only need a lane)
bool begin_device(Cascador* c) {
  std::lock_guard<std::mutex> lk(c->mu);
  return ensure_device(c);
}

// Reserves the lane's staging buffer for n host frames.  defer = false: copies them now and waits; defer = true:
// leaves the copies to run_device (per sub-batch).
bool stage_frames(Lane* ln, const unsigned char* const* frames, int n, size_t fbytes, size_t* stride,
                         bool defer) {
  *stride = (fbytes + 255) & ~(size_t)255;
  for (int i = 0; i < n; i++)
    if (!frames[i]) { fail("null frame pointer"); return false; }
  if (!ln->frames.reserve(*stride * (size_t)std::max(n, 1))) return false;
  if (defer) return true;
  if (!copy_frames_h2d((uint8_t*)ln->frames.p, *stride, frames, n, fbytes, ln->stream)) return false;
  JDA_HIP(hipStreamSynchronize(ln->stream));
  return true;
}

// dialect C batch -> per-frame jdaResult.  Frames on the device (d_frames) or, with host_frames set, in host memory
// (staged through the call's first lane).
int detect_c_device(Cascador* c, const uint8_t* d_frames, size_t stride, int n, int width, int height,
                           float scale, int min_size, int max_size, float th, const jdaDetectOptions* opt,
                           jdaResult* out, const unsigned char* const* host_frames) {
  const double t_call = now_ms();
  if (!c || !out || n < 0) { fail("bad arguments"); return -1; }
  const int L = c->hm.L;
  for (int i = 0; i < n; i++) { out[i].n = 0; out[i].landmark_n = L; out[i].bboxes = nullptr; out[i].shapes = nullptr; out[i].scores = nullptr; }
  if (host_frames) stride = (size_t)width * height;
  ScanPlan sp;
  PlanEntry* pe = nullptr;
  if (!plan_c_call(c, stride, width, height, scale, min_size, max_size, &sp, &pe)) return -1;
  PlanPin pin{c, pe};
  LaneSet lanes(c);
  HostFrames host;
  if (host_frames) {
    if (!lanes.take(1, (size_t)std::max<long long>(1, sp.windows))) return -1;
    if (!stage_frames(lanes.v[0], host_frames, n, (size_t)width * height, &stride, true)) return -1;
    d_frames = (const uint8_t*)lanes.v[0]->frames.p;
    host.ptrs = host_frames; host.fbytes = (size_t)width * height;
  }
  // per-frame sort, NMS and relocation on the device for batches (k_post; the host form stays for single frames, for
  // what the kernel declines, and is what it is tested against)
  host.device_post = c->kn.device_post >= 1 && n >= c->kn.device_post_min_frames;
  host.nms = !opt || opt->nms; host.nms_overlap = opt ? opt->nms_overlap : 0.3f;
  RawDets<float> dets;
  RunStats rs;
  rs.timed = opt && opt->stats;
  if (!run_device<float>(c, lanes, pe, d_frames, stride, n, true, th, opt ? (hipStream_t)opt->hip_stream : nullptr, &dets, nullptr, &rs, host))
    return -1;
  const double post_ms = post_c(c, sp, dets, n, opt, out);
  fill_stats(opt ? opt->stats : nullptr, rs, sp.windows * n, c->hm.T, c->hm.K, post_ms);
  if (opt && opt->stats) opt->stats->call_ms = now_ms() - t_call;
  return 0;
}

}  // namespace jda
