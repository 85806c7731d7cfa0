// libjda.so, host side: dialect CPP (the fp64 `src/jda` path), detect method 1 -- the growing window of
// detectMultiScale1 (reference src/jda/cascador.cpp:310-376) followed by Detect's NMS and relocation (431-477) -- on a
// batch of equally sized frames, in host memory or resident on the device.
#include "detect.h"

namespace jda {

jdaResultD empty_result_d(int landmark_n) {
  jdaResultD r;
  r.n = 0; r.landmark_n = landmark_n;
  r.rects = (int*)std::malloc(sizeof(int));
  r.shapes = (double*)std::malloc(sizeof(double));
  r.scores = (double*)std::malloc(sizeof(double));
  return r;
}

void emit_cpp_result(const int* rc, const double* sc, const double* shapes, int n, int L, double overlap, bool nms, jdaResultD* out) {
  const int dim = 2 * L;
  static thread_local std::vector<int> pick;
  if (nms) pick = nms_dialect_cpp(rc, sc, n, overlap);                  // cascador.cpp:444-446
  else { pick.resize((size_t)n); std::iota(pick.begin(), pick.end(), 0); }   // cascador.cpp:447-451
  jdaResultD& r = *out;
  r.n = (int)pick.size(); r.landmark_n = L;
  r.rects = (int*)std::malloc(std::max<size_t>(1, pick.size() * 4) * sizeof(int));
  r.scores = (double*)std::malloc(std::max<size_t>(1, pick.size()) * sizeof(double));
  r.shapes = (double*)std::malloc(std::max<size_t>(1, pick.size() * dim) * sizeof(double));
  if (!r.rects || !r.scores || !r.shapes) {
    std::free(r.rects); std::free(r.scores); std::free(r.shapes);
    r.rects = nullptr; r.scores = nullptr; r.shapes = nullptr; r.n = 0;
    throw std::bad_alloc();
  }
  for (size_t i = 0; i < pick.size(); i++) {
    const int k = pick[i];
    std::memcpy(r.rects + 4 * i, rc + 4 * k, 4 * sizeof(int));
    r.scores[i] = sc[k];
    double* sh = r.shapes + i * dim;
    std::memcpy(sh, shapes + (size_t)k * dim, dim * sizeof(double));
    relocate_dialect_cpp(sh, L, rc[4 * k], rc[4 * k + 1], rc[4 * k + 2], rc[4 * k + 3]);   // cascador.cpp:462-474
  }
}

// NMS + relocation of every frame of a uniform batch from its raw detections (sorted by gid = frame, then scan order)
static double post_cpp(Cascador* c, const ScanPlan& sp, const RawDets<double>& dets, int n, const CppCall& call, jdaResultD* out) {
  const double t0 = now_ms();
  const int L = c->hm.L, dim = c->hm.dim();
  std::vector<size_t> first(n + 1, dets.gid.size());
  {
    size_t i = 0;
    for (int f = 0; f < n; f++) {
      first[f] = i;
      while (i < dets.gid.size() && dets.gid[i] / (uint32_t)sp.windows == (uint32_t)f) i++;
    }
    first[n] = i;
  }
  parallel_for(n, [&](int f) {
    const size_t a = first[f], cnt = first[f + 1] - a;
    static thread_local std::vector<int> rc;
    rc.resize(cnt * 4);
    for (size_t i = 0; i < cnt; i++) {
      const WinRef wr = locate(sp, dets.gid[a + i]);
      rc[4 * i] = wr.x; rc[4 * i + 1] = wr.y; rc[4 * i + 2] = wr.win; rc[4 * i + 3] = wr.win;   // Rect roi_o, cascador.cpp:339
    }
    emit_cpp_result(rc.data(), dets.score.data() + a, dets.shape.data() + a * dim, (int)cnt, L, call.overlap, call.nms != 0, &out[f]);
  }, dets.gid.size() < 6000);
  return now_ms() - t0;
}

int detect_cpp_device(Cascador* c, const uint8_t* d_frames, size_t stride, int n, int width, int height, const CppCall& call,
                      jdaStats* stats, jdaResultD* out, const unsigned char* const* host_frames) {
  const double t_call = now_ms();
  if (!c || !out || n < 0 || (!d_frames && !host_frames && n > 0)) { fail("bad arguments"); return -1; }
  const int L = c->hm.L;
  for (int i = 0; i < n; i++) { out[i].n = 0; out[i].landmark_n = L; out[i].rects = nullptr; out[i].shapes = nullptr; out[i].scores = nullptr; }
  if (!cpp_model_complete(c)) return -1;
  ScanPlan sp; std::string err;
  if (!plan_dialect_cpp(width, height, call.minimum_size, call.step, call.factor, &sp, &err)) { fail(err); return -1; }
  if (!host_frames && stride < (size_t)width * height) { fail("frame_stride smaller than a frame"); return -1; }
  unsigned long long fb; std::memcpy(&fb, &call.factor, 8);
  PlanKey key{width, height, JDA_DIALECT_CPP, call.minimum_size, call.step, c->similarity, fb};
  PlanEntry* pe = nullptr;
  if (!begin_call<double>(c, key, sp, JDA_DIALECT_CPP, &pe)) return -1;
  PlanPin pin{c, pe};
  LaneSet lanes(c);
  HostFrames host;
  if (host_frames) {
    if (!lanes.take(1) || !stage_frames(lanes.v[0], host_frames, n, (size_t)width * height, &stride, true)) return -1;
    d_frames = (const uint8_t*)lanes.v[0]->frames.p;
    host.ptrs = host_frames; host.fbytes = (size_t)width * height;
  }
  RawDets<double> dets;
  RunStats rs;
  rs.timed = stats != nullptr;
  if (!run_device<double>(c, lanes, pe, d_frames, stride, n, false, 0.0, nullptr, &dets, nullptr, &rs, host)) return -1;
  const double post_ms = post_cpp(c, sp, dets, n, call, out);
  fill_stats(stats, rs, sp.windows * n, c->hm.T, c->hm.K, post_ms);
  if (stats) stats->call_ms = now_ms() - t_call;
  return 0;
}

}  // namespace jda
