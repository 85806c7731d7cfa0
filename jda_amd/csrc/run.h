// libjda.so, host side: a call's frames over its lanes (run_device), and the shared opening of an entry (begin_call).
#pragma once
#include "pass.h"

namespace jda {

// test hook: JDA_TEST_WPF_SCALE pretends every frame has that many times more windows (the gid-overflow guard
// is otherwise only reachable with thousands of 4K frames)
inline bool jda_gid_overflow(const Knobs& kn, long long n, long long wpf) {
  const long long scale = std::max<long long>(1, kn.test_wpf_scale);
  return (double)n * (double)wpf * (double)scale > 4294967295.0;
}

// The queue capacities of a pass over `windows` windows on plan pe, from the plan's remembered fractions (read under c->mu);
// *dense: the plan's last pass kept most windows alive -- the pass will run k_stage and needs the per-window state.
inline QueueCaps plan_queue_caps(Cascador* c, PlanEntry* pe, size_t windows, bool trace, bool* dense) {
  std::lock_guard<std::mutex> lk(c->mu);
  *dense = c->kn.dense == 2 || (c->kn.dense != 0 && pe->dense_hint);
  // (k_enqueue puts every window of a level without a scan tile into the hand-off queue: the worst case is the common one)
  const bool full = trace || *dense || !pe->fast_scan || pe->any_untiled;
  return queue_caps(c->kn, windows, pe->pred_tail, pe->pred_mid, pe->pred_out, full);
}

// Frames of a call that are still in host memory: run_device copies them sub-batch by sub-batch into the staging buffer
// of the call's first lane (the copies of one sub-batch then overlap the kernels of the other lane).
struct HostFrames {
  const unsigned char* const* ptrs = nullptr;
  size_t fbytes = 0;
  // dialect C: the passes may post-process their frames on the device (RawDets::p_*), with these NMS settings
  bool device_post = false, nms = true;
  float nms_overlap = 0.3f;
  // dialect CPP, method 0 on a multi-scale model: per-window patches of these sides instead of half / quarter images
  int patch_hs = 0, patch_qs = 0;
};

// Runs the device pipeline over n frames in device memory (d_frames; with host.ptrs set they are copied there first,
// sub-batch by sub-batch).  `lanes` holds the call's first lane; a large batch takes a second one from the pool and
// is split into sub-batches that alternate between the two (streams with their own workspace), see Pass.
template <typename Real>
bool run_device_impl(Cascador* c, LaneSet& lanes_held, PlanEntry* pe, const uint8_t* d_frames, size_t stride, int n,
                            bool apply_th, Real th, hipStream_t user_stream, RawDets<Real>* dets,
                            const TraceOut<Real>* trace, RunStats* rs, HostFrames host);

// A pass that fails half way (an allocation, a launch, a detection list beyond its capacity) leaves work queued on the
// lanes' streams: kernels that still read the caller's frames, copies out of the caller's host memory, writes into
// the lanes' pinned buffers.  The lanes go back to the pool and the caller may free its frames as soon as this
// returns, so everything queued is waited for first.
template <typename Real>
bool run_device(Cascador* c, LaneSet& lanes_held, PlanEntry* pe, const uint8_t* d_frames, size_t stride, int n,
                       bool apply_th, Real th, hipStream_t user_stream, RawDets<Real>* dets,
                       const TraceOut<Real>* trace, RunStats* rs, HostFrames host = HostFrames()) {
  if (run_device_impl<Real>(c, lanes_held, pe, d_frames, stride, n, apply_th, th, user_stream, dets, trace, rs, host)) return true;
  for (Lane* l : lanes_held.v) {
    (void)hipStreamSynchronize(l->stream);
    if (l->side) (void)hipStreamSynchronize(l->side);
  }
  if (user_stream) (void)hipStreamSynchronize(user_stream);
  if (host.ptrs && c->h2d) (void)hipStreamSynchronize(c->h2d);
  (void)hipGetLastError();
  return false;
}

template <typename Real>
bool run_device_impl(Cascador* c, LaneSet& lanes_held, PlanEntry* pe, const uint8_t* d_frames, size_t stride, int n,
                            bool apply_th, Real th, hipStream_t user_stream, RawDets<Real>* dets,
                            const TraceOut<Real>* trace, RunStats* rs, HostFrames host) {
  constexpr int dialect = Sel<Real>::dialect;
  const HostModel& hm = c->hm;
  const int dim = hm.dim();
  const long long wpf = pe->sp.windows;
  const bool want_trace = trace != nullptr;
  const bool multi = hm.multi_scale();
  const unsigned char* const* host_frames = host.ptrs;
  const size_t host_fbytes = host.fbytes;
  if (n == 0) return true;
  if (lanes_held.v.empty() && !lanes_held.take(1)) return false;
  if (wpf == 0) {     // nothing to scan; still honour the staging contract
    Lane* l0 = lanes_held.v[0];
    if (host_frames && !copy_frames_h2d(const_cast<uint8_t*>(d_frames), stride, host_frames, n, host_fbytes, l0->stream)) return false;
    if (host_frames) JDA_HIP(hipStreamSynchronize(l0->stream));
    return true;
  }

  // two lanes when the batch is big enough for each half to fill the machine
  const long long lanes_min = c->kn.lanes_min_windows;
  int lanes = (int)c->kn.lanes;
  if (lanes < 1) lanes = 1;
  if (lanes > 2) lanes = 2;
  if (n < 2 || (long long)n * wpf < lanes_min * 2) lanes = 1;
  // frames still on the host: smaller sub-batches on two lanes, so that the (host-blocking, pageable)
  // copy of one sub-batch overlaps the kernels of the previous one
  const long long host_chunk = c->kn.host_chunk;
  if (host_frames && n >= 2 * host_chunk && c->kn.lanes >= 2) lanes = 2;

  // frames per sub-batch, bounded by the workspace budget (shared by the lanes)
  // (method 0 on a multi-scale model: every window also owns a half_size^2 + quarter_size^2 patch in the lane's pyramid
  // buffer -- 1.9 KB with the shipped 36 / 24 -- which must come out of the same budget, or a batch asks for several
  // times workspace_mb and fails instead of running in more passes)
  const size_t bpw = bytes_per_window<Real>(dim, want_trace) +
                     (multi && host.patch_hs > 0 ? (size_t)host.patch_hs * host.patch_hs + (size_t)host.patch_qs * host.patch_qs : 0);
  const long long budget = (c->kn.workspace_mb << 20) / lanes;
  // (r06: the queues are sized from the plan's remembered fractions, not for every window: far more frames fit the budget.
  // The patches of method 0 stay per window.)
  bool want_dense = false;
  const size_t patch_bpw = bpw - bytes_per_window<Real>(dim, want_trace);
  auto pass_bytes = [&](long long frames) {
    const size_t wn = (size_t)frames * (size_t)wpf;
    const QueueCaps qc = plan_queue_caps(c, pe, wn, want_trace, &want_dense);
    return (long long)(workspace_bytes<Real>(wn, qc.q, qc.m, want_trace, want_dense, dim) + patch_bpw * wn);
  };
  long long fpp = std::max<long long>(1, (n + lanes - 1) / lanes);
  fpp = std::min<long long>(fpp, std::max<long long>(1, 0x7fffffffLL / wpf));
  while (fpp > 1 && pass_bytes(fpp) > budget) fpp = std::max<long long>(1, std::min<long long>(fpp - 1, (long long)((double)fpp * (double)budget / (double)pass_bytes(fpp))));
  fpp = std::min<long long>(fpp, (n + lanes - 1) / lanes);
  if (host_frames && lanes > 1) fpp = std::min<long long>(fpp, std::max<long long>(1, host_chunk));
  fpp = std::min<long long>(fpp, 0x7fffffffLL / wpf);
  fpp = std::min<long long>(fpp, 65535);                       // the queues pack the frame index in 16 bits
  if (fpp < 1) { fail("frame too large for 32-bit window ids"); return false; }
  // detections carry a 32-bit gid over the WHOLE batch (frame * windows-per-frame + scan index): the
  // frame split in the post-processing divides by windows-per-frame, so a wrapped gid would land in
  // the wrong frame silently
  if (jda_gid_overflow(c->kn, n, wpf)) {
    fail("batch too large: frames x windows per frame exceeds 2^32 window ids -- split the batch");
    return false;
  }
  const size_t cap = (size_t)fpp * (size_t)wpf;
  if (!lanes_held.take(lanes, cap)) return false;
  lanes = std::min(lanes, (int)lanes_held.v.size());          // (the pool is at max_lanes: the sub-batches share the lane(s) at hand)
  const QueueCaps qc = plan_queue_caps(c, pe, cap, want_trace, &want_dense);
  for (int l = 0; l < lanes; l++)
    if (!ensure_workspace<Real>(lanes_held.v[l], cap, want_trace, dim, qc.q, qc.m, want_dense)) return false;

  int hw = 0, hh = 0, qw = 0, qh = 0;
  size_t hs = 0, qs = 0;
  if (multi && host.patch_hs > 0) {
    // (one level per plan; the patches of a pass: windows x (hs^2 + qs^2) bytes per frame)
    hw = hh = host.patch_hs; qw = qh = host.patch_qs;
    hs = (((size_t)wpf * hw * hh) + 255) & ~(size_t)255; qs = (((size_t)wpf * qw * qh) + 255) & ~(size_t)255;
    for (int l = 0; l < lanes; l++)
      if (!lanes_held.v[l]->pyr.reserve((hs + qs) * (size_t)fpp + 512)) return false;
  } else if (multi) {
    if (dialect == JDA_DIALECT_C) {
      const float r = 1.f / sqrtf(2.f);                     // c/jda.c:450-456
      hw = (int)((float)pe->sp.width * r); hh = (int)((float)pe->sp.height * r);
    } else {
      hw = (int)(pe->sp.width / std::sqrt(2.)); hh = (int)(pe->sp.height / std::sqrt(2.));   // cascador.cpp:323-324
    }
    qw = pe->sp.width / 2; qh = pe->sp.height / 2;
    if (hw < 1 || hh < 1 || qw < 1 || qh < 1) { fail("frame too small for the half/quarter images"); return false; }
    hs = ((size_t)hw * hh + 255) & ~(size_t)255; qs = ((size_t)qw * qh + 255) & ~(size_t)255;
    for (int l = 0; l < lanes; l++)
      if (!lanes_held.v[l]->pyr.reserve((hs + qs) * (size_t)fpp + 512)) return false;
  }

  // lane 0 runs on the caller's stream when one was given; the other lane is ordered after the
  // work already queued there
  if (user_stream && lanes > 1) {
    JDA_HIP(hipEventRecord(lanes_held.v[0]->ev_user, user_stream));
    for (int l = 1; l < lanes; l++) JDA_HIP(hipStreamWaitEvent(lanes_held.v[l]->stream, lanes_held.v[0]->ev_user, 0));
  }

  if (host.device_post && dets) { dets->p_n.assign((size_t)n, -1); dets->p_first.assign((size_t)n, 0); }
  std::vector<Pass<Real>> ps;
  for (int f0 = 0; f0 < n;) {
    // one round: up to `lanes` sub-batches in flight, their steps interleaved
    ps.clear();
    for (int l = 0; l < lanes && f0 < n; l++) {
      Pass<Real> p;
      p.c = c; p.pe = pe; p.trace = trace; p.dets = dets; p.rs = rs; p.apply_th = apply_th; p.th = th; p.multi = multi;
      p.solo = lanes == 1;
      p.want_post = host.device_post; p.post_nms = host.nms; p.post_overlap = host.nms_overlap;
      p.bind(lanes_held.v[l], l, l == 0 ? user_stream : nullptr);
      p.cap = cap;
      p.f0 = f0; p.nf = std::min<int>((int)fpp, n - f0);
      p.w.frames = d_frames + (size_t)f0 * stride; p.w.frame_stride = stride; p.w.n_frames = p.nf;
#ifdef JDA_BOUNDS_CHECK
      // (bounds-check build: the bytes the caller vouches for -- n frames `stride` apart, the last one width x height)
      // (JDA_BOUNDS_TEST_SHRINK: the checker's own negative control -- with the range cut short it MUST report)
      static const long long bc_shrink = env_ll("JDA_BOUNDS_TEST_SHRINK", 0);
      p.w.bc_lo = d_frames; p.w.bc_hi = d_frames + (size_t)(n - 1) * stride + (size_t)pe->sp.width * pe->sp.height - bc_shrink;
#endif
      if (host_frames) { p.host_frames = host_frames + f0; p.host_fbytes = host_fbytes; }
      p.w.half = nullptr; p.w.quarter = nullptr; p.w.half_stride = p.w.quarter_stride = 0;
      p.w.hw = hw; p.w.hh = hh; p.w.qw = qw; p.w.qh = qh;
      p.w.patch_hs = multi ? host.patch_hs : 0; p.w.patch_qs = multi ? host.patch_qs : 0;
      f0 += p.nf;
      ps.push_back(std::move(p));
    }
    for (auto& p : ps) {
      uint8_t* hbuf = multi ? (uint8_t*)p.ln->pyr.p : nullptr;
      if (!p.issue_scan(hbuf, hs, hbuf ? hbuf + hs * (size_t)fpp : nullptr, qs, nullptr)) return false;
    }
    for (auto& p : ps) if (!p.after_tail()) return false;
    for (auto& p : ps) if (!p.after_mid()) return false;
    for (auto& p : ps) if (!p.issue_counters()) return false;
    // per lane in frame order (dets stay sorted by gid): the first lane's host work overlaps
    // the other lane's last kernels
    for (auto& p : ps) if (!p.after_counters() || !p.collect()) return false;
    // scan time of the round: the lanes' scans run side by side, so their union (first scan
    // start to last scan end) is what one step spends scanning, not the sum of the spans
    if (!ps[0].timed) continue;
    float ms_scan = 0;
    for (auto& p : ps) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ps[0].ev[1], p.ev[2]) == hipSuccess) ms_scan = std::max(ms_scan, ms);
    }
    rs->scan_ms += ms_scan;
    if (ps.size() == 1 && ps[0].lds_span) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ps[0].ev[1], ps[0].ev[4]) == hipSuccess) rs->scan_lds_ms += ms;
    }
    // device time of the round: first lane's start to the last lane's end
    float ms_all = 0;
    for (auto& p : ps) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ps[0].ev[0], p.ev[3]) == hipSuccess) ms_all = std::max(ms_all, ms);
    }
    rs->gpu_ms += ms_all;
    if (c->kn.debug_times) {
      for (auto& p : ps) {
        float a = 0, b = 0, d = 0;
        (void)hipEventElapsedTime(&a, p.ev[0], p.ev[1]); (void)hipEventElapsedTime(&b, p.ev[1], p.ev[2]);
        (void)hipEventElapsedTime(&d, p.ev[2], p.ev[3]);
        fprintf(stderr, "[jda] lane %d frames %d: pre %.3f scan %.3f finish %.3f ms (n_tail %lld)\n", p.lane, p.nf, a, b, d, p.n_tail);
      }
    }
  }
  return true;
}

// (compiled once, in pass.cpp)
#define JDA_RUN_INST(KW, Real)                                                                                                   \
  KW template bool run_device_impl<Real>(Cascador*, LaneSet&, PlanEntry*, const uint8_t*, size_t, int, bool, Real, hipStream_t,   \
                                         RawDets<Real>*, const TraceOut<Real>*, RunStats*, HostFrames);                          \
  KW template bool run_device<Real>(Cascador*, LaneSet&, PlanEntry*, const uint8_t*, size_t, int, bool, Real, hipStream_t,        \
                                    RawDets<Real>*, const TraceOut<Real>*, RunStats*, HostFrames);
#ifndef JDA_PASS_CPP
JDA_RUN_INST(extern, float)
JDA_RUN_INST(extern, double)
#endif

struct PlanPin {           // unpins on scope exit
  Cascador* c; PlanEntry* pe;
  ~PlanPin() { unpin_plan(c, pe); }
};

// The shared part of an entry, under c->mu: device, the model of dialect Real on the device, the plan (pinned).
template <typename Real>
bool begin_call(Cascador* c, const PlanKey& key, const ScanPlan& sp, int dialect, PlanEntry** pe) {
  std::unique_lock<std::mutex> lk(c->mu);
  if (!ensure_device(c) || !upload_model<Real>(c)) return false;
  return get_plan(c, lk, key, sp, dialect, pe);
}
#ifndef JDA_PASS_CPP
extern template bool begin_call<float>(Cascador*, const PlanKey&, const ScanPlan&, int, PlanEntry**);
extern template bool begin_call<double>(Cascador*, const PlanKey&, const ScanPlan&, int, PlanEntry**);
#endif


}  // namespace jda
