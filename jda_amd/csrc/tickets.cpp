// libjda.so, host side: jdaDetectBatchSubmit[Host] / jdaDetectBatchWait -- a batch queued on a lane of its own, collected later.
#include "detect.h"

namespace jda {

// ---- submit / wait: batches in flight on one cascador, driven by one host thread ----
// Submit queues a batch (no host wait) on a lane of its own; Wait collects and post-processes it.  A caller that
// submits batch i+1 before it waits for batch i keeps the GPU busy with batch i+1 while the host parts of batch i run.
int submit_c_device(Cascador* c, const uint8_t* d_frames, size_t stride, int n, int width, int height,
                           float scale, int min_size, int max_size, float th, const jdaDetectOptions* opt,
                           const unsigned char* const* host_frames) {
  if (!c || n <= 0 || (!d_frames && !host_frames)) { fail("bad arguments"); return -1; }
  if (host_frames) stride = (((size_t)width * height) + 255) & ~(size_t)255;      // frames of the staging buffer
  if (c->hm.multi_scale()) { fail("submit/wait supports models whose split nodes read the original image only"); return -1; }
  ScanPlan sp;
  PlanEntry* pe = nullptr;
  if (!plan_c_call(c, stride, width, height, scale, min_size, max_size, &sp, &pe)) return -1;
  PlanPin pin{c, pe};                       // released on the error paths; handed to the ticket on success
  const long long wpf = sp.windows;
  if (wpf <= 0) { fail("no candidate window in these frames"); return -1; }
  if ((long long)n * wpf > 0x7fffffffLL || n > 65535) { fail("batch too large for one submit: split it"); return -1; }
  const size_t cap = (size_t)n * (size_t)wpf;
  LaneSet lanes(c);
  if (!lanes.take(1, cap)) return -1;
  Lane* ln = lanes.v[0];
  bool want_dense = false;
  const QueueCaps qc = plan_queue_caps(c, pe, cap, false, &want_dense);
  if (!ensure_workspace<float>(ln, cap, false, c->hm.dim(), qc.q, qc.m, want_dense)) return -1;
  if (host_frames) {
    // frames still on the host: the ticket's lane stages them on its own stream (the copy of batch i+1 then runs next
    // to the kernels of batch i, which live on the other ticket's stream)
    for (int i = 0; i < n; i++) if (!host_frames[i]) { fail("null frame pointer"); return -1; }
    if (!ln->frames.reserve(stride * (size_t)n)) return -1;
    d_frames = (const uint8_t*)ln->frames.p;
  }
  int slot = -1;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->pending) c->pending = new PendingBatch[kTickets];
    for (int i = 0; i < kTickets; i++) if (!c->pending[i].active && !c->pending[i].reserved) { slot = i; break; }
    if (slot >= 0) c->pending[slot].reserved = true;
  }
  if (slot < 0) { fail("every submit slot is in use: wait for a batch first"); return -1; }
  // The reservation is given back on EVERY way out that has not committed the ticket -- an error return or an exception
  // (std::bad_alloc from the assignments below; the C ABI's barrier turns it into -1): a slot left reserved would be
  // lost to the cascador for good.
  struct Reservation {
    Cascador* c; PendingBatch* pb; bool committed = false;
    ~Reservation() { if (!committed) { std::lock_guard<std::mutex> lk(c->mu); pb->reserved = false; } }
  } reservation{c, &c->pending[slot]};
  PendingBatch& pb = c->pending[slot];
  pb.join_issuer();
  pb.reset();
  pb.sp = sp; pb.pe = pe;
  pb.n = n; pb.opt_set = opt != nullptr; if (opt) pb.opt = *opt;
  pb.rs.timed = opt && opt->stats;        // (a flag here: the statistics themselves are handed to Wait)
  pb.opt.stats = nullptr;
  pb.t_submit = now_ms();
  Pass<float>& p = pb.pass;
  p = Pass<float>();
  p.c = c; p.pe = pb.pe; p.trace = nullptr; p.dets = &pb.dets; p.rs = &pb.rs; p.apply_th = true; p.th = th; p.multi = false;
  p.solo = true;
  if (c->kn.device_post >= 1 && n >= c->kn.device_post_min_frames) {      // (k_post: even with the host work hidden behind the other tickets' kernels, +1 %)
    p.want_post = true; p.post_nms = !opt || opt->nms; p.post_overlap = opt ? opt->nms_overlap : 0.3f;
    pb.dets.p_n.assign((size_t)n, -1); pb.dets.p_first.assign((size_t)n, 0);
  }
  p.bind(ln, 0, nullptr);
  p.f0 = 0; p.nf = n;
  p.w.frames = d_frames; p.w.frame_stride = stride; p.w.n_frames = n;
#ifdef JDA_BOUNDS_CHECK
  p.w.bc_lo = d_frames; p.w.bc_hi = d_frames + (size_t)(n - 1) * stride + (size_t)width * height;
#endif
  p.w.half = nullptr; p.w.quarter = nullptr; p.w.half_stride = p.w.quarter_stride = 0;
  p.w.hw = p.w.hh = p.w.qw = p.w.qh = 0;
  if (host_frames) {
    pb.host_ptrs.assign(host_frames, host_frames + n);       // (the helper thread reads them after Submit has returned)
    p.host_frames = pb.host_ptrs.data(); p.host_fbytes = (size_t)width * height;
  }
  auto give_up = [&]() { return -1; };      // (the Reservation guard releases the slot)
  // opt->hip_stream: the stream the caller produced the frames on -- the scan is ordered behind the work
  // already queued there (the batch itself still runs on the lane's own stream)
  if (opt && opt->hip_stream) {
    if (hipEventRecord(ln->ev_user, (hipStream_t)opt->hip_stream) != hipSuccess ||
        hipStreamWaitEvent(p.st, ln->ev_user, 0) != hipSuccess) { fail("cannot order the batch behind opt->hip_stream"); return give_up(); }
  }
  auto commit = [&]() {
    std::lock_guard<std::mutex> lk(c->mu);
    pb.lane = lanes.detach(0);             // the ticket holds the lane (still busy) and the plan pin until its Wait
    pin.pe = nullptr;
    pb.active = true; pb.reserved = false;
    reservation.committed = true;
  };
  if (host_frames && c->kn.host_submit_thread) {
    // the copy + scan launches of this ticket on their own thread (joined by Wait): a pageable H2D copy blocks
    // its caller for the whole transfer (1.4 ms per 256 frames 640x480), time in which the submitting thread can
    // already collect and post-process the other ticket
    commit();
    pb.issue_ok = true; pb.issue_err.clear();
    PendingBatch* pbp = &pb;
    const int dev = c->device;
    auto issue = [pbp, dev]() {
      // (a thread body: nothing may be thrown out of it -- std::terminate -- so a failed allocation becomes issue_ok = false)
      try {
        if (hipSetDevice(dev) != hipSuccess || !pbp->pass.issue_scan(nullptr, 0, nullptr, 0, nullptr)) {
          pbp->issue_ok = false;
          pbp->issue_err = g_err.empty() ? std::string("issuing the batch failed") : g_err;
        }
      } catch (...) {
        pbp->issue_ok = false;
        try { pbp->issue_err = "issuing the batch failed: C++ exception (out of host memory?)"; } catch (...) {}
      }
    };
    // (no C++ exception may cross the C ABI: when the process cannot start another thread the batch is issued here,
    // as with host_submit_thread = 0 -- the ticket is committed already, Wait reports issue_ok)
    try { pb.issuer = std::thread(issue); }
    catch (const std::system_error&) { issue(); }
    return slot;
  }
  if (!p.issue_scan(nullptr, 0, nullptr, 0, nullptr)) { (void)hipStreamSynchronize(ln->stream); return give_up(); }
  commit();
  return slot;
}

int wait_c_device(Cascador* c, int slot, jdaStats* stats, jdaResult* out) {
  if (!c || slot < 0 || slot >= kTickets || !out) { fail("no pending batch in this slot"); return -1; }
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->pending || !c->pending[slot].active || c->pending[slot].waiting) { fail("no pending batch in this slot"); return -1; }
    c->pending[slot].waiting = true;           // (two threads waiting for one ticket: the second is refused)
    if (!ensure_device(c)) { c->pending[slot].waiting = false; return -1; }   // the waiting thread's current device may differ
  }
  PendingBatch& pb = c->pending[slot];
  // Whatever happens below -- success, an error, an exception out of the host post-processing -- the ticket is closed on
  // the way out: its stream is drained first when the pass did not complete (kernels may still read the caller's frames
  // and write the lane's pinned buffers), then the lane goes back to the pool and the plan is unpinned.  A ticket left
  // `active` and `waiting` would hold its lane and its plan pin for the life of the cascador.
  struct Closer {
    Cascador* c; PendingBatch* pb; bool done = false;
    ~Closer() {
      if (!done && pb->lane) { (void)hipStreamSynchronize(pb->lane->stream); if (pb->lane->side) (void)hipStreamSynchronize(pb->lane->side); (void)hipGetLastError(); }
      std::lock_guard<std::mutex> lk(c->mu);
      if (pb->pe && pb->pe->pins > 0) pb->pe->pins--;
      pb->pe = nullptr;
      if (pb->lane) { pb->lane->busy = false; c->lane_cv.notify_all(); }
      pb->lane = nullptr;
      pb->active = false; pb->waiting = false;
    }
  } closer{c, &pb};
  const int L = c->hm.L, n = pb.n;
  for (int i = 0; i < n; i++) { out[i].n = 0; out[i].landmark_n = L; out[i].bboxes = nullptr; out[i].shapes = nullptr; out[i].scores = nullptr; }
  Pass<float>& p = pb.pass;
  pb.join_issuer();
  bool ok = pb.issue_ok;
  if (!ok) fail(pb.issue_err);
  p.dets = &pb.dets; p.rs = &pb.rs;
  ok = ok && p.after_tail() && p.after_mid() && p.issue_counters() && p.after_counters() && p.collect();
  double post_ms = 0;
  if (ok) {
    float ms_scan = 0, ms_all = 0;
    if (p.timed) {
      (void)hipEventElapsedTime(&ms_scan, p.ev[1], p.ev[2]);
      (void)hipEventElapsedTime(&ms_all, p.ev[0], p.ev[3]);
    }
    pb.rs.scan_ms += ms_scan; pb.rs.gpu_ms += ms_all;
    if (p.lds_span && p.timed) { float ms = 0; if (hipEventElapsedTime(&ms, p.ev[1], p.ev[4]) == hipSuccess) pb.rs.scan_lds_ms += ms; }
    post_ms = post_c(c, pb.sp, pb.dets, n, pb.opt_set ? &pb.opt : nullptr, out);
    fill_stats(stats, pb.rs, pb.sp.windows * n, c->hm.T, c->hm.K, post_ms);
    if (stats) stats->call_ms = now_ms() - pb.t_submit;
    closer.done = true;
  }
  return ok ? 0 : -1;
}

}  // namespace jda
