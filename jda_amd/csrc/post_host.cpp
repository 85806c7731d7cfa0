// libjda.so, host side: detections of a batch -> per-frame results (sort back into scan order is done by the pass;
// here: window of a gid, the worker pool of the per-frame post-processing, NMS + relocation through post.cpp, the
// jdaResult structs and the statistics block).  Reference: c/jda.c:237-316, 414-440.
#include "detect.h"

namespace jda {

// window of a gid
WinRef locate(const ScanPlan& sp, uint32_t gid) {
  WinRef r;
  r.frame = (int)(gid / (uint32_t)sp.windows);
  const long long wid = gid - (long long)r.frame * sp.windows;
  size_t l = 0;
  for (size_t i = 1; i < sp.levels.size(); i++)
    if (wid >= sp.levels[i].base) l = i;
  const Level& lv = sp.levels[l];
  const long long rel = wid - lv.base;
  r.y = (int)(rel / lv.nx) * lv.step;
  r.x = (int)(rel % lv.nx) * lv.step;
  r.win = lv.win;
  return r;
}

// Host post-processing pool: a few persistent workers for the per-frame NMS + result assembly of
// a batch (0.9 us per frame, 0.22 ms per 256-frame batch when done by the calling thread alone;
// starting threads per call would cost more than that).  One job at a time; a caller that finds
// the pool busy (other cascadors on other threads) does its own work serially.
class PostPool {
 public:
  static PostPool& get() { static PostPool p; return p; }
  // A batch call announces its post-processing job ahead of time (when it starts its GPU work):
  // the workers wake up now and spin until the job arrives or `ms` have passed.
  void prewake(int n, double ms) {
    if (ms <= 0 || n < 64 || !ready_.load(std::memory_order_acquire)) return;     // off by default: measured neutral to slightly negative
    { std::lock_guard<std::mutex> lk(mu_); armed_until_.store(now_ms() + ms); }
    cv_.notify_all();
  }
  // heavy: the items are expensive (many detections per frame), worth spreading even a few of them
  void run(int n, const std::function<void(int)>& fn, bool heavy) {
    // a heavy job (thousands of detections: the per-frame NMS is quadratic) starts the workers if nobody did
    if (heavy && n >= 2 && auto_ && !ready_.load(std::memory_order_acquire)) {
      std::lock_guard<std::mutex> lk(spawn_mu_);
      if (!ready_.load(std::memory_order_acquire)) {
        const unsigned hwc = std::thread::hardware_concurrency();
        const int nw = (int)std::min<unsigned>(6, hwc > 2 ? hwc / 2 : 0);
        spawn(nw);
        if (!workers_.empty()) ready_.store(true, std::memory_order_release); else auto_ = false;
      }
    }
    const bool use = !ready_.load(std::memory_order_acquire) ? false : (heavy ? n >= 2 : n >= 64);
    if (!use || !job_mu_.try_lock()) { for (int i = 0; i < n; i++) fn(i); return; }
    auto job = std::make_shared<Job>();
    job->chunk = heavy ? 1 : 8;
    job->fn = &fn; job->n = n; job->chunks = (n + job->chunk - 1) / job->chunk;
    { std::lock_guard<std::mutex> lk(mu_); job_ = job; gen_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
    work(*job);
    while (job->done.load(std::memory_order_acquire) < job->chunks) std::this_thread::yield();
    { std::lock_guard<std::mutex> lk(mu_); job_.reset(); armed_until_.store(0.0); }
    job_mu_.unlock();
    // an item threw on a worker (an allocation inside fn): it was caught there -- an exception that leaves a std::thread
    // body ends the process -- and is raised again here, on the caller's thread, where the C ABI's barrier turns it
    // into the entry's error value
    if (job->threw.load(std::memory_order_acquire)) throw std::bad_alloc();
  }

 private:
  struct Job {
    const std::function<void(int)>* fn = nullptr;   // valid until every chunk is done (run() waits for that)
    int n = 0, chunks = 0, chunk = 8;
    std::atomic<int> next{0}, done{0};
    std::atomic<bool> threw{false};
  };
  static void work(Job& j) {
    for (int c; (c = j.next.fetch_add(1)) < j.chunks;) {
      const int e = std::min(j.n, (c + 1) * j.chunk);
      try { for (int i = c * j.chunk; i < e; i++) (*j.fn)(i); }
      catch (...) { j.threw.store(true, std::memory_order_release); }
      j.done.fetch_add(1, std::memory_order_release);
    }
  }
  PostPool() {
    // Off by default: typically 0.22 -> 0.08 ms per 256-frame batch with 6 workers, but 1 run in ~50 on the
    // shared GPU boxes had a worker descheduled in mid-chunk (a multi-millisecond stall of the whole call);
    // the serial path is deterministic.  Opt in with JDA_POST_THREADS=6 on a quiet host.
    // JDA_POST_THREADS: -1 (default) = workers only for heavy jobs, started by the first one; 0 = never; n = n workers
    // from the start, for light jobs too (see above)
    const long long want = env_ll("JDA_POST_THREADS", -1);
    auto_ = want < 0;
    const unsigned hwc = std::thread::hardware_concurrency();
    const int nw = (int)std::max<long long>(0, std::min<long long>(want, hwc > 1 ? hwc - 1 : 0));
    spawn(nw);
    ready_.store(!workers_.empty());
  }
  // (a thread that cannot be started is not an error: the job runs on fewer workers, or serially on the caller)
  void spawn(int nw) {
    for (int i = 0; i < nw; i++) {
      try { workers_.emplace_back([this]() { loop(); }); }
      catch (const std::system_error&) { break; }
    }
  }
  ~PostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return stop_ || gen_.load() != seen || now_ms() < armed_until_.load(); });
        if (stop_) return;
      }
      // armed (a batch call is in flight): stay awake until its job arrives -- a sleeping worker
      // can take longer to wake than the whole 0.2 ms job lasts
      while (gen_.load(std::memory_order_acquire) == seen && now_ms() < armed_until_.load(std::memory_order_relaxed))
        std::this_thread::yield();
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (stop_) return;
        if (gen_.load() == seen) continue;       // the arming ran out without a job
        seen = gen_.load();
        job = job_;               // may already be gone (a late wake-up): nothing to do then
      }
      if (job) work(*job);        // a finished job hands out no chunk, so its fn is never called late
    }
  }
  std::mutex mu_, job_mu_, spawn_mu_;
  bool auto_ = false;
  std::atomic<bool> ready_{false};        // workers exist
  std::condition_variable cv_;
  std::shared_ptr<Job> job_;
  std::atomic<unsigned long long> gen_{0};
  std::atomic<double> armed_until_{0.0};
  bool stop_ = false;
  std::vector<std::thread> workers_;
};

void parallel_for(int n, const std::function<void(int)>& fn, bool small_job) {
  PostPool::get().run(n, fn, !small_job);
}

void fill_stats(jdaStats* st, const RunStats& rs, long long patch_n, int T, int K, double host_ms) {
  if (!st) return;
  std::memset(st, 0, sizeof(*st));
  st->patch_n = patch_n;
  st->face_patch_n = rs.out;
  st->nonface_patch_n = patch_n - rs.out;
  st->cart_total_n = rs.carts;
  st->cart_gothrough_n = rs.carts - rs.out * (long long)T * K;   // faces walked all T*K carts
  for (int t = 0; t < T && t < 16; t++) st->stage_done_n[t] = rs.stage_done[t];
  st->average_cart_n = st->nonface_patch_n > 0 ? (double)st->cart_gothrough_n / (double)st->nonface_patch_n : 0.0;
  st->gpu_ms = rs.gpu_ms; st->scan_ms = rs.scan_ms; st->host_ms = host_ms;
  st->scan_cart_n = rs.carts_scan; st->scan_patch_n = rs.win_scan; st->scan_launches = rs.scan_launches;
  st->handoff_n = rs.tail;
  st->dense_passes = rs.dense_passes;
  st->scan_fallbacks = rs.scan_fallbacks;
  st->ws_regrows = rs.ws_regrows;
  st->scan_lds_ms = rs.scan_lds_ms; st->scan_lds_cart_n = rs.carts_scan - rs.carts_scan_glb;
}

jdaResult empty_result(int landmark_n) {
  jdaResult r;
  r.n = 0; r.landmark_n = landmark_n;
  r.bboxes = (int*)std::malloc(sizeof(int));
  r.shapes = (float*)std::malloc(sizeof(float));
  r.scores = (float*)std::malloc(sizeof(float));
  return r;
}

// NMS, relocation and the jdaResult of every frame of a dialect-C batch from its raw detections
// (sorted by gid = frame, then scan order).  Returns the time it took (ms).
double post_c(Cascador* c, const ScanPlan& sp, const RawDets<float>& dets, int n, const jdaDetectOptions* opt,
                     jdaResult* out) {
  const double t0 = now_ms();
  const int L = c->hm.L, dim = c->hm.dim();
  const bool do_nms = !opt || opt->nms;
  const float overlap = opt ? opt->nms_overlap : 0.3f;
  // split by frame (dets are sorted by gid)
  std::vector<size_t> first(n + 1, dets.gid.size());
  {
    size_t i = 0;
    for (int f = 0; f < n; f++) {
      first[f] = i;
      while (i < dets.gid.size() && dets.gid[i] / (uint32_t)sp.windows == (uint32_t)f) i++;
    }
    first[n] = i;
  }
  const bool some_posted = dets.p_n.size() == (size_t)n;
  parallel_for(n, [&](int f) {
    if (some_posted && dets.p_n[f] >= 0) {
      // this frame's pass was post-processed on the device (k_post): kept detections, relocated, in scan order
      const size_t k = (size_t)dets.p_n[f], r0 = (size_t)dets.p_first[f];
      jdaResult& r = out[f];
      r.n = (int)k; r.landmark_n = L;
      r.bboxes = (int*)std::malloc(std::max<size_t>(1, k * 3) * sizeof(int));
      r.scores = (float*)std::malloc(std::max<size_t>(1, k) * sizeof(float));
      r.shapes = (float*)std::malloc(std::max<size_t>(1, k * dim) * sizeof(float));
      if (k) {
        std::memcpy(r.bboxes, &dets.p_bb[r0 * 3], k * 3 * sizeof(int));
        std::memcpy(r.scores, &dets.p_sc[r0], k * sizeof(float));
        std::memcpy(r.shapes, &dets.p_sh[r0 * dim], k * dim * sizeof(float));
      }
      return;
    }
    const size_t a = first[f], cnt = first[f + 1] - a;
    static thread_local std::vector<int> bb, keep;          // per-frame scratch, grown once per thread
    bb.resize(cnt * 3);
    for (size_t i = 0; i < cnt; i++) {
      const WinRef wr = locate(sp, dets.gid[a + i]);
      bb[3 * i] = wr.x; bb[3 * i + 1] = wr.y; bb[3 * i + 2] = wr.win;
    }
    if (do_nms) nms_dialect_c_into(bb.data(), dets.score.data() + a, (int)cnt, overlap, &keep);
    else { keep.resize(cnt); std::iota(keep.begin(), keep.end(), 0); }
    jdaResult& r = out[f];
    r.n = (int)keep.size(); r.landmark_n = L;
    r.bboxes = (int*)std::malloc(std::max<size_t>(1, keep.size() * 3) * sizeof(int));
    r.scores = (float*)std::malloc(std::max<size_t>(1, keep.size()) * sizeof(float));
    r.shapes = (float*)std::malloc(std::max<size_t>(1, keep.size() * dim) * sizeof(float));
    for (size_t i = 0; i < keep.size(); i++) {
      const int k = keep[i];
      std::memcpy(r.bboxes + 3 * i, &bb[3 * k], 3 * sizeof(int));
      r.scores[i] = dets.score[a + k];
      float* sh = r.shapes + i * dim;
      std::memcpy(sh, &dets.shape[(a + k) * dim], dim * sizeof(float));
      relocate_dialect_c(sh, L, bb[3 * k], bb[3 * k + 1], bb[3 * k + 2]);
    }
  }, dets.gid.size() < 6000);
  return now_ms() - t0;
}

}  // namespace jda
