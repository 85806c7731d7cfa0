// libjda.so, host side: one sub-batch of frames through the device pipeline on one lane (Pass), and a submitted batch.
#pragma once
#include "host.h"

namespace jda {

// ---- the persistent scan's configuration, the parts that do not depend on the pass (k_scan_p.hip) ----
// Cart ranges (buckets), task forms, rings, workgroup size of a cascador's persistent scan launches; returns the carts
// it evaluates (the hand-off).
inline int scan_p_base_cfg(const Cascador* c, PScanCfg* cfg, int* block, int* wgs) {
  const Knobs& kn = c->kn;
  const int K = std::min(c->hm.K, (int)(kn.scan_p_handoff > 0 ? kn.scan_p_handoff : kn.handoff));
  const long long bs[5] = {kn.scan_p_b0, kn.scan_p_b1, kn.scan_p_b2, kn.scan_p_b3, kn.scan_p_b4};
  int digits[kPScanMaxBuckets] = {6, 6, 6, 6, 6, 6}, nd = 0;
  { long long v = std::max<long long>(0, kn.scan_p_lg); int tmp[16]; int n = 0; while (v > 0 && n < 16) { tmp[n++] = (int)(v % 10); v /= 10; }
    for (int i = n - 1; i >= 0 && nd < kPScanMaxBuckets; i--) digits[nd++] = tmp[i]; }
  int last = 0;
  for (int i = 0; i < 5 && cfg->nb < kPScanMaxBuckets; i++) {
    const int b = (int)bs[i];
    if (b <= last || b >= K) continue;
    cfg->bound[cfg->nb] = b;
    const int d = digits[cfg->nb];
    cfg->lg[cfg->nb] = ((d == 4 || d == 5 || d == 7 || d == 8 || d == 9) && c->hm.leaf_n() <= 256) || d == 2 || d == 3 ? d : 6;
    cfg->nb++;
    last = b;
  }
  cfg->bound[cfg->nb] = K;
  cfg->bound_last = K;
  cfg->any_norm = stage0_any_norm(c->hm, K, true) ? 1 : 0;
  *block = (int)std::max<long long>(64, std::min<long long>(1024, kn.scan_p_block)) & ~63;
  *wgs = (int)std::max<long long>(1, std::min<long long>(8, kn.scan_p_wgs));
  cfg->ring_cap[0] = (int)std::max<long long>(64, std::min<long long>(4096, kn.scan_p_ring));
  scan_p_ring_caps(cfg, *block / 64);
  return K;
}
// Pixel-tile slots a persistent workgroup gets for tiles of cfg->slot_bytes, or 0 when the level is left to k_scan's closed
// tiles: fewer than scan_p_min_slots slots (few resident windows per wave) or too few tiles to keep the workgroups fed.
// capped: other kernels are in flight next to the pass (scan_p_slots).  The answer for "will it take the level at all" does
// not depend on `capped` (scan_p_slots >= scan_p_min_slots) -- ragged_build_chunk asks it ahead of the pass.
inline long long scan_p_slots_for(const Cascador* c, const PScanCfg& cfg_in, int K, int block, int wgs, long long n_tiles, bool capped) {
  const Knobs& kn = c->kn;
  PScanCfg cfg = cfg_in;
  cfg.slots = 0;
  const long long fixed = (long long)scan_p_lds_bytes(cfg, K, c->hm.node_n(), c->hm.leaf_n(), block / 64);
  const long long budget = std::max<long long>(16, std::min<long long>(160, kn.scan_p_lds_kb)) * 1024 / wgs;
  long long slots = (budget - fixed) / std::max(1, cfg.slot_bytes);
  if (kn.scan_p_slots > 0 && capped) slots = std::min<long long>(slots, kn.scan_p_slots);
  slots = std::min<long long>(slots, 8);
  if (slots < 2 || fixed + slots * cfg.slot_bytes >= (1 << 18)) return 0;
  if (kn.scan_p == 1 && slots < kn.scan_p_min_slots) return 0;     // few resident windows per wave: k_scan's closed tiles do better there
  if (kn.scan_p == 1 && n_tiles < (long long)c->n_cus * wgs * 4) return 0;   // too few tiles to keep persistent workgroups fed
  return slots;
}
// Will the persistent scan take a single-level launch of a ragged chunk (tiles of at most pix_bytes, n_tiles of them)?
inline bool scan_p_takes_ragged(const Cascador* c, int pix_bytes, long long n_tiles) {
  if (!c->kn.scan_p || !c->kn.scan_p_ragged) return false;
  PScanCfg cfg{};
  int block = 0, wgs = 1;
  const int K = scan_p_base_cfg(c, &cfg, &block, &wgs);
  cfg.slot_bytes = (pix_bytes + 15) & ~15;
  return scan_p_slots_for(c, cfg, K, block, wgs, n_tiles, false) > 0;
}

// One sub-batch of frames going through the device pipeline on one lane (stream + workspace).
// The pipeline has four host-visible waits (hand-off count, mid-queue count, counters, results);
// the methods are the pieces between them, so that run_device can interleave two lanes: while
// one lane's latency-bound finishing kernels and host work run, the other lane's scan keeps
// the machine busy.
template <typename Real>
struct Pass {
  Cascador* c; PlanEntry* pe; const TraceOut<Real>* trace; RawDets<Real>* dets; RunStats* rs;
  bool apply_th; Real th; bool multi = false;   // multi: hm().multi_scale(), a scan of the model: computed once
  Lane* ln = nullptr; int lane = 0; bool solo = true;   // lane: index inside the call; solo: the only lane of this call
  hipStream_t st = nullptr; hipEvent_t* ev = nullptr; unsigned long long* h_cnt = nullptr;
  // the plan's hints as they stood when the pass was set up (the plan is shared with concurrent callers: read and
  // written under c->mu only, see bind())
  bool hint_dense = false; double pred_tail = -1, pred_out = -1, pred_mid = -1;
  int busy_lanes = 1;               // lanes of the cascador in use when the pass was set up (concurrent callers)
  void bind(Lane* l, int index, hipStream_t stream) {
    ln = l; lane = index; st = stream ? stream : l->stream; ev = l->ev; h_cnt = l->h_cnt;
    timed = !rs || rs->timed || c->kn.debug_times;
    w = Sel<Real>::work(l); cap = l->cap; cap_q = w.cap_q; cap_m = w.cap_m;
    std::lock_guard<std::mutex> lk(c->mu);
    hint_dense = pe->dense_hint; pred_tail = pe->pred_tail; pred_out = pe->pred_out; pred_mid = pe->pred_mid;
    busy_lanes = 0;
    for (auto& up : c->lanes) busy_lanes += up->busy ? 1 : 0;
  }
  WorkT<Real> w; size_t cap = 0;
  size_t cap_q = 0, cap_m = 0;     // entries of the hand-off queue / of the mid queue and the detection list (WorkT::cap_q, cap_m)
  bool counted = false;            // after_counters has folded and tallied this pass's counters
  int overflow_runs = 0;           // times this pass has been issued again because a queue was too small (recover_overflow)
  // The lane's workspace has been carved again (grown): its array pointers replace the pass's, what the pass itself set
  // (frames, pyramid images, ragged tables) stays.
  void adopt_workspace() {
    WorkT<Real> nw = Sel<Real>::work(ln);
    nw.frames = w.frames; nw.frame_stride = w.frame_stride; nw.n_frames = w.n_frames;
#ifdef JDA_BOUNDS_CHECK
    nw.bc_lo = w.bc_lo; nw.bc_hi = w.bc_hi;
#endif
    nw.half = w.half; nw.half_stride = w.half_stride; nw.hw = w.hw; nw.hh = w.hh;
    nw.quarter = w.quarter; nw.quarter_stride = w.quarter_stride; nw.qw = w.qw; nw.qh = w.qh;
    nw.patch_hs = w.patch_hs; nw.patch_qs = w.patch_qs;
    nw.segs = w.segs; nw.blk = w.blk; nw.img_off = w.img_off;
    w = nw; cap = ln->cap; cap_q = w.cap_q; cap_m = w.cap_m;
  }
  // Dense mode keeps per-window state in the mid-queue arrays (k_stage): every window needs an entry.
  bool grow_for_dense() {
    if (ln->dense_ws && cap_m >= (size_t)windows()) return true;
    JDA_HIP(hipStreamSynchronize(st));
    if (!ensure_workspace<Real>(ln, std::max(cap, (size_t)windows()), want_trace(), hm().dim(), cap_q, 0, true)) return false;
    adopt_workspace();
    return true;
  }
  int f0 = 0, nf = 0;
  const unsigned char* const* host_frames = nullptr; size_t host_fbytes = 0;   // frames of this sub-batch still on the host
  const RaggedChunk* rag = nullptr;   // ragged pass: images of different sizes (w.segs / w.blk / w.img_off set by stage_ragged)
  // state between the steps
  bool dense = false, finished = false, lds_span = false;
  bool timed = true;               // RunStats::timed
  bool predicted = false;          // the finishing launches were sized from PlanEntry::pred_tail, no host wait in between
  bool counters_issued = false, results_pending = false;
  int p_launches = 0;              // k_scan_p launches of this pass so far (each deals its tiles from its own counter words)
  bool want_post = false, post_nms = true; float post_overlap = 0.3f;   // the caller takes device-post-processed frames (RawDets::p_*)
  bool post_issued = false, posted = false; size_t post_cap = 0;         // k_post queued for this pass / its results are good
  bool mid_direct = false;         // a scan launch of this pass put stage-0 survivors into the mid queue itself (k_scan_p up to cart K)
  bool no_scan_p = false;          // this pass has been started over without k_scan_p (recover_scan): its watchdog tripped or it covered too few windows
  int my_scan_launches = 0;        // scan launches of this pass counted into rs so far
  uint8_t* a_hbuf = nullptr; size_t a_hs = 0; uint8_t* a_qbuf = nullptr; size_t a_qs = 0;   // issue_scan's arguments (recover_scan issues it again)
  long long n_tail = -1;
  size_t n_out = 0, out_copied = 0;   // detections of the pass / how many of them are already on their way to the host

  const HostModel& hm() const { return c->hm; }
  const DevModelT<Real>& model() const { return Sel<Real>::model(c).m; }
  bool want_trace() const { return trace != nullptr; }
  // 64-cart groups walked per round by a window that is expected to pass whole stages: the count in
  // 2..4 that wastes the fewest speculative walks past cart K-1 (K = 540: 3 groups, 576 walks, not 768)
  int stage_groups() const {
    const int K = hm().K;
    int best = 4, best_waste = 1 << 30;
    for (int g = 4; g >= 2; g--) {
      const int per = 64 * g, waste = ((K + per - 1) / per) * per - K;
      if (waste < best_waste) { best = g; best_waste = waste; }
    }
    return best;
  }
  // resolved stage-0 tables for k_finish (A/B switch: JDA_FIN_S0=0)
  // k_finish reads the level-major copy of the stage-0 tables (second half of the allocation)
  const S0Node* s0_tbl() const { return (pe->fast_scan && pe->table && pe->lm_ok && c->kn.fin_s0) ? pe->table + pe->table_cap : nullptr; }
  const Knobs& kn() const { return c->kn; }
  // k_filter0 + k_finish(survivors) can take this pass's hand-off queue (every level has a resolved stage-0 table)
  bool filter0_ok() const { return kn().filter0 && s0_tbl() != nullptr && pe->fast_scan && !pe->any_untiled && !multi; }
  long long windows() const { return rag ? rag->windows : (long long)nf * pe->sp.windows; }

  bool dense_ok(int* pix_cap, int* lds_max) const {
    constexpr int dialect = Sel<Real>::dialect;
    const long long dense_env = kn().dense;                       // 0 off, 1 auto, 2 always
    *lds_max = (int)kn().dense_lds_max;
    const int dim = hm().dim();
    const int fixed = (int)stage_lds_bytes(dim, hm().node_n(), hm().leaf_n(), (int)sizeof(Real));
    *pix_cap = std::max(0, std::min<int>((int)kn().dense_pix, *lds_max - fixed));
    return dense_env != 0 && !multi && !(dialect == JDA_DIALECT_CPP && c->similarity) &&
           dim <= 160 && hm().leaf_n() <= 256 && fixed <= *lds_max;
  }
  bool run_dense() {
    int pix_cap, lds_max;
    (void)dense_ok(&pix_cap, &lds_max);
    for (int t = 0; t < hm().T; t++)
      for (int l = 0; l < pe->hp.n_levels; l++)
        JDA_HIP(launch_stage<Real>(want_trace(), l, t, apply_th, th, pe->dp, pe->hp, model(), w, pix_cap, lds_max, st));
    return true;
  }
  bool clear_counters() {
    JDA_HIP(hipMemsetAsync(w.counters, 0, sizeof(unsigned long long) * kCntShards * kCntStride, st));
    if (want_trace()) {
      JDA_HIP(hipMemsetAsync(w.tr_carts, 0, sizeof(int) * (size_t)windows(), st));
      JDA_HIP(launch_trace_fill<Real>(model(), w, (unsigned)windows(), st));
    }
    return true;
  }
  bool read_counter(int counter) {     // asynchronous: the value is in h_cnt[0] after the next stream sync
    // (the hand-off count comes with the counters up to the mid queue's: h_cnt[kCntMid - kCntTail] = windows k_scan_p
    // put there itself)
    const size_t n = counter == kCntTail ? (size_t)(kCntMid - kCntTail + 1) : 1;
    JDA_HIP(hipMemcpyAsync(h_cnt, w.counters + counter, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    return true;
  }

  // Host frames -> staging buffer, ahead of this pass on its stream.
  bool upload_frames(uint8_t* dst, size_t stride, const unsigned char* const* frames, int n, size_t fbytes) {
    // (small uploads -- single frames of concurrent jdaDetect callers -- stay on the lane: they do not fill the link,
    // and a host wait per call under one mutex would serialise the callers)
    if (!kn().h2d_stream || (long long)n * (long long)fbytes < kn().h2d_min_bytes) return copy_frames_h2d(dst, stride, frames, n, fbytes, st);
    {
      std::lock_guard<std::mutex> lk(c->h2d_mu);
      // (created by the first upload: HIP spreads its streams over four hardware queues in creation order, and a stream
      // that callers with resident frames never use would still shift which lanes share a queue)
      if (!c->h2d && !(c->h2d = c->streams.take(StreamPool::kSide, StreamPool::kNone, nullptr))) return false;
      JDA_HIP(hipEventRecord(ln->ev_h2d[0], st));                // (whatever read the staging buffer before is done)
      JDA_HIP(hipStreamWaitEvent(c->h2d, ln->ev_h2d[0], 0));
      if (!copy_frames_h2d(dst, stride, frames, n, fbytes, c->h2d)) return false;
      if (kn().h2d_stream != 1) JDA_HIP(hipEventRecord(ln->ev_h2d[1], c->h2d));
      else JDA_HIP(hipStreamSynchronize(c->h2d));
    }
    // The pass is enqueued once its frames are up, not behind a device-side wait: HIP multiplexes its streams onto
    // four hardware queues, and a barrier packet that sits out a 1.4-ms upload also stalls whichever other lane shares
    // that queue (seen in the copy/kernel timeline: a lane's second scan launch waiting for the NEXT batch's upload).
    if (kn().h2d_stream == 2) { JDA_HIP(hipStreamWaitEvent(st, ln->ev_h2d[1], 0)); }
    else if (kn().h2d_stream == 3) JDA_HIP(hipEventSynchronize(ln->ev_h2d[1]));
    return true;
  }

  // The persistent form of an LDS-tiled level's scan (k_scan_p.hip): dialect C, no trace.  false = not applicable
  // (the caller launches k_scan).
  // rl: a launch of a ragged chunk (all of ONE level): its tiles come from the chunk's block map, re-cut per image
  // dry: only say whether the level would be taken (nothing is launched, no state changes)
  bool scan_persistent(int level, hipStream_t s, const RaggedChunk::Launch* rl = nullptr, bool dry = false) {
    if constexpr (sizeof(Real) != 4) { (void)level; (void)s; (void)rl; (void)dry; return false; }
    else {
      if (!kn().scan_p || want_trace() || no_scan_p) return false;
      if (rl && (!kn().scan_p_ragged || level < 0)) return false;
      const DevModelT<Real>& m = model();
      const DevLevel& lv = pe->hp.lv[level];
      if (lv.win > kn().scan_p_win_max) return false;
      PScanCfg cfg{};
      int block = 0, wgs = 1;
      const int K = scan_p_base_cfg(c, &cfg, &block, &wgs);
      // all of stage 0 in this kernel: its survivors are what k_filter0 would leave in the mid queue (launch_finishers
      // then takes the k_filter0 + k_finish(survivors) form whatever the size of the hand-off queue)
      cfg.to_mid = (K == m.K && kn().scan_p_mid && filter0_ok()) ? 1 : 0;
      if (!rl) { const unsigned mg = ((1u << 20) + (unsigned)lv.tw - 1u) / (unsigned)lv.tw; bool ok = true;
        for (unsigned i = 0; i < (unsigned)(lv.tw * lv.th + 64) && ok; i++) ok = ((i * mg) >> 20) == i / (unsigned)lv.tw;
        cfg.tw_magic = ok ? (int)mg : 0; }
      cfg.opts = (int)kn().scan_p_opts;
      // the kernel's own cut of the tile in y (same row pitch and tile width: the resolved node offsets hold): small
      // tiles turn over faster and leave room for more slots.  Candidates are the heights whose windows fill their
      // waves to 90 % (or the best filled one); the tallest that keeps the pixel tile within scan_p_tile_kb, else the
      // smallest
      cfg.th = lv.th;
      if (kn().scan_p_tile_kb > 0 && !rl) {
        double top = 0;
        auto fill_of = [&](int th) { const int n = lv.tw * th; return (double)n / (double)(((n + 63) / 64) * 64); };
        for (int th = 1; th <= lv.th; th++) top = std::max(top, fill_of(th));
        const double want = std::min(0.9, top);
        int fit = 0, smallest = 0;
        for (int th = 1; th <= lv.th; th++) {
          if (fill_of(th) < want) continue;
          if (!smallest) smallest = th;
          if ((long long)lv.pitch * (lv.win + (th - 1) * lv.step) <= kn().scan_p_tile_kb * 1024) fit = th;
        }
        cfg.th = fit ? fit : smallest;
      }
      cfg.tiles_y = (lv.ny + cfg.th - 1) / cfg.th;
      cfg.slot_bytes = rl ? ((rl->pix_bytes + 15) & ~15) : ((lv.pitch * (lv.win + (cfg.th - 1) * lv.step) + 15) & ~15);
      // (the cap only where another batch's kernels are in flight next to this pass -- a second lane of this call,
      // other tickets or callers; alone, the workgroup takes every slot that fits)
      const long long n_tiles = rl ? (long long)rl->blk_n : (long long)lv.tiles_x * cfg.tiles_y * nf;
      const long long slots = scan_p_slots_for(c, cfg, K, block, wgs, n_tiles, !solo || busy_lanes > 1);
      if (slots <= 0) return false;
      cfg.slots = (int)slots;
      if (dry) return true;
      cfg.dyn_slot = (kn().scan_p_dyn && p_launches < kCntMidScan - kCntTotal) ? p_launches : -1;
      const int grid = kn().scan_p_grid > 0 ? (int)std::min<long long>(kn().scan_p_grid, 1 << 16) : c->n_cus * wgs;
      const hipError_t e = rl ? launch_scan_persistent(level, cfg, block, grid, pe->dp, pe->hp, m, pe->table, w, s, rl->blk_base, rl->blk_n)
                              : launch_scan_persistent(level, cfg, block, grid, pe->dp, pe->hp, m, pe->table, w, s);
      if (e == hipErrorInvalidValue) { (void)hipGetLastError(); return false; }
      if (e != hipSuccess) { fail(std::string("launch_scan_persistent failed: ") + hipGetErrorString(e)); return false; }
      if (cfg.to_mid) mid_direct = true;
      p_launches++;
      return true;
    }
  }

  // step 1: pyramids (multi-scale models), stage-0 scan (or everything, in dense mode)
  bool issue_scan(uint8_t* hbuf, size_t hs, uint8_t* qbuf, size_t qs, hipEvent_t scan_after) {
    constexpr int dialect = Sel<Real>::dialect;
    const DevModelT<Real>& m = model();
    a_hbuf = hbuf; a_hs = hs; a_qbuf = qbuf; a_qs = qs;
    if (timed && !(rag && rag->images_issued)) JDA_HIP(hipEventRecord(ev[0], st));     // (else: recorded in front of the images' repack, ragged.cpp)
    if (rag) return issue_scan_ragged();
    if (host_frames && !upload_frames(const_cast<uint8_t*>(w.frames), w.frame_stride, host_frames, nf, host_fbytes)) return false;
    if (multi && w.patch_hs > 0) {             // method 0: every window's ROI -> its half_size / quarter_size patches (cascador.cpp:243-245)
      const DevLevel& lv = pe->hp.lv[0];
      JDA_HIP(launch_resize_cv_patches(w.frames, w.frame_stride, nf, pe->sp.width, lv.nx, lv.ny, lv.step, lv.win, hbuf, hs, w.patch_hs, st));
      JDA_HIP(launch_resize_cv_patches(w.frames, w.frame_stride, nf, pe->sp.width, lv.nx, lv.ny, lv.step, lv.win, qbuf, qs, w.patch_qs, st));
      w.half = hbuf; w.half_stride = hs; w.quarter = qbuf; w.quarter_stride = qs;
    } else if (multi) {
      const int W = pe->sp.width, H = pe->sp.height;
      const size_t stride = w.frame_stride;
      if (dialect == JDA_DIALECT_C) {      // jdaImageResize, c/jda.c:203-230
        JDA_HIP(launch_resize(w.frames, stride, nf, W, H, hbuf, hs, w.hw, w.hh, (float)(W - 1) / w.hw, (float)(H - 1) / w.hh, st));
        JDA_HIP(launch_resize(w.frames, stride, nf, W, H, qbuf, qs, w.qw, w.qh, (float)(W - 1) / w.qw, (float)(H - 1) / w.qh, st));
      } else {                             // cv::resize, cascador.cpp:330-331
        JDA_HIP(launch_resize_cv(w.frames, stride, nf, W, H, hbuf, hs, w.hw, w.hh, st));
        JDA_HIP(launch_resize_cv(w.frames, stride, nf, W, H, qbuf, qs, w.qw, w.qh, st));
      }
      w.half = hbuf; w.half_stride = hs; w.quarter = qbuf; w.quarter_stride = qs;
    }
    if (!clear_counters()) return false;
    // ---- dense mode (k_stage): when most windows survive the first carts, whole stages are
    //      walked tile by tile instead of window by window.  Decided from the previous pass on
    //      this plan (pe->dense_hint) or, in after_tail, from the hand-off count of this pass;
    //      the results do not depend on the choice. ----
    int pix_cap, lds_max;
    const bool ok = dense_ok(&pix_cap, &lds_max);
    dense = ok && (kn().dense == 2 || hint_dense);
    if (dense) {
      if (!grow_for_dense() || !clear_counters()) return false;     // (the counters moved with the workspace)
      if (timed) JDA_HIP(hipEventRecord(ev[1], st));
      if (timed) JDA_HIP(hipEventRecord(ev[2], st));
      finished = true;
      return run_dense();
    }
    // ---- windows k_scan does not cover enter the hand-off queue at cart 0 ----
    if (!pe->fast_scan || pe->any_untiled) JDA_HIP(launch_enqueue<Real>(pe->dp, pe->hp, !pe->fast_scan, w, st));
    // ---- stage-0 scan: first `handoff` carts, one launch per LDS-tiled level ----
    // (staggering a lane's scan behind the previous lane's was measured SLOWER than letting both scans share the
    // machine: 2.65 ms vs 2.39 ms per 256-frame step -- half-size scans are less efficient and k_finish is
    // throughput bound itself)
    (void)scan_after;
    if (timed) JDA_HIP(hipEventRecord(ev[1], st));
    if (pe->fast_scan) {
      const int handoff = (int)kn().handoff;
      const int cp_max = (int)std::max<long long>(0, std::min<long long>(256, kn().cp_max));
      bool any_glb = false, any_wide = false, side_pending = false;
      long long lds_blocks = 0;
      for (int l = 0; l < pe->hp.n_levels; l++) {
        if (pe->hp.lv[l].tiled == 2) any_glb = true;
        if (pe->hp.lv[l].tiled == 3) any_wide = true;
        if (pe->hp.lv[l].tiled == 1) lds_blocks += (long long)pe->hp.lv[l].tiles_x * pe->hp.lv[l].tiles_y * nf;
      }
      auto scan = [&](int mode, int level, hipStream_t s) -> bool {
        // (opts bit 0, 8 trees in flight per lane in the LDS-tiled modes, was measured neutral to slower: off)
        const int opts = ((int)(std::max<long long>(4, std::min<long long>(64, kn().first_phase)) & ~3LL) << 8) |
                         (kn().scan_lean && !stage0_any_norm(c->hm, handoff, sizeof(Real) == 4) ? 2 : 0);
        if (mode == 1 && level >= 0 && scan_persistent(level, s)) { rs->scan_launches++; my_scan_launches++; return true; }
        JDA_HIP(launch_scan<Real>(mode, level, want_trace(), handoff, cp_max, opts, pe->dp, pe->hp, m, pe->table, w, s));
        rs->scan_launches++; my_scan_launches++;
        return true;
      };
      // the global-pixel launch of a lone lane goes to a side stream, forked here and joined before the
      // hand-off count is read, so that it runs next to the LDS-tiled launches (with two lanes the other
      // lane already provides that mix; measured slower there)
      auto fork_glb = [&]() -> bool {
        hipStream_t sd = ln->side;
        JDA_HIP(hipEventRecord(ln->ev_side[0], st));
        JDA_HIP(hipStreamWaitEvent(sd, ln->ev_side[0], 0));
        if (!scan(2, -1, sd)) return false;
        JDA_HIP(hipEventRecord(ln->ev_side[1], sd));
        any_glb = false;
        side_pending = true;
        return true;
      };
      const bool small = lds_blocks <= kn().merge_blocks;
      if (small) {
        // small job (a frame or a few): all levels of a pixel mode in one launch -- every workgroup
        // is resident at once anyway, so per-level launches would only serialise their latency
        // (a side stream per caller costs concurrent single-frame callers throughput: only while the cascador is
        // otherwise quiet, like k_finish_wide)
        if (any_glb && solo && kn().side_small && busy_lanes <= kn().wide_busy_max && ln->ensure_side() && !fork_glb()) return false;
        if (lds_blocks > 0 && !scan(1, -1, st)) return false;
        if (any_wide && !scan(3, -1, st)) return false;
      } else {
        // odd lanes go through the levels in the opposite order (big windows first): the launches of
        // one lane then run next to different ones of the other instead of next to their twins
        const bool rev = (lane & 1) && kn().lanes_reverse;
        // (... and only while this is the cascador's only pass in flight: next to another ticket's pass the other pass is
        // the mix, and the fork costs -- submit/wait step 1.41 -> 1.36 ms with three tickets, 1.47 -> 1.34 with two, once
        // every lane has a hardware queue of its own; r06, profiles/r06_hwq.txt section 6)
        const bool side = any_glb && solo && kn().side_stream && busy_lanes <= 1 && ln->ensure_side();
        int fork_in = side ? (int)std::max<long long>(0, kn().side_after) : -1;
        if (fork_in == 0) { if (!fork_glb()) return false; fork_in = -1; }
        if (rev && any_glb) { if (!scan(2, -1, st)) return false; any_glb = false; }
        // The LDS-tiled levels the persistent kernel leaves to k_scan's closed tiles (the 71- and 88-pixel levels of
        // 640x480: too few slots) share ONE launch when they lie next to each other: a launch of ~1,800 workgroups is
        // three and a half rounds of the 512 resident ones, two of them back to back pay the partial round twice.
        // Only levels of one occupancy class (workgroups per CU by their LDS, workgroup size) merge: a launch takes the LDS of its
        // largest tile, and dialect CPP's ten closed-tile levels in ONE launch ran the small-window levels at the big ones'
        // occupancy (uniform 256-frame batch 10.4 -> 11.5-12.2 ms, r06).  run_first[l] / run_last[l]: the run level l belongs to.
        int run_first[kMaxLevels], run_last[kMaxLevels];
        for (int l = 0; l < pe->hp.n_levels; l++) run_first[l] = run_last[l] = -1;
        // (dialect C only: the fp64 batch's thirteen closed-tile levels in three class launches measured no shorter on the
        // device and 4 % longer per call -- its two lanes interleave their per-level launches better, session r06_s20)
        if (!want_trace() && sizeof(Real) == 4) {
          const int chunk = std::min(std::min(m.K, handoff), scan_handoff_cap(m.node_n, m.leaf_n, (int)sizeof(Real)));
          int cur_first = -1, cur_key = -1, prev = -1;
          auto close = [&](int last) { for (int l = cur_first; cur_first >= 0 && l <= last; l++) if (run_first[l] == cur_first) run_last[l] = last; };
          for (int l = 0; l < pe->hp.n_levels; l++) {
            const DevLevel& lv = pe->hp.lv[l];
            if (lv.tiled != 1) continue;
            if (scan_persistent(l, st, nullptr, true)) { close(prev); cur_first = -1; cur_key = -1; prev = -1; continue; }
            const int block = lv.tw * lv.th > 256 ? 512 : 256;
            const int key = lds_wgs_per_cu((long long)scan_lds_bytes(lv.pitch * (lv.win + (lv.th - 1) * lv.step), chunk, m.node_n, m.leaf_n, (int)sizeof(Real), false, block)) * 1024 + block;
            if (key != cur_key) { close(prev); cur_first = l; cur_key = key; }
            run_first[l] = cur_first; prev = l;
          }
          close(prev);
        }
        for (int li = 0; li < pe->hp.n_levels; li++) {
          const int l = rev ? pe->hp.n_levels - 1 - li : li;
          const int mode = pe->hp.lv[l].tiled;
          if (mode != 1 && mode != 3) continue;
          // big-window levels of a batch are short launches: merge all of them into one (at the first one met)
          if (mode == 3) { if (any_wide) { if (!scan(3, -1, st)) return false; any_wide = false; } continue; }
          if (run_first[l] >= 0 && run_last[l] > run_first[l]) {
            // (the run's launch goes where its first member -- in this lane's order -- stands)
            const int lead = rev ? run_last[l] : run_first[l];
            if (l == lead) {
              const int opts = ((int)(std::max<long long>(4, std::min<long long>(64, kn().first_phase)) & ~3LL) << 8) |
                               (kn().scan_lean && !stage0_any_norm(c->hm, handoff, sizeof(Real) == 4) ? 2 : 0);
              JDA_HIP(launch_scan<Real>(1, -1, false, handoff, cp_max, opts, pe->dp, pe->hp, m, pe->table, w, st, run_first[l], run_last[l] + 1));
              rs->scan_launches++; my_scan_launches++;
              if (fork_in > 0 && --fork_in == 0) { if (!fork_glb()) return false; fork_in = -1; }
            }
            continue;
          }
          if (!scan(1, l, st)) return false;
          if (fork_in > 0 && --fork_in == 0) { if (!fork_glb()) return false; fork_in = -1; }
        }
        if (fork_in > 0 && !fork_glb()) return false;
      }
      lds_span = !side_pending && !(((lane & 1) && kn().lanes_reverse) && !small);   // LDS launches first, back to back
      if (lds_span && timed) JDA_HIP(hipEventRecord(ev[4], st));
      if (any_glb && !scan(2, -1, st)) return false;
      if (side_pending) JDA_HIP(hipStreamWaitEvent(st, ln->ev_side[1], 0));
    }
    if (timed) JDA_HIP(hipEventRecord(ev[2], st));
    return issue_rest();
  }

  // With a prediction of the hand-off queue's length (earlier passes on this plan) everything else is queued
  // right behind the scan: finishing launches sized by the prediction, counters and a predicted prefix of
  // the detections -> host.  The pass is then one enqueue and ONE host wait (after_counters).  Without one, the
  // host reads the queue length first (after_tail).
  bool issue_rest() {
    int pix_cap, lds_max;
    if (kn().predict && pred_tail >= 0 && !(kn().dense == 1 && dense_ok(&pix_cap, &lds_max) && pred_tail + std::max(0.0, pred_mid) >= 0.4)) {
      const long long nw = windows();
      const long long guess = std::min<long long>((long long)cap_q, (long long)(pred_tail * (double)nw * 1.1) + 64);
      if (!launch_finishers(guess)) return false;
      predicted = true;
      const double po = pred_out >= 0 ? pred_out : 0.0;
      const size_t to = std::min<size_t>(cap_m, (size_t)(po * (double)nw * 1.25) + 64);
      bool ok;
      if (sizeof(Real) == 4 && want_post && kn().kernel_d2h && dets && to > 0 && !want_trace() && !dense) {
        // dialect C, uniform batch: scan order, score order, NMS and relocation per frame on the device, results straight
        // into pinned memory (k_post); a frame or a row count it declines sends the pass through the host path below
        ok = issue_post(to) && issue_counters();
      } else if (kn().kernel_d2h && dets && to > 0) ok = issue_results(0, to, true);      // counters + prefix in one launch
      else ok = issue_counters() && issue_results(0, to);
      return ok;
    }
    // the hand-off queue length sizes the finishing launches (one workgroup per window)
    return read_counter(kCntTail);
  }

  // Ragged pass: tables and images -> device (tight rows repacked to the common pitch), then the scan launches of the
  // chunk's block map.  Never dense, never traced (the caller falls back to per-image passes for those).
  bool issue_scan_ragged() {
    const DevModelT<Real>& m = model();
    const RaggedChunk& ch = *rag;
    uint8_t* tab = (uint8_t*)ln->rag_tab.p;
    if (ch.images_issued) {
      // (the image records and k_repack are on the stream already: segments and block map follow)
      JDA_HIP(hipMemcpyAsync(tab + ch.images_bytes, (const uint8_t*)ln->h_tab.p + ch.images_bytes, ch.table_bytes - ch.images_bytes, hipMemcpyHostToDevice, st));
    } else {
      JDA_HIP(hipMemcpyAsync(tab, ln->h_tab.p, ch.table_bytes, hipMemcpyHostToDevice, st));
      const uint8_t* raw = ch.d_raw;
      if (ch.d_uploaded) {
        raw = ch.d_uploaded;              // (detect_ragged waited for the upload on the host before it called this)
      } else if (ch.host_imgs) {
        // tight images -> device: one copy when they lie back to back in the caller's memory, else through the lane's
        // pinned staging buffer (filled by build_chunk)
        const void* src = ch.host_contiguous ? (const void*)ch.host_imgs[0] : ln->h_raw.p;
        JDA_HIP(hipMemcpyAsync(ln->rag_raw.p, src, ch.raw_bytes, hipMemcpyHostToDevice, st));
        raw = (const uint8_t*)ln->rag_raw.p;
      }
      JDA_HIP(launch_repack(raw, (uint8_t*)ln->rag_frames.p, (const RagImg*)(tab + ch.off_rimg), ch.n, ch.max_h, ch.pitch, st));
    }
    w.frames = (const uint8_t*)ln->rag_frames.p; w.frame_stride = 0; w.n_frames = ch.n;
#ifdef JDA_BOUNDS_CHECK
    w.bc_lo = w.frames; w.bc_hi = w.frames + ch.frame_bytes;       // (bounds-check build: the staged images of the chunk)
#endif
    w.segs = (const RagSeg*)(tab + ch.off_segs); w.blk = (const RagBlk*)(tab + ch.off_blk);
    w.img_off = (const unsigned long long*)(tab + ch.off_imgoff);
    if (!clear_counters()) return false;
    if (timed) JDA_HIP(hipEventRecord(ev[1], st));
    const int handoff = (int)kn().handoff;
    const int cp_max = (int)std::max<long long>(0, std::min<long long>(256, kn().cp_max));
    const int opts = ((int)(std::max<long long>(4, std::min<long long>(64, kn().first_phase)) & ~3LL) << 8) |
                     (kn().scan_lean && !stage0_any_norm(c->hm, handoff, sizeof(Real) == 4) ? 2 : 0);
    // A job that is ONE chunk (a rank's shard of a sharded job: 4 M windows) has no other chunk's kernels next to its own,
    // and its launches -- a few hundred to two thousand workgroups each -- do not fill the machine one after the other:
    // the global-pixel launch is forked to the lane's side stream, next to the LDS-tiled ones (r06: the scan chain of a
    // 356-image shard 0.78 -> 0.6 ms), like the lone lane of a uniform pass does.
    bool side_pending = false, any_glb = false;
    for (const RaggedChunk::Launch& l : ch.launches) any_glb = any_glb || l.mode == 2;
    const bool fork_glb = any_glb && solo && kn().side_stream && kn().ragged_side && ch.launches.size() > 1 && busy_lanes <= kn().wide_busy_max && ln->ensure_side();
    if (fork_glb) {       // (forked HERE, in front of the LDS-tiled launches: the side stream only waits for the images and the counters)
      JDA_HIP(hipEventRecord(ln->ev_side[0], st));
      JDA_HIP(hipStreamWaitEvent(ln->side, ln->ev_side[0], 0));
    }
    // (ragged_side = 2: the closed-tile LDS launches -- the levels the persistent kernel declines -- follow the global-pixel
    // launch on the side stream, so that the lane's own stream carries the persistent launches only)
    auto issue = [&](const RaggedChunk::Launch& l, hipStream_t s, bool try_persistent) -> bool {
      // (the persistent form for the levels it suits, as in a uniform pass: one workgroup per CU walks the level's tiles
      // of every image of the chunk through its slots)
      if (try_persistent && l.mode == 1 && l.level >= 0 && scan_persistent(l.level, s, &l)) { rs->scan_launches++; my_scan_launches++; return true; }
      JDA_HIP(launch_scan_ragged<Real>(l.mode, l.block, false, handoff, cp_max, opts, pe->dp, m, pe->table, w, l.pix_bytes,
                                       l.blk_base, l.blk_n, s));
      rs->scan_launches++; my_scan_launches++;
      return true;
    };
    if (fork_glb) {
      for (const RaggedChunk::Launch& l : ch.launches)
        if (l.mode == 2) { if (!issue(l, ln->side, false)) return false; side_pending = true; }
    }
    for (const RaggedChunk::Launch& l : ch.launches) {
      if (l.mode == 2 && fork_glb) continue;
      if (fork_glb && kn().ragged_side == 2 && l.mode != 2) {
        // would the persistent kernel take it?  (asked by trying: a declined level costs nothing)
        if (l.mode == 1 && l.level >= 0 && scan_persistent(l.level, st, &l)) { rs->scan_launches++; my_scan_launches++; continue; }
        if (!issue(l, ln->side, false)) return false;
        continue;
      }
      if (!issue(l, st, true)) return false;
    }
    if (side_pending) JDA_HIP(hipEventRecord(ln->ev_side[1], ln->side));
    if (side_pending) JDA_HIP(hipStreamWaitEvent(st, ln->ev_side[1], 0));
    if (timed) JDA_HIP(hipEventRecord(ev[2], st));
    return issue_rest();
  }

  // Finishing launches for a hand-off queue of (about) n_grid windows: the kernels take the true length from the
  // device counter and stride over it, n_grid only sizes the grids.
  bool launch_finishers(long long n_grid) {
    const int T = hm().T;
    const int gm = kn().fin_gm > 0 ? (int)kn().fin_gm : stage_groups();
    const int g2 = kn().fin_g2 > 0 ? (int)kn().fin_g2 : stage_groups();
    n_grid = std::max<long long>(n_grid, 1);
    if (mid_direct) {
      // the mid queue already holds stage-0 survivors (k_scan_p): the rest of the hand-off queue is filtered into it,
      // then everybody goes through k_finish(survivors)
      const long long nmid = pred_mid >= 0 ? (long long)(pred_mid * (double)windows() * 1.25) + 64 : 0;
      const long long wg2 = std::min<long long>((long long)cap_m, std::max<long long>(std::max<long long>(2048, n_grid / std::max<long long>(1, kn().fin_grid_div)), nmid));
      JDA_HIP(launch_filter0<Real>(want_trace(), pe->dp, model(), w, n_grid, s0_tbl(), st));
      JDA_HIP(launch_finish<Real>(want_trace(), 0, T, apply_th, th, pe->dp, model(), w, g2, wg2, s0_tbl(), (int)kn().fin_tile, st, true));
      finished = true;
      return true;
    }
    // (k_finish_wide is the LATENCY form: a whole CU per window.  With several callers on the cascador at once the
    // machine is shared and throughput counts: they get the one-wave-per-window kernel)
    if (n_grid <= kn().wide_max && busy_lanes <= kn().wide_busy_max && finish_wide_ok(hm().dim(), hm().K, hm().leaf_n(), (int)sizeof(Real), multi, Sel<Real>::dialect == JDA_DIALECT_CPP && c->similarity)) {
      // a small job (a frame or a few): the call's time is the latency of one window's chain through the stages --
      // every queued window gets a whole workgroup (k_wide.hip)
      JDA_HIP(launch_finish_wide<Real>(want_trace(), apply_th, th, pe->dp, model(), w, n_grid, s0_tbl(), st, kn().wide_conc != 0));
      finished = true;
      return true;
    }
    // (a trainer snapshot with the similarity transform: its stage in training walks with the parameter the stage before it
    // computed, which k_finish keeps in its scratch -- every stage of a window in ONE launch)
    const bool st_snapshot = sizeof(Real) == 8 && c->similarity && hm().hdr_stage >= 0 && hm().hdr_stage < T;
    if (T == 1 || n_grid <= kn().finish_merge || st_snapshot) {
      // few windows left: one launch walks them through every remaining stage (no balance problem,
      // one launch less)
      JDA_HIP(launch_finish<Real>(want_trace(), 0, T, apply_th, th, pe->dp, model(), w, gm, n_grid, s0_tbl(), (int)kn().fin_tile, st));
      finished = true;
      return true;
    }
    const long long wg2 = std::min<long long>(n_grid, std::max<long long>(2048, n_grid / std::max<long long>(1, kn().fin_grid_div)));
    if (filter0_ok()) {
      // the dying majority is filtered by a lean kernel (four windows per workgroup, stage 0 only); the survivors --
      // a few per cent -- go through k_finish for the regression of stage 0 and every later stage
      JDA_HIP(launch_filter0<Real>(want_trace(), pe->dp, model(), w, n_grid, s0_tbl(), st));
      JDA_HIP(launch_finish<Real>(want_trace(), 0, T, apply_th, th, pe->dp, model(), w, g2, wg2, s0_tbl(), (int)kn().fin_tile, st, true));
      finished = true;
      return true;
    }
    // Two launches so that the few windows that pass stage 0 (and then cost whole stages each) are spread over
    // the machine again.  The second is queued right behind the first, without a host round trip for the length
    // of the mid queue (the kernel reads it from the device counter): its grid is a quarter of the hand-off count
    // -- one workgroup per window as long as fewer than 25 % pass stage 0 (6.7 % in the cascade regime), a grid-stride
    // loop beyond that; the surplus workgroups exit at once (an empty workgroup costs ~1.3 ns of dispatcher time).
    JDA_HIP(launch_finish<Real>(want_trace(), 0, 1, apply_th, th, pe->dp, model(), w, (int)kn().fin_g1, n_grid, s0_tbl(), (int)kn().fin_tile1, st));
    JDA_HIP(launch_finish<Real>(want_trace(), 1, T, apply_th, th, pe->dp, model(), w, g2, wg2, nullptr, (int)kn().fin_tile, st));
    finished = true;
    return true;
  }

  // step 2 (passes without a prediction): every survivor of the scan: remaining carts of stage 0 (+ all stages
  // when few are left)
  bool after_tail() {
    if (finished) return true;
    JDA_HIP(hipStreamSynchronize(st));
    // (the counters count every window the scan kept, also those a queue had no room for)
    const unsigned long long true_tail = h_cnt[0], true_mid = mid_direct ? h_cnt[kCntMid - kCntTail] : 0ull;
    n_tail = (long long)std::min<unsigned long long>(true_tail, cap_q);
    const long long n_alive = (long long)(true_tail + true_mid);
    int pix_cap, lds_max;
    const double dense_frac = (double)kn().dense_pct / 100.0;
    if (dense_ok(&pix_cap, &lds_max) && (double)n_alive >= dense_frac * (double)windows() && n_alive > 4096) {
      // most windows are still alive after the scan: start over in dense mode (the scan's work
      // is a small part of T*K carts per window) and remember the choice for the next pass
      { std::lock_guard<std::mutex> lk(c->mu); pe->dense_hint = true; }
      if (!rag) {                                  // (a ragged pass finishes window by window; the NEXT job runs image by image, dense)
        dense = true; finished = true;
        if (!grow_for_dense() || !clear_counters()) return false;
        return run_dense();
      }
    }
    if (true_tail > cap_q || true_mid > cap_m) return recover_overflow(true_tail, true_mid, 0);
    return launch_finishers(n_tail);
  }

  // step 3: (nothing left to wait for between the two finishing launches)
  bool after_mid() { return true; }

  // step 4: counters -> host (asynchronous)
  bool issue_counters() {
    if (counters_issued) return true;
    counters_issued = true;
    if (timed) JDA_HIP(hipEventRecord(ev[3], st));
    if (kn().kernel_d2h) {
      const void* src[1] = {w.counters}; void* dst[1] = {h_cnt};
      const size_t nb[1] = {sizeof(unsigned long long) * kCntShards * kCntStride};
      JDA_HIP(launch_copy_out(src, dst, nb, 1, st));
      return true;
    }
    JDA_HIP(hipMemcpyAsync(h_cnt, w.counters, sizeof(unsigned long long) * kCntShards * kCntStride, hipMemcpyDeviceToHost, st));
    return true;
  }

  // k_post for this pass: at most `rows` detections kept in all
  bool issue_post(size_t rows) {
    if constexpr (sizeof(Real) == 4) {
      const int dim = hm().dim();
      if (!ln->h_pn.reserve(((size_t)2 * nf + 4) * sizeof(int)) || !ln->h_pbb.reserve(rows * 3 * sizeof(int)) ||
          !ln->h_psc.reserve(rows * sizeof(float)) || !ln->h_psh.reserve(rows * dim * sizeof(float))) return false;
      int* pn = (int*)ln->h_pn.p;
      pn[2 * nf] = 0;                         // the kernel's "declined" flag (the lane's last pass has been collected)
      PostOut o;
      o.n = pn; o.first = pn + nf; o.flag = pn + 2 * nf;
      o.bb = (int*)ln->h_pbb.p; o.score = (float*)ln->h_psc.p; o.shape = (float*)ln->h_psh.p;
      o.cursor = w.counters + (size_t)8 * kCntStride + kCntPostCursor;
      o.cap_rows = (unsigned)std::min<size_t>(rows, 0x7fffffffu);
      const uint32_t* rag_gid = nullptr; const RagImg* rag_img = nullptr;
      if (rag) {                                       // (the chunk's tables are on the device: issue_scan_ragged)
        const uint8_t* tab = (const uint8_t*)ln->rag_tab.p;
        rag_gid = (const uint32_t*)(tab + rag->off_gidbase); rag_img = (const RagImg*)(tab + rag->off_rimg);
      }
      JDA_HIP(launch_post(pe->dp, w, dim, nf, post_nms, post_overlap, o, st, rag_gid, rag_img));
      post_issued = true; post_cap = rows;
      return true;
    } else {
      (void)rows;
      return false;
    }
  }

  // detections [from, to) of the device list -> the lane's pinned host arrays (asynchronous)
  bool issue_results(size_t from, size_t to, bool with_counters = false) {
    const int dim = hm().dim();
    if (!dets || to <= from) return true;
    HostPinned &hg = ln->h_gid, &hs = ln->h_score, &hh = ln->h_shape;
    if (!hg.reserve(to * 4, from * 4) || !hs.reserve(to * sizeof(Real), from * sizeof(Real)) ||
        !hh.reserve(to * dim * sizeof(Real), from * dim * sizeof(Real))) return false;
    const size_t n = to - from;
    if (kn().kernel_d2h && from == 0) {          // (a 16-byte aligned start: the predicted prefix; a later rest goes by the copy engine)
      const void* src[4] = {w.out_gid, w.out_score, w.out_shape, w.counters};
      void* dst[4] = {hg.p, hs.p, hh.p, h_cnt};
      const size_t nb[4] = {n * 4, n * sizeof(Real), n * dim * sizeof(Real), sizeof(unsigned long long) * kCntShards * kCntStride};
      if (with_counters) { counters_issued = true; if (timed) JDA_HIP(hipEventRecord(ev[3], st)); }
      JDA_HIP(launch_copy_out(src, dst, nb, with_counters ? 4 : 3, st));
      out_copied = to;
      results_pending = true;
      return true;
    }
    JDA_HIP(hipMemcpyAsync((uint32_t*)hg.p + from, w.out_gid + from, n * 4, hipMemcpyDeviceToHost, st));
    JDA_HIP(hipMemcpyAsync((Real*)hs.p + from, w.out_score + from, n * sizeof(Real), hipMemcpyDeviceToHost, st));
    JDA_HIP(hipMemcpyAsync((Real*)hh.p + from * dim, w.out_shape + from * dim, n * dim * sizeof(Real), hipMemcpyDeviceToHost, st));
    out_copied = to;
    results_pending = true;
    return true;
  }

  // step 5: statistics, (the rest of) the detections -> host (asynchronous)
  bool after_counters() {
    const int T = hm().T;
    if (counted) return true;        // (a rerun inside after_tail has already been through here: the counters are folded and tallied ONCE)
    JDA_HIP(hipStreamSynchronize(st));
    results_pending = false;
    for (int shd = 1; shd < kCntShards; shd++) {   // fold the counter shards into shard 0
      for (int i = 0; i < kCntTotal; i++) h_cnt[i] += h_cnt[shd * kCntStride + i];
      h_cnt[kCntMidScan] += h_cnt[shd * kCntStride + kCntMidScan];
    }
    if (p_launches > 0 && !dense && !no_scan_p) {
      // The persistent scan ran in this pass: did its watchdogs stay quiet, and did it cover every window it was given?
      // (k_scan_p.hip: a tripped launch loses windows, it never corrupts one -- so the check is a count)
      const unsigned long long err = h_cnt[(size_t)kCntScanErrShard * kCntStride + kCntScanErr];
      long long expect = 0;
      if (rag) expect = rag->windows;          // (a ragged job has no untiled level: ragged_prepare)
      else
        for (int l = 0; l < pe->hp.n_levels; l++)
          if (pe->hp.lv[l].tiled != 0) expect += (long long)pe->hp.lv[l].nx * pe->hp.lv[l].ny * nf;
      if (err != 0 || (long long)h_cnt[kCntWinScan] != expect) return recover_scan(err, (long long)h_cnt[kCntWinScan], expect);
    }
    if (!dense && (h_cnt[kCntTail] > cap_q || h_cnt[kCntMid] > cap_m || h_cnt[kCntOut] > cap_m))
      return recover_overflow(h_cnt[kCntTail], h_cnt[kCntMid], h_cnt[kCntOut]);
    rs->carts += (long long)h_cnt[kCntCarts];
    rs->carts_scan += (long long)h_cnt[kCntCartsScan];
    rs->carts_scan_glb += (long long)h_cnt[kCntCartsScanGlb];
    rs->win_scan += (long long)h_cnt[kCntWinScan];
    for (int t = 0; t < T; t++) rs->stage_done[t] += (long long)h_cnt[kCntStage0 + t];
    rs->tail += (long long)h_cnt[kCntTail] + (long long)h_cnt[kCntMidScan];     // (alive at the scan's hand-off, whichever queue took them)
    const double nw = (double)windows();
    const double dense_frac = (double)kn().dense_pct / 100.0;
    n_tail = (long long)h_cnt[kCntTail];
    n_out = (size_t)h_cnt[kCntOut];
    rs->out += (long long)n_out;
    if (n_out > cap_m) { fail("internal: more detections than the detection list holds"); return false; }
    if (dense) rs->dense_passes++;
    {
      std::lock_guard<std::mutex> lk(c->mu);            // the plan and the cascador's hints are shared with concurrent callers
      if (dense) {
        // fall back to the sparse pipeline when stage 0 rejects most windows after all
        if ((double)h_cnt[kCntStage0] < 0.5 * dense_frac * nw) pe->dense_hint = false;
      } else {
        // what the next pass on this plan (and a new plan of this cascador) may expect; a prediction decays slowly,
        // so that one quiet batch does not undersize the launches of the next busy one
        const double ft = (double)h_cnt[kCntTail] / nw;
        pe->pred_tail = std::max(ft, pe->pred_tail * 0.9);
        c->pred_tail = pe->pred_tail;
        pe->pred_out = std::max((double)n_out / nw, pe->pred_out * 0.9);
        c->pred_out = pe->pred_out;
        pe->pred_mid = std::max((double)h_cnt[kCntMid] / nw, pe->pred_mid * 0.9);
        int pix_cap, lds_max;
        const double f_alive = ft + (mid_direct ? (double)h_cnt[kCntMid] / nw : 0.0);     // (alive after the scan, or more)
        if (predicted && kn().dense == 1 && dense_ok(&pix_cap, &lds_max) && f_alive >= dense_frac && f_alive * nw > 4096)
          pe->dense_hint = true;       // this pass went through k_finish window by window; the next one runs dense
      }
      c->last_dense = pe->dense_hint;
    }
    counted = true;
    if (post_issued) {
      posted = ((const int*)ln->h_pn.p)[2 * nf] == 0;
      if (posted) return true;                 // (nothing else to fetch: the frames' results are in pinned memory)
    }
    if (n_out > out_copied && !issue_results(out_copied, n_out)) return false;   // the prediction fell short (or there was none)
    return true;
  }

  // The persistent scan of this pass gave up (watchdog) or came back short: nothing of the pass has been counted or
  // collected yet, so the whole pass is issued again on the same lane with k_scan's closed tiles, and the caller gets
  // correct results, a note on stderr and jdaStats::scan_fallbacks (the call itself succeeds: jdaGetLastError() stays
  // empty -- an empty result with a non-empty error string is jdaDetect's only failure signal).
  bool recover_scan(unsigned long long err, long long got, long long expect) {
    char msg[256];
    std::snprintf(msg, sizeof msg, "k_scan_p: watchdog word %llu, %lld of %lld windows covered -- pass of %d frame(s) run again with k_scan",
                  err, got, expect, nf);
    std::fprintf(stderr, "libjda: %s\n", msg);
    no_scan_p = true;
    rs->scan_launches -= my_scan_launches; my_scan_launches = 0;
    dense = false; finished = false; lds_span = false; predicted = false; counters_issued = false; results_pending = false;
    p_launches = 0; post_issued = false; posted = false; post_cap = 0; mid_direct = false; n_tail = -1; n_out = 0; out_copied = 0; counted = false;
    host_frames = nullptr;                 // (already in the staging buffer)
    if (!issue_scan(a_hbuf, a_hs, a_qbuf, a_qs, nullptr) || !after_tail() || !issue_counters() || !after_counters()) return false;
    rs->scan_fallbacks++;                  // (jdaStats::scan_fallbacks: the error channel stays for errors)
    return true;
  }

  // A queue of this pass was too small for what the scan (or a finishing kernel) kept: the kernels dropped what did not
  // fit and kept counting, so nothing of the pass is usable but its counts.  The workspace grows to them -- a queue
  // downstream of the one that overflowed has only seen a part of its input: its count is scaled up -- and the whole pass
  // is issued again on the same lane; a third attempt takes the worst-case sizes.  The caller gets correct results, a note
  // on stderr and jdaStats::ws_regrows; the plan's fractions are updated by the rerun, so the next pass is sized right.
  bool recover_overflow(unsigned long long tail, unsigned long long mid, unsigned long long out) {
    overflow_runs++;
    const size_t nw = (size_t)windows();
    size_t nq = cap_q, nm = cap_m;
    // (a scan that keeps a quarter of its windows or more is no cascade: the worst-case sizes at once, not in two steps)
    if (overflow_runs >= 3 || tail * 4 > nw) { nq = nw; nm = nw; }
    else {
      const double up = tail > cap_q ? (double)tail / (double)std::max<size_t>(1, cap_q) : 1.0;   // what the truncated hand-off queue hid from the later counts
      if (tail > cap_q) nq = std::min(nw, (size_t)((double)tail * 1.25) + 64);
      const double need_m = (double)std::max(mid, out) * up;
      if (need_m > (double)cap_m || tail > cap_q) nm = std::min(nw, std::max(cap_m, (size_t)(need_m * 1.5) + 64));
    }
    std::fprintf(stderr, "libjda: a queue of a pass over %zu windows was too small (hand-off %llu of %zu, mid %llu / detections %llu of %zu) -- "
                         "workspace grown to %zu / %zu entries, pass run again\n", nw, tail, cap_q, mid, out, cap_m, nq, nm);
    JDA_HIP(hipStreamSynchronize(st));
    if (ln->side) JDA_HIP(hipStreamSynchronize(ln->side));
    if (!ensure_workspace<Real>(ln, std::max(cap, nw), want_trace(), hm().dim(), nq, nm, ln->dense_ws)) return false;
    adopt_workspace();
    rs->scan_launches -= my_scan_launches; my_scan_launches = 0;
    dense = false; finished = false; lds_span = false; predicted = false; counters_issued = false; results_pending = false;
    p_launches = 0; post_issued = false; posted = false; post_cap = 0; mid_direct = false; n_tail = -1; n_out = 0; out_copied = 0; counted = false;
    host_frames = nullptr;                 // (already in the staging buffer)
    pred_tail = -1;                        // (no prediction for the rerun: the host reads the hand-off count first)
    rs->ws_regrows++;
    return issue_scan(a_hbuf, a_hs, a_qbuf, a_qs, nullptr) && after_tail() && issue_counters() && after_counters();
  }

  // step 6: detections of this pass sorted back into scan order and appended; trace arrays
  bool collect() {
    const int dim = hm().dim();
    const long long wpf = rag ? 0 : pe->sp.windows;
    const double t_dbg = now_ms();
    if (posted && dets) {
      // the frames of this pass as k_post left them: rows appended, first rows rebased
      const int* pn = (const int*)ln->h_pn.p;
      size_t rows = 0;
      for (int f = 0; f < nf; f++) rows = std::max(rows, (size_t)pn[nf + f] + (size_t)std::max(0, pn[f]));
      const size_t o0 = dets->p_sc.size();
      dets->p_bb.resize((o0 + rows) * 3); dets->p_sc.resize(o0 + rows); dets->p_sh.resize((o0 + rows) * dim);
      if (rows) {
        std::memcpy(&dets->p_bb[o0 * 3], ln->h_pbb.p, rows * 3 * sizeof(int));
        std::memcpy(&dets->p_sc[o0], ln->h_psc.p, rows * sizeof(Real));
        std::memcpy(&dets->p_sh[o0 * dim], ln->h_psh.p, rows * dim * sizeof(Real));
      }
      for (int f = 0; f < nf; f++) { dets->p_n[(size_t)f0 + f] = pn[f]; dets->p_first[(size_t)f0 + f] = (int)o0 + pn[nf + f]; }
    } else if (n_out && dets) {
      if (results_pending) JDA_HIP(hipStreamSynchronize(st));
      results_pending = false;
      if (kn().debug_times) fprintf(stderr, "[jda] lane %d: results D2H wait %.3f ms (%zu detections)\n", lane, now_ms() - t_dbg, n_out);
      const uint32_t* g = (const uint32_t*)ln->h_gid.p;
      const Real* sc = (const Real*)ln->h_score.p;
      const Real* sh = (const Real*)ln->h_shape.p;
      // back into scan order: sort (gid, arrival index) packed in one word -- gids are unique
      std::vector<unsigned long long> key(n_out);
      for (size_t i = 0; i < n_out; i++) key[i] = ((unsigned long long)g[i] << 32) | (unsigned long long)i;
      std::sort(key.begin(), key.end());
      const size_t o0 = dets->gid.size();
      dets->gid.resize(o0 + n_out); dets->score.resize(o0 + n_out); dets->shape.resize((o0 + n_out) * dim);
      const uint32_t gid_off = (uint32_t)((size_t)f0 * wpf);
      for (size_t i = 0; i < n_out; i++) {
        const uint32_t j = (uint32_t)(key[i] & 0xffffffffu);
        dets->gid[o0 + i] = g[j] + gid_off;
        dets->score[o0 + i] = sc[j];
        std::memcpy(&dets->shape[(o0 + i) * dim], &sh[(size_t)j * dim], dim * sizeof(Real));
      }
      if (kn().debug_times) fprintf(stderr, "[jda] lane %d: collect total %.3f ms\n", lane, now_ms() - t_dbg);
    }
    if (want_trace()) {
      JDA_HIP(hipStreamSynchronize(st));
      const size_t nw = (size_t)windows(), o = (size_t)f0 * wpf;
      if (trace->carts_n) JDA_HIP(hipMemcpy(trace->carts_n + o, w.tr_carts, nw * 4, hipMemcpyDeviceToHost));
      if (trace->score) JDA_HIP(hipMemcpy(trace->score + o, w.tr_score, nw * sizeof(Real), hipMemcpyDeviceToHost));
      if (trace->path_hash) JDA_HIP(hipMemcpy(trace->path_hash + o, w.tr_hash, nw * 4, hipMemcpyDeviceToHost));
      if (trace->shapes) JDA_HIP(hipMemcpy(trace->shapes + o * dim, w.tr_shape, nw * dim * sizeof(Real), hipMemcpyDeviceToHost));
    }
    return true;
  }
};

// The member functions of Pass are compiled ONCE, in pass.cpp (explicit instantiation); the other translation units that
// include this header only see the declarations' bodies, they do not generate them again.
extern template struct Pass<float>;
extern template struct Pass<double>;

struct PendingBatch {
  bool active = false;       // submitted, not yet collected
  bool reserved = false;     // a submit is filling this slot
  bool waiting = false;      // a Wait is collecting it
  Lane* lane = nullptr;      // held (busy) from Submit to the end of Wait
  Pass<float> pass;
  RawDets<float> dets;
  RunStats rs;
  PlanEntry* pe = nullptr;   // pinned from Submit to the end of Wait
  ScanPlan sp;
  int n = 0;
  bool opt_set = false;
  jdaDetectOptions opt{};
  double t_submit = 0;
  // host-frame submits: the H2D copy (blocking for pageable memory) and the scan launches run on a helper thread,
  // so that the submitting thread is free to collect the other ticket meanwhile
  std::thread issuer;
  std::vector<const unsigned char*> host_ptrs;   // the caller's frame pointers, copied at Submit (only the frame BYTES must stay valid until Wait)
  bool issue_ok = true;
  std::string issue_err;
  void join_issuer() { if (issuer.joinable()) issuer.join(); }
  void reset() {             // (keeps `reserved`; the issuer has been joined)
    lane = nullptr; pass = Pass<float>(); dets = RawDets<float>(); rs = RunStats(); pe = nullptr; sp = ScanPlan();
    n = 0; opt_set = false; opt = jdaDetectOptions{}; t_submit = 0; issue_ok = true; issue_err.clear();
  }
};

}  // namespace jda
