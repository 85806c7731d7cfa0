// libjda.so, host side: the device of a cascador, its lanes (stream + workspace + staging buffers) and their workspaces.
#include "host.h"

namespace jda {

// ---------------------------------------------------------------- streams on hardware queues (StreamPool, host.h)

namespace {
// spin lengths in ticks of the 100-MHz wall clock: 200 us for the first known queue, 50 us more for every further one, so
// that the spins end in a known order, well apart, and well after the stamp of a stream that waits for none of them
constexpr long long kSpinBase = 20000, kSpinStep = 5000;
}

// The known queue `s` shares -- its packets run behind that queue's -- or -1: none of them (-3: a HIP call failed).
int StreamPool::classify(hipStream_t s) {
  const int m = (int)rep.size();
  if (m == 0) return -1;
  if (!stamps) {
    if (hipHostMalloc((void**)&stamps, sizeof(unsigned long long) * kSlots * kRow, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); stamps = nullptr; return -3; }
    std::memset(stamps, 0, sizeof(unsigned long long) * kSlots * kRow);
  }
  // (a row per probe, in turn: a spin that sits behind a busy lane's work may write long after its probe has been read)
  volatile unsigned long long* h = stamps + (size_t)(probe_no++ % kSlots) * kRow;
  for (int k = 0; k <= m; k++) h[k] = 0;
  for (int k = 0; k < m; k++)
    if (launch_hwq_spin(kSpinBase + k * kSpinStep, (unsigned long long*)(h + k), rep[k]) != hipSuccess) return -3;
  if (launch_hwq_stamp((unsigned long long*)(h + m), s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -3;
  probes++;
  // Only the new stream is waited for.  A spin that has not ended by now (its stamp still zero) cannot be in front of
  // the new stream's kernel; of those that have, the one that ended last before it ran is the queue it waited in.
  const unsigned long long t = h[m];
  int best = -1;
  unsigned long long best_d = ~0ull;
  for (int k = 0; k < m; k++) {
    const unsigned long long e = h[k];
    if (e != 0 && e <= t && t - e < best_d) { best = k; best_d = t - e; }
  }
  return best;
}

// One more stream, probed and filed (unused) under its queue; a queue not seen before becomes a class of its own.
bool StreamPool::create_one(int* cls_out) {
  hipStream_t s = nullptr;
  JDA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int cls = classify(s);
  if (cls == -3) { (void)hipGetLastError(); cls = -1; dry = 4; }      // (no probing on this device: the runtime's deal)
  else if (cls == -1 && (int)rep.size() < kMaxCls) {
    rep.push_back(s); mains.push_back(0); others.push_back(0);
    cls = (int)rep.size() - 1;
    dry = 0;
  } else if (cls >= 0) dry++;
  items.push_back(Item{s, cls, false, kAux});
  created++;
  *cls_out = cls;
  return true;
}

// The known queue a stream of this role should go to: fewest main streams, then fewest other streams (side, upload,
// table builds), then one the pool holds an unused stream of.
int StreamPool::best_class(Role, int avoid) const {
  // (side / upload streams by the TOTAL number of streams on a queue instead of main streams first: the same rates,
  // profiles/r06_hwq.txt)
  int best = -1, best_free = 0;
  for (int q = 0; q < (int)rep.size(); q++) {
    if (q == avoid) continue;
    int free_q = 0;
    for (const Item& it : items) if (!it.used && it.cls == q) { free_q = 1; break; }
    if (best < 0 || mains[q] < mains[best] || (mains[q] == mains[best] && (others[q] < others[best] || (others[q] == others[best] && free_q > best_free)))) { best = q; best_free = free_q; }
  }
  return best;
}

hipStream_t StreamPool::take(Role role, int avoid, int* cls_out) {
  std::lock_guard<std::mutex> lk(mu);
  auto hand_out = [&](Item& it) {
    it.used = true; it.role = role;
    if (it.cls >= 0) { if (role == kMain) mains[it.cls]++; else others[it.cls]++; }
    if (cls_out) *cls_out = it.cls;
    return it.s;
  };
  if (!place) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { fail(std::string("hipStreamCreateWithFlags failed: ") + hipGetErrorString(hipGetLastError())); return nullptr; }
    items.push_back(Item{s, kNone, false, role}); created++;
    return hand_out(items.back());
  }
  for (int attempt = 0; attempt < 8; attempt++) {
    const int b = best_class(role, avoid);
    const bool all_known = dry >= 4 || (int)rep.size() >= std::min(kMaxCls, hw_queues);
    if (b < 0 && all_known && !rep.empty()) break;      // (every known queue is the one to keep away from -- a runtime with ONE hardware queue: whatever is at hand)
    if (b >= 0 && (all_known || (mains[b] == 0 && others[b] == 0)))
      for (Item& it : items) if (!it.used && it.cls == b) return hand_out(it);
    // nothing of that queue at hand (or a queue nobody uses may still be out there): one more stream, wherever it lands
    int cls = kNone;
    if (!create_one(&cls)) return nullptr;
    if (cls == -1) return hand_out(items.back());          // a queue of its own
  }
  // the runtime would not give us the queue we wanted: the least loaded of what is at hand
  Item* pick = nullptr;
  auto load = [&](const Item& it) { return it.cls < 0 ? 0 : (it.cls == avoid ? 1000 : 0) + 10 * mains[it.cls] + others[it.cls]; };
  for (Item& it : items) if (!it.used && (!pick || load(it) < load(*pick))) pick = &it;
  if (pick) return hand_out(*pick);
  int cls = kNone;
  if (!create_one(&cls)) return nullptr;
  return hand_out(items.back());
}

void StreamPool::give_back(hipStream_t s) {
  if (!s) return;
  std::lock_guard<std::mutex> lk(mu);
  for (Item& it : items)
    if (it.s == s && it.used) {
      it.used = false;
      if (it.cls >= 0) { if (it.role == kMain) mains[it.cls]--; else others[it.cls]--; }
    }
}

void StreamPool::destroy() {
  std::lock_guard<std::mutex> lk(mu);
  for (Item& it : items) { (void)hipStreamSynchronize(it.s); (void)hipStreamDestroy(it.s); }
  items.clear(); rep.clear(); mains.clear(); others.clear();
  if (stamps) (void)hipHostFree(stamps);
  stamps = nullptr;
}

// ---------------------------------------------------------------- device init, lanes

// Makes the cascador's device current for the calling thread; first use picks the device (caller holds c->mu then).
bool ensure_device(Cascador* c) {
  if (c->dev_init) {
    JDA_HIP(hipSetDevice(c->device));
    return true;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    fail("no usable HIP device (hipGetDeviceCount: " + std::string(hipGetErrorString(e)) +
         "); libjda has no CPU fallback for the cascade");
    return false;
  }
  if (c->device < 0) {
    int cur = 0;
    JDA_HIP(hipGetDevice(&cur));
    c->device = cur;
  }
  if (c->device >= n) { fail("device ordinal out of range"); return false; }
  JDA_HIP(hipSetDevice(c->device));
  { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && v > 0) c->n_cus = v; }
  c->streams.place = c->kn.hwq_place != 0;
  c->streams.hw_queues = (int)std::max<long long>(1, env_ll("GPU_MAX_HW_QUEUES", 4));   // (the runtime's own setting, default four)
  if (!(c->aux = c->streams.take(StreamPool::kAux, StreamPool::kNone, nullptr))) return false;
  c->dev_init = true;
  return true;
}

// A free lane (caller holds c->mu): the one whose workspace fits `want_cap` windows most tightly, else the largest,
// else a new one -- unless the pool has reached max_lanes: then nullptr with *exhausted set (the caller waits for a
// lane to come back, or goes on with the lanes it holds).  Free lanes that were passed over `lane_idle_calls` times
// give their buffers back: into `trimmed`, which the caller lets go out of scope AFTER it has released c->mu (the
// frees synchronise the device; no device work runs under the mutex).
Lane* acquire_lane_locked(Cascador* c, size_t want_cap, bool* exhausted, Lane::Bag* trimmed) {
  Lane* best = nullptr;
  for (auto& up : c->lanes) {
    Lane* l = up.get();
    if (l->busy) continue;
    if (!best) { best = l; continue; }
    const bool fit = l->cap >= want_cap, bfit = best->cap >= want_cap;
    if (fit != bfit ? fit : (fit ? l->cap < best->cap : l->cap > best->cap)) best = l;
  }
  if (!best) {
    if ((long long)c->lanes.size() >= std::max<long long>(1, c->kn.max_lanes)) { if (exhausted) *exhausted = true; return nullptr; }
    std::unique_ptr<Lane> l(new (std::nothrow) Lane());
    if (!l || !l->create(&c->streams)) { if (l) l->destroy(); return nullptr; }
    best = l.get();
    c->lanes.push_back(std::move(l));
  }
  const long long idle_max = c->kn.lane_idle_calls;
  for (auto& up : c->lanes) {
    Lane* l = up.get();
    if (l->busy || l == best) continue;
    if (idle_max > 0 && ++l->idle > (unsigned long long)idle_max && (l->ws.p || l->frames.p || l->rag_frames.p)) l->trim(trimmed);
  }
  best->busy = true;
  best->idle = 0;
  return best;
}

// ---------------------------------------------------------------- workspace

// Bytes of the per-window arrays of a pass sized for the WORST case: every window in every queue (what a pass costs when
// its queues cannot be bounded: traces, dense mode, plans without a prediction under ws_bound = 0).
template <typename Real>
size_t bytes_per_window(int dim, bool trace) {
  size_t b = (4 + sizeof(Real) + 4 + 8) + 2 * (4 + sizeof(Real) + (size_t)dim * sizeof(Real)) + 8 + 4;
  if (trace) b += 4 + 4 + sizeof(Real) + 4 + (size_t)dim * sizeof(Real);
  return b;
}

namespace {
// The arrays of a workspace for `cap` windows with cap_q hand-off entries and cap_m mid-queue / detection entries, carved
// out of one allocation (base == nullptr: only the size is computed).
template <typename Real>
size_t carve_workspace(void* base, WorkT<Real>& w, size_t cap, size_t cap_q, size_t cap_m, bool trace, bool dense, int dim) {
  Carver cv(base);
  w.q_gid = cv.take<uint32_t>(cap_q);
  w.q_score = cv.take<Real>(cap_q);
  w.q_kstart = cv.take<uint32_t>(cap_q);
  w.q_xy = cv.take<uint32_t>(cap_q);
  w.q_wf = cv.take<uint32_t>(cap_q);
  w.q_hash = trace ? cv.take<uint32_t>(cap_q) : nullptr;
  w.m_gid = cv.take<uint32_t>(cap_m);
  w.m_score = cv.take<Real>(cap_m);
  w.m_hash = trace ? cv.take<uint32_t>(cap_m) : nullptr;
  w.m_shape = cv.take<Real>(cap_m * dim);
  w.m_xy = cv.take<uint32_t>(cap_m);
  w.m_wf = cv.take<uint32_t>(cap_m);
  w.st_carts = dense ? cv.take<int>(cap) : nullptr;          // (dense mode only: k_stage)
  w.out_gid = cv.take<uint32_t>(cap_m);
  w.out_score = cv.take<Real>(cap_m);
  w.out_shape = cv.take<Real>(cap_m * dim);
  w.counters = cv.take<unsigned long long>(kCntShards * kCntStride);
#ifdef JDA_SCAN_TIMING
  w.dbg = cv.take<unsigned long long>(65536 * 32);
#endif
  if (trace) {
    w.tr_carts = cv.take<int>(cap); w.tr_score = cv.take<Real>(cap);
    w.tr_hash = cv.take<uint32_t>(cap); w.tr_shape = cv.take<Real>(cap * dim);
  } else {
    w.tr_carts = nullptr; w.tr_score = nullptr; w.tr_hash = nullptr; w.tr_shape = nullptr;
  }
  return cv.off + 256;
}
}  // namespace

template <typename Real>
size_t workspace_bytes(size_t cap, size_t cap_q, size_t cap_m, bool trace, bool dense, int dim) {
  WorkT<Real> w{};
  if (dense) cap_m = std::max(cap_m, cap);
  return carve_workspace<Real>(nullptr, w, cap, std::min(cap_q, cap), std::min(cap_m, cap), trace, dense, dim);
}

// The lane's arrays for a pass over `cap` windows of dialect Real whose hand-off queue holds cap_q and whose mid queue
// and detection list hold cap_m entries (0: cap, the worst case); dense: the per-window state of k_stage too (then
// cap_m >= cap).  Grow-only per size; the lane is idle: its holder has collected whatever ran on it.
template <typename Real>
bool ensure_workspace(Lane* ln, size_t cap, bool trace, int dim, size_t cap_q, size_t cap_m, bool dense) {
  if (cap_q == 0 || cap_q > cap) cap_q = cap;
  if (cap_m == 0 || cap_m > cap) cap_m = cap;
  if (dense) cap_m = cap;
  const bool same = ln->dim == dim && ln->real_bytes == (int)sizeof(Real);
  if (same && ln->cap >= cap && ln->cap_q >= cap_q && ln->cap_m >= cap_m && (ln->trace || !trace) && (ln->dense_ws || !dense)) return true;
  if (same) {
    trace = trace || ln->trace; dense = dense || ln->dense_ws;
    cap = std::max(cap, ln->cap); cap_q = std::max(cap_q, ln->cap_q); cap_m = std::max(cap_m, ln->cap_m);
  }
  if (dense) cap_m = std::max(cap_m, cap);
  WorkT<Real>& w = Sel<Real>::work(ln);
  if (ln->stream) (void)hipStreamSynchronize(ln->stream);      // nothing may still use the old carving
  w = WorkT<Real>{};
  const size_t bytes = carve_workspace<Real>(nullptr, w, cap, cap_q, cap_m, trace, dense, dim);
  if (!ln->ws.reserve(bytes)) {
    // the old allocation is gone: forget every pointer carved out of it
    ln->cap = 0; ln->cap_q = 0; ln->cap_m = 0; ln->trace = false; ln->dense_ws = false; ln->real_bytes = 0;
    ln->wf = WorkT<float>{}; ln->wd = WorkT<double>{};
    return false;
  }
  (void)carve_workspace<Real>(ln->ws.p, w, cap, cap_q, cap_m, trace, dense, dim);
  w.cap = (unsigned)cap; w.cap_q = (unsigned)cap_q; w.cap_m = (unsigned)cap_m;
  ln->cap = cap; ln->cap_q = cap_q; ln->cap_m = cap_m; ln->trace = trace; ln->dense_ws = dense; ln->dim = dim; ln->real_bytes = (int)sizeof(Real);
  return true;
}

// Copies a run of host frames to the staging buffer (frame i at dst + i*stride) on a stream.
// Frames that lie back to back in host memory (one array) go as ONE strided copy: 256 separate
// 300-KB copies from pageable memory reach ~15 GB/s, one copy of the batch 57 GB/s (tools/pcie_bw.py).
bool copy_frames_h2d(uint8_t* dst, size_t stride, const unsigned char* const* frames, int n, size_t fbytes,
                            hipStream_t st) {
  for (int i = 0; i < n;) {
    int j = i + 1;
    while (j < n && frames[j] == frames[j - 1] + fbytes) j++;
    if (j - i == 1) {
      JDA_HIP(hipMemcpyAsync(dst + (size_t)i * stride, frames[i], fbytes, hipMemcpyHostToDevice, st));
    } else if (stride == fbytes) {
      JDA_HIP(hipMemcpyAsync(dst + (size_t)i * stride, frames[i], fbytes * (size_t)(j - i), hipMemcpyHostToDevice, st));
    } else {
      JDA_HIP(hipMemcpy2DAsync(dst + (size_t)i * stride, stride, frames[i], fbytes, fbytes, (size_t)(j - i),
                               hipMemcpyHostToDevice, st));
    }
    i = j;
  }
  return true;
}

template size_t bytes_per_window<float>(int, bool);
template size_t bytes_per_window<double>(int, bool);
template bool ensure_workspace<float>(Lane*, size_t, bool, int, size_t, size_t, bool);
template bool ensure_workspace<double>(Lane*, size_t, bool, int, size_t, size_t, bool);
template size_t workspace_bytes<float>(size_t, size_t, size_t, bool, bool, int);
template size_t workspace_bytes<double>(size_t, size_t, size_t, bool, bool, int);

}  // namespace jda
