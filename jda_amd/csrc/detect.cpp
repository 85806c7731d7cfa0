// libjda.so: the C ABI of include/jda.h on top of the HIP kernels.
//
// Boundary of the device work: the host enumerates pyramid
// levels and does NMS + relocation; the device does resize, cascade walk,
// stage regression and compaction.  There is no CPU fallback for the cascade:
// without a usable HIP device every detect entry fails loudly.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/jda.h"
#include "kernels.h"
#include "model.h"
#include "plan.h"
#include "post.h"

namespace jda {

// ---------------------------------------------------------------- error channel

static thread_local std::string g_err;

static void fail(const std::string& msg) {
  g_err = msg;
  std::fprintf(stderr, "libjda: %s\n", msg.c_str());
}

#define JDA_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      fail(std::string(#expr) + " failed: " + hipGetErrorString(e_));                   \
      return false;                                                                     \
    }                                                                                   \
  } while (0)

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

static long long env_ll(const char* name, long long dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoll(v) : dflt;
}

// ---------------------------------------------------------------- knobs
// Tuning values of a cascador.  Read ONCE, when the cascador is created, from the JDA_* environment variables of
// DESIGN.md section 8 (experiments set them before jdaCascadorCreate*); jdaSetOption changes the documented ones
// afterwards.  Nothing on the call path touches the environment.
#define JDA_KNOBS(X)                                                                                   \
  X(handoff, "JDA_HANDOFF", 128)            /* carts of stage 0 k_scan evaluates before k_finish takes over */ \
  X(first_phase, "JDA_FIRST_PHASE", 16)     /* carts before k_scan's first compaction */                \
  X(cp_max, "JDA_CP_MAX", 128)              /* windows per tile at or below which a phase spreads (window, cart) pairs */ \
  X(lds_win_max, "JDA_LDS_WIN_MAX", 100)    /* largest window that gets an LDS pixel tile */            \
  X(tile_cglb, "JDA_TILE_CGLB", 800)        /* cost per window of the global-pixel mode (tile chooser) */ \
  X(glb_tile_fit, "JDA_GLB_TILE_FIT", 1)                                                               \
  X(no_global_scan, "JDA_NO_GLOBAL_SCAN", 0)                                                           \
  X(no_lds_scan, "JDA_NO_LDS_SCAN", 0)                                                                 \
  X(no_fast_scan, "JDA_NO_FAST_SCAN", 0)                                                               \
  X(debug_tiles, "JDA_DEBUG_TILES", 0)                                                                 \
  X(plan_cache, "JDA_PLAN_CACHE", 64)       /* scan plans kept per cascador */                          \
  X(fin_s0, "JDA_FIN_S0", 1)                                                                           \
  X(dense, "JDA_DENSE", 1)                  /* 0 off, 1 auto, 2 always */                              \
  X(dense_lds_max, "JDA_DENSE_LDS_MAX", 160 * 1024)                                                    \
  X(dense_pix, "JDA_DENSE_PIX", 16 * 1024)                                                             \
  X(dense_pct, "JDA_DENSE_PCT", 50)                                                                    \
  X(merge_blocks, "JDA_MERGE_BLOCKS", 2048) /* workgroups below which the LDS-tiled levels share one launch */ \
  X(side_small, "JDA_SIDE_SMALL", 1)                                                                   \
  X(side_stream, "JDA_SIDE_STREAM", 1)                                                                 \
  X(side_after, "JDA_SIDE_AFTER", 0)        /* ... forked after this many LDS-tiled launches have been queued (the persistent scan takes its CUs first, the global-pixel workgroups fill what it leaves) */ \
  X(lanes_reverse, "JDA_LANES_REVERSE", 1)                                                             \
  X(finish_merge, "JDA_FINISH_MERGE", 4096) /* hand-off count below which one k_finish launch does all stages */ \
  X(wide_max, "JDA_WIDE_MAX", 1024)         /* ... and below which a window gets a whole workgroup (k_finish_wide) */ \
  X(wide_busy_max, "JDA_WIDE_BUSY_MAX", 2)  /* ... unless more than this many lanes of the cascador are in use */ \
  X(h2d_stream, "JDA_H2D_STREAM", 1)        /* host frames go up on ONE stream per cascador, batch after batch, not lane by lane */ \
  X(h2d_min_bytes, "JDA_H2D_MIN_BYTES", 8 << 20) /* ... for uploads of at least this many bytes */ \
  X(ragged_uploader, "JDA_RAGGED_UPLOADER", 1) /* ragged job from one packed host buffer: a helper thread uploads chunk after chunk */ \
  X(ragged_stage_threads, "JDA_RAGGED_STAGE_THREADS", 4) /* ... and this many threads gather separate host arrays into its pinned buffers */ \
  X(kernel_d2h, "JDA_KERNEL_D2H", 1)        /* counters and detections -> pinned host memory by a kernel, not the copy engine */ \
  X(filter0, "JDA_FILTER0", 1)              /* large hand-off queues: k_filter0 + k_finish(survivors) instead of two k_finish passes */ \
  X(fin_gm, "JDA_FIN_GM", 0)                /* speculative 64-cart groups per k_finish round (0: from K) */ \
  X(fin_g1, "JDA_FIN_G1", 1)                                                                           \
  X(fin_g2, "JDA_FIN_G2", 0)                                                                           \
  X(fin_tile, "JDA_FIN_TILE", -1)           /* k_finish LDS window tile: -1 auto, 0 off, n pixels */    \
  X(fin_tile1, "JDA_FIN_TILE1", 0)                                                                     \
  X(fin_grid_div, "JDA_FIN_GRID_DIV", 4)                                                               \
  X(predict, "JDA_PREDICT", 1)              /* size the finishing launches from the previous pass (no host round trip) */ \
  X(debug_times, "JDA_DEBUG_TIMES", 0)                                                                 \
  X(test_wpf_scale, "JDA_TEST_WPF_SCALE", 1) /* test hook of the 32-bit window-id guard */             \
  X(lanes, "JDA_LANES", 2)                  /* sub-batch lanes of one synchronous call */               \
  X(lanes_min_windows, "JDA_LANES_MIN_WINDOWS", 2000000)                                               \
  X(host_chunk, "JDA_HOST_CHUNK", 128)      /* frames per sub-batch when the frames come from host memory */ \
  X(workspace_mb, "JDA_WORKSPACE_MB", 24 * 1024)                                                       \
  X(host_submit_thread, "JDA_HOST_SUBMIT_THREAD", 1)                                                   \
  X(ragged_chunk_windows, "JDA_RAGGED_CHUNK_WINDOWS", 6000000) /* windows per chunk of a ragged batch */ \
  X(ragged_tile_grow_pct, "JDA_RAGGED_TILE_GROW_PCT", 150)      /* pixel bytes of a re-cut tile, % of the level's nominal tile */ \
  X(max_lanes, "JDA_MAX_LANES", 16)         /* lanes (stream + workspace + staging) a cascador creates at most; further concurrent callers wait for one */ \
  X(lane_idle_calls, "JDA_LANE_IDLE_CALLS", 256) /* lane hand-outs a free lane sits out before its workspace and staging buffers are released (0: never) */ \
  X(w_pad, "JDA_W_PAD", 1)                  /* k_finish gathers its weight rows from a copy whose rows start on 128-byte lines (0: from the tight table) */ \
  X(lm_deep, "JDA_LM_DEEP", 1)              /* trees of five or more node levels: k_finish reads the levels from the fourth on as whole records grouped per path (0: every level from the level-major split copy) */ \
  X(w_stream_mb, "JDA_W_STREAM_MB", 8)      /* ... with non-temporal loads when one stage's rows exceed this many MB (they would only push the stage's nodes out of L2); 0: never */ \
  X(scan_lean, "JDA_SCAN_LEAN", 1)          /* scan kernels without the per-cart test of the normalisation flag where no cart of the scanned range normalises */ \
  X(scan_p, "JDA_SCAN_P", 1)                /* persistent scan kernel (k_scan_p): 0 off, 1 for the levels of large uniform batches it suits, 2 whenever it fits */ \
  X(scan_p_block, "JDA_SCAN_P_BLOCK", 768)  /* ... threads per workgroup */                             \
  X(scan_p_min_slots, "JDA_SCAN_P_MIN_SLOTS", 4) /* ... pixel-tile slots a level's workgroup must have room for (scan_p = 1) */ \
  X(scan_p_wgs, "JDA_SCAN_P_WGS", 1)        /* ... workgroups per CU */                                 \
  X(scan_p_slots, "JDA_SCAN_P_SLOTS", 5)    /* ... pixel-tile slots per workgroup at most while other batches are in flight on the cascador (0: as many as fit, at most 8).  Five, not the six the 46-pixel level has room for: the 28 KB left per CU let workgroups of the other batch run next to it (submit/wait step 1.495 -> 1.45 ms) */ \
  X(scan_p_b0, "JDA_SCAN_P_B0", 32)         /* ... cart counts at which windows are re-bucketed */       \
  X(scan_p_b1, "JDA_SCAN_P_B1", 64)                                                                    \
  X(scan_p_b2, "JDA_SCAN_P_B2", 0)                                                                     \
  X(scan_p_b3, "JDA_SCAN_P_B3", 0)                                                                     \
  X(scan_p_b4, "JDA_SCAN_P_B4", 0)                                                                     \
  X(scan_p_handoff, "JDA_SCAN_P_HANDOFF", 0) /* ... carts of stage 0 it evaluates (0: `handoff`).  Its cart tables are loaded once per workgroup and its deep windows pooled over all tiles, so a later hand-off costs it little */ \
  X(scan_p_ring, "JDA_SCAN_P_RING", 256)    /* ... items per ring (rounded up to a power of two) */     \
  X(scan_p_lg, "JDA_SCAN_P_LG", 64)         /* ... task form per bucket, one decimal digit each: 6 lane = window, 5 / 4 / 7 / 8 pair tasks of 32 / 16 / 8 / 4 windows, 9 a pair task of 1 to 4 windows taken as soon as one waits */ \
  X(scan_p_opts, "JDA_SCAN_P_OPTS", 0)      /* ... bit 0 / 1: 8 trees in flight per lane in fresh / bucket tasks */ \
  X(scan_p_tile_kb, "JDA_SCAN_P_TILE_KB", 0) /* ... its own cut of a level's tile in y: as many rows of windows as keep the pixel tile within this many KB (0: the plan's tile) */ \
  X(scan_p_lds_kb, "JDA_SCAN_P_LDS_KB", 160) /* ... LDS a workgroup may take: what it leaves of the CU's 160 KB is where the other batch's kernels (global-pixel scan: 23.1 KB per workgroup, k_finish: 7.5 KB) find room next to it */ \
  X(scan_p_win_max, "JDA_SCAN_P_WIN_MAX", 100000) /* ... largest window of a level it takes */ \
  X(scan_p_dyn, "JDA_SCAN_P_DYN", 1)        /* ... tiles dealt to the workgroups at run time (a workgroup that starts late takes fewer) instead of in fixed shares */ \
  X(scan_p_grid, "JDA_SCAN_P_GRID", 0)      /* ... workgroups of a launch (0: one per CU x scan_p_wgs) */ \
  X(scan_p_mid, "JDA_SCAN_P_MID", 1)        /* ... with scan_p_handoff >= K: windows that pass stage 0 go straight to the mid queue */

struct Knobs {
#define X(name, env, dflt) long long name = (dflt);
  JDA_KNOBS(X)
#undef X
  void load() {
#define X(name, env, dflt) name = env_ll(env, (dflt));
    JDA_KNOBS(X)
#undef X
  }
  // Values no code path can work with are refused (jdaSetOption returns -1): negative sizes and counts; the rest of
  // a knob's range is clamped where it is used.
  bool set(const char* key, long long v) {
    static const char* const non_negative[] = {"workspace_mb", "handoff", "plan_cache", "lanes", "host_chunk", "ragged_chunk_windows",
                                               "h2d_min_bytes", "merge_blocks", "finish_merge", "wide_max", "lanes_min_windows",
                                               "ragged_stage_threads", "scan_p_handoff", "scan_p_slots", "max_lanes", "lane_idle_calls", "scan_p_tile_kb", "scan_p_grid"};
    for (const char* k : non_negative) if (std::strcmp(key, k) == 0 && v < 0) return false;
    if (std::strcmp(key, "workspace_mb") == 0 && v < 1) return false;
#define X(name, env, dflt) if (std::strcmp(key, #name) == 0) { name = v; return true; }
    JDA_KNOBS(X)
#undef X
    return false;
  }
  bool get(const char* key, long long* v) const {
#define X(name, env, dflt) if (std::strcmp(key, #name) == 0) { *v = name; return true; }
    JDA_KNOBS(X)
#undef X
    return false;
  }
};

// Does any of the first K carts of stage 0 normalise its score ((mean, std) != (0, 1), c/jda.c:397), in the precision
// the dialect computes in?  The scan kernels drop the per-cart test of the flag from their loops when none does.
static bool stage0_any_norm(const HostModel& hm, int K, bool fp32) {
  K = std::min(K, hm.K);
  for (int k = 0; k < K; k++) {
    const bool plain = fp32 ? ((float)hm.cart_mean[k] == 0.f && (float)hm.cart_std[k] == 1.f) : (hm.cart_mean[k] == 0.0 && hm.cart_std[k] == 1.0);
    if (!plain) return true;
  }
  return false;
}

// ---------------------------------------------------------------- device buffers

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  // Grow-only.  On failure the buffer is EMPTY (p == nullptr, bytes == 0): callers that carved
  // pointers out of the old allocation must drop them (ensure_workspace does).
  bool reserve(size_t n) {
    if (n <= bytes) return true;
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, n);
    if (e != hipSuccess) {
      (void)hipGetLastError();            // clear the sticky out-of-memory error: a smaller request may follow
      fail("hipMalloc(" + std::to_string(n) + " bytes) failed: " + hipGetErrorString(e));
      return false;
    }
    p = q; bytes = n;
    return true;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// carve typed arrays out of one allocation
struct Carver {
  unsigned char* base; size_t off = 0;
  explicit Carver(void* b) : base((unsigned char*)b) {}
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* r = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return r;
  }
};

template <typename Real>
struct ModelOnDevice {
  DevModelT<Real> m{};
  DevBuf buf;
  bool ready = false;
};

struct PlanKey {
  int w, h, dialect, a, b, c;
  unsigned long long f;
  bool operator<(const PlanKey& o) const {
    return std::tie(w, h, dialect, a, b, c, f) < std::tie(o.w, o.h, o.dialect, o.a, o.b, o.c, o.f);
  }
};

struct PlanEntry {
  ScanPlan sp;
  DevPlan hp{};
  DevPlan* dp = nullptr;
  S0Node* table = nullptr;
  bool fast_scan = false;       // stage 0 has only scale==0 nodes: LDS-tiled scan is valid
  bool lm_ok = false;           // every tiled level's windows fit k_finish's stage-0 table ((x, y) in 11 bits each)
  bool any_untiled = false;
  size_t table_cap = 0;         // S0Node entries the table allocation holds (evicted allocations are recycled)
  bool dense_hint = false;      // the last pass on this plan kept most windows alive: go straight to k_stage
  // Hand-off queue length and detections of earlier passes on this plan, as fractions of the pass's windows (< 0:
  // none yet).  With a prediction the finishing launches are sized and queued right behind the scan, and a prefix
  // of the detection list is copied back speculatively: the whole pass is ONE enqueue and one host wait.  The kernels
  // read the true lengths from the device counters (grid-stride), so a wrong prediction costs time, never results.
  double pred_tail = -1, pred_out = -1;
  double pred_mid = -1;         // ... and the mid queue's (windows that passed stage 0)
  int pins = 0;                 // submitted batches that still use this plan (never evicted while > 0)
  unsigned long long last_use = 0;
};

// k_finish: windows up to this side are copied to LDS before the walks of stages >= 1 (-1: as large as the LDS
// budget of launch_finish allows, 72 pixels for the 27-landmark 540-cart model)
constexpr int kFinishTileWin = -1;
// ... and in the stage-0 launch (JDA_FIN_TILE1): off -- most hand-off windows die within a round or two of carts and
// the copy is one more dependent step in front of them (measured: 1.816 ms of GPU time per step with tiles of 46, 57
// or 72 pixels against 1.806 without)
constexpr int kFinishTileWin1 = 0;   // carts of stage 0 k_scan evaluates before k_finish takes over (JDA_HANDOFF)
// tickets of the submit/wait entries (frames coming over PCIe: a ticket lives for upload 1.4 ms + kernels 1.7 ms + host
// work, so the link and the GPU are only both kept busy with three in flight)
constexpr int kTickets = 3;
constexpr int kRaggedLanes = 3;      // chunks of a ragged job in flight

struct PendingBatch;          // a submitted, not yet collected batch (submit/wait entries), defined after Pass

// Pinned host memory, grow-only (results of a pass land here by asynchronous D2H copies).
struct HostPinned {
  void* p = nullptr;
  size_t bytes = 0;
  // keep: bytes at the front that must survive a reallocation
  bool reserve(size_t n, size_t keep = 0) {
    if (n <= bytes) return true;
    n = std::max<size_t>(n + n / 2, (size_t)1 << 20);
    void* q = nullptr;
    hipError_t e = hipHostMalloc(&q, n, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); fail("hipHostMalloc(" + std::to_string(n) + " bytes) failed: " + hipGetErrorString(e)); return false; }
    if (p && keep) std::memcpy(q, p, std::min(keep, bytes));
    if (p) (void)hipHostFree(p);
    p = q; bytes = n;
    return true;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};

// One caller's share of the device: a stream with its events, a workspace and the staging buffers of a pass.
// A call takes lanes from the cascador's pool for as long as it runs (a big synchronous batch takes two, a ragged job
// up to three, a submitted batch holds one until its Wait) and gives them back; the pool grows with the number of
// concurrent callers.  Nothing in a lane is touched by anybody but its current holder, which is what makes
// jdaDetect re-entrant on ONE cascador (the reference has no globals and no locks, c/jda.c:443-480).
struct Lane {
  bool busy = false;
  unsigned idle = 0;                         // lane hand-outs since this one was last used (free lanes only)
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;                // global-pixel scan launch of a lone lane, next to its LDS-tiled launches
  hipEvent_t ev[5] = {};
  hipEvent_t ev_side[2] = {};
  hipEvent_t ev_user = nullptr;
  hipEvent_t ev_h2d[2] = {};                 // staging buffer free / frames uploaded (Cascador::h2d)
  unsigned long long* h_cnt = nullptr;       // pinned copy of the work counters
  HostPinned h_gid, h_score, h_shape;        // detections of the lane's pass
  DevBuf ws;                                 // per-window arrays, carved for one dialect at a time
  size_t cap = 0; bool trace = false; int dim = 0, real_bytes = 0;
  WorkT<float> wf{};
  WorkT<double> wd{};
  DevBuf frames;                             // staging of host frames (the call's first lane holds the whole batch)
  DevBuf pyr;                                // half + quarter images (multi-scale models), method-0 levels
  // ragged passes: images at the common pitch, tight images, tables (segments, block map, image records)
  DevBuf rag_frames, rag_raw, rag_tab;
  HostPinned h_tab, h_raw;
  bool create() {
    JDA_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (auto& e : ev) JDA_HIP(hipEventCreate(&e));
    JDA_HIP(hipEventCreateWithFlags(&ev_user, hipEventDisableTiming));
    for (auto& e : ev_h2d) JDA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    JDA_HIP(hipHostMalloc((void**)&h_cnt, sizeof(unsigned long long) * kCntShards * kCntStride, hipHostMallocDefault));
    return true;
  }
  bool ensure_side() {
    if (side) return true;
    JDA_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    for (auto& e : ev_side) JDA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return true;
  }
  // The memory of a lane nobody has used for a while (a burst of concurrent callers leaves lanes behind, each with a
  // workspace of up to workspace_mb): everything that is re-created on demand.  The lane is free and its holder has
  // collected what ran on it, so nothing is in flight.
  void trim() {
    ws.release(); frames.release(); pyr.release(); rag_frames.release(); rag_raw.release(); rag_tab.release();
    h_gid.release(); h_score.release(); h_shape.release(); h_tab.release(); h_raw.release();
    cap = 0; trace = false; dim = 0; real_bytes = 0;
    wf = WorkT<float>{}; wd = WorkT<double>{};
  }
  void destroy() {
    if (stream) (void)hipStreamSynchronize(stream);
    if (side) (void)hipStreamSynchronize(side);
    trim();
    if (h_cnt) (void)hipHostFree(h_cnt);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ev_side) if (e) (void)hipEventDestroy(e);
    if (ev_user) (void)hipEventDestroy(ev_user);
    for (auto& e : ev_h2d) if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
    if (side) (void)hipStreamDestroy(side);
  }
};

struct Cascador {
  HostModel hm;
  Knobs kn;
  // Guards the shared parts only -- device/model initialisation, the plan cache, the lane pool, the tickets and the
  // hints below -- for the few microseconds those take; no device work runs under it.
  std::mutex mu;
  // queue lengths of the last pass as fractions of its windows (hand-off queue, detections): a new plan starts from
  // them, see PlanEntry::pred_tail
  double pred_tail = -1, pred_out = -1;
  bool last_dense = false;
  int similarity = 0;          // dialect CPP: Config::with_similarity_transform (reference common.cpp:214)
  int device = -1;
  int n_cus = 256;             // compute units of the device
  bool dev_init = false;
  hipStream_t aux = nullptr;   // stage-0 table builds (under mu)
  // Frame uploads of every lane, in the order they are issued (under h2d_mu).  Uploads issued lane by lane run
  // CONCURRENTLY on the copy engines, each at a fraction of the link: two batches then both arrive late, and their
  // kernels collide afterwards.  One after the other, batch i+1 goes up while batch i computes.
  hipStream_t h2d = nullptr;
  std::mutex h2d_mu;
  std::vector<std::unique_ptr<Lane>> lanes;
  std::condition_variable lane_cv;           // a lane was given back (callers beyond max_lanes wait here, with mu)
  ModelOnDevice<float> mf;
  ModelOnDevice<double> md;
  std::map<PlanKey, PlanEntry> plans;
  struct PlanBuffers { DevPlan* dp; S0Node* table; size_t table_cap; };
  std::vector<PlanBuffers> plan_pool;     // device allocations of evicted plans (hipFree + hipMalloc per miss cost ~0.1 ms)
  unsigned long long plan_clock = 0;
  PendingBatch* pending = nullptr;           // [kTickets], allocated by the first submit
};

template <typename Real> struct Sel;
template <> struct Sel<float> {
  static ModelOnDevice<float>& model(Cascador* c) { return c->mf; }
  static WorkT<float>& work(Lane* l) { return l->wf; }
  static constexpr int dialect = JDA_DIALECT_C;
};
template <> struct Sel<double> {
  static ModelOnDevice<double>& model(Cascador* c) { return c->md; }
  static WorkT<double>& work(Lane* l) { return l->wd; }
  static constexpr int dialect = JDA_DIALECT_CPP;
};

// ---------------------------------------------------------------- device init, lanes

// Makes the cascador's device current for the calling thread; first use picks the device (caller holds c->mu then).
static bool ensure_device(Cascador* c) {
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
  JDA_HIP(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
  c->dev_init = true;
  return true;
}

// A free lane (caller holds c->mu): the one whose workspace fits `want_cap` windows most tightly, else the largest,
// else a new one -- unless the pool has reached max_lanes: then nullptr with *exhausted set (the caller waits for a
// lane to come back, or goes on with the lanes it holds).  Free lanes that were passed over `lane_idle_calls` times
// give their buffers back.
static Lane* acquire_lane_locked(Cascador* c, size_t want_cap, bool* exhausted = nullptr) {
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
    if (!l || !l->create()) { if (l) l->destroy(); return nullptr; }
    best = l.get();
    c->lanes.push_back(std::move(l));
  }
  const long long idle_max = c->kn.lane_idle_calls;
  for (auto& up : c->lanes) {
    Lane* l = up.get();
    if (l->busy || l == best) continue;
    if (idle_max > 0 && ++l->idle > (unsigned long long)idle_max && (l->ws.p || l->frames.p || l->rag_frames.p)) l->trim();
  }
  best->busy = true;
  best->idle = 0;
  return best;
}

// The lanes a call holds; given back when it leaves.
struct LaneSet {
  Cascador* c;
  std::vector<Lane*> v;
  explicit LaneSet(Cascador* c_) : c(c_) {}
  LaneSet(const LaneSet&) = delete;
  LaneSet& operator=(const LaneSet&) = delete;
  // Up to n lanes in all.  With the pool at max_lanes and nothing free, a caller that holds no lane yet waits (callers
  // queue up, they do not fail); one that already holds a lane goes on with what it has -- check v.size() -- so that
  // two callers can never wait for each other's lanes.  all = true (a caller that holds none and needs all n, at most
  // max_lanes of them): waits until it can have them all at once.
  bool take(int n, size_t want_cap = 0, bool all = false) {
    std::unique_lock<std::mutex> lk(c->mu);
    const int cap_lanes = (int)std::max<long long>(1, c->kn.max_lanes);
    if (all && v.empty()) {
      n = std::min(n, cap_lanes);
      for (;;) {
        int avail = cap_lanes - (int)c->lanes.size();
        for (auto& up : c->lanes) avail += up->busy ? 0 : 1;
        if (avail >= n) break;
        c->lane_cv.wait(lk);
      }
    }
    while ((int)v.size() < n) {
      bool exhausted = false;
      Lane* l = acquire_lane_locked(c, want_cap, &exhausted);
      if (!l) {
        if (!exhausted) return false;
        if (!v.empty()) return true;
        c->lane_cv.wait(lk);
        continue;
      }
      v.push_back(l);
    }
    return true;
  }
  Lane* detach(size_t i) { Lane* l = v[i]; v.erase(v.begin() + i); return l; }   // the caller keeps it (submitted batch)
  ~LaneSet() {
    if (v.empty()) return;
    { std::lock_guard<std::mutex> lk(c->mu); for (Lane* l : v) l->busy = false; }
    c->lane_cv.notify_all();
  }
};

template <typename Real>
static bool upload_model(Cascador* c) {
  ModelOnDevice<Real>& mo = Sel<Real>::model(c);
  if (mo.ready) return true;
  const HostModel& h = c->hm;
  using Node = typename std::conditional<sizeof(Real) == 4, NodeF, NodeD>::type;
  const size_t carts = (size_t)h.carts();
  const int node_n = h.node_n(), leaf_n = h.leaf_n(), dim = h.dim();
  std::vector<Node> nodes(carts * node_n);
  // stage-0 similarity transform, reference data.cpp:64-114 (see stp_calc in k_finish.hip for the
  // restated OpenCV details); identity when off
  double stp0[5] = {1., 1., 0., 0., 1.};
  if (sizeof(Real) == 8 && c->similarity) {
    const int L = h.L;
    std::vector<double> s1(dim), t1(dim), t2(dim);
    const std::vector<double>& s2 = h.mean_shape;
    const volatile double zero = 0.;
    for (int i = 0; i < dim; i++) s1[i] = s2[i] + zero;
    double x1c = 0., y1c = 0., x2c = 0., y2c = 0.;
    for (int i = 0; i < L; i++) { x1c += s1[2 * i]; y1c += s1[2 * i + 1]; x2c += s2[2 * i]; y2c += s2[2 * i + 1]; }
    x1c /= (double)L; y1c /= (double)L; x2c /= (double)L; y2c /= (double)L;
    for (int i = 0; i < L; i++) {
      t1[2 * i] = s1[2 * i] - x1c; t1[2 * i + 1] = s1[2 * i + 1] - y1c;
      t2[2 * i] = s2[2 * i] - x2c; t2[2 * i + 1] = s2[2 * i + 1] - y2c;
    }
    auto cvnorm = [](const std::vector<double>& v) {
      double a = 0.; size_t i = 0; const size_t n = v.size();
      for (; i + 4 <= n; i += 4) { const double v0 = v[i], v1 = v[i + 1], v2 = v[i + 2], v3 = v[i + 3]; a += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3; }
      for (; i < n; i++) a += v[i] * v[i];
      return std::sqrt(a);
    };
    const double scale1 = cvnorm(t1), scale2 = cvnorm(t2);
    stp0[0] = scale1 / scale2;
    const double a1 = 1. / scale1, a2 = 1. / scale2;
    for (int i = 0; i < dim; i++) { t1[i] = t1[i] * a1 + zero; t2[i] = t2[i] * a2 + zero; }
    double num = 0., den = 0.;
    for (int i = 0; i < L; i++) {
      num += t1[2 * i + 1] * t2[2 * i] - t1[2 * i] * t2[2 * i + 1];
      den += t1[2 * i] * t2[2 * i] + t1[2 * i + 1] * t2[2 * i + 1];
    }
    const double norm = std::sqrt(num * num + den * den);
    const double sn = num / norm, cs = den / norm;
    stp0[1] = cs; stp0[2] = -sn; stp0[3] = sn; stp0[4] = cs;
  }
  for (size_t i = 0; i < nodes.size(); i++) {
    const SplitNode& s = h.nodes[i];
    Node& d = nodes[i];
    d.scale = s.scale; d.lm1x2 = s.lm1 * 2; d.lm2x2 = s.lm2 * 2; d.th = s.th;
    if (sizeof(Real) == 4) {
      // plain narrowing casts, reference c/jda.c:525-532
      d.o1x = (Real)s.off[0]; d.o1y = (Real)s.off[1]; d.o2x = (Real)s.off[2]; d.o2y = (Real)s.off[3];
    } else {
      // STParameter::Apply on each offset pair (data.hpp:42-45, data.cpp:33-34) with the parameter
      // that is the same for every window: the identity when the similarity transform is off, and
      // -- for STAGE 0 only, where every window holds the mean shape -- Calc(mean+0, mean) when it is
      // on.  Later stages keep the raw offsets; k_finish applies each window's own parameter.
      const bool raw = c->similarity && i >= (size_t)h.K * node_n;
      const volatile double sc = stp0[0], r00 = stp0[1], r01 = stp0[2], r10 = stp0[3], r11 = stp0[4];
      if (raw) {
        d.o1x = (Real)s.off[0]; d.o1y = (Real)s.off[1]; d.o2x = (Real)s.off[2]; d.o2y = (Real)s.off[3];
      } else {
        d.o1x = (Real)(sc * (r00 * s.off[0] + r01 * s.off[1]));
        d.o1y = (Real)(sc * (r10 * s.off[0] + r11 * s.off[1]));
        d.o2x = (Real)(sc * (r00 * s.off[2] + r01 * s.off[3]));
        d.o2y = (Real)(sc * (r10 * s.off[2] + r11 * s.off[3]));
      }
    }
  }
  auto cast = [](const std::vector<double>& v) {
    std::vector<Real> o(v.size());
    for (size_t i = 0; i < v.size(); i++) o[i] = (Real)v[i];
    return o;
  };
  std::vector<Real> leaf = cast(h.leaf_score), cth = cast(h.cart_th), cmean = cast(h.cart_mean),
                    cstd = cast(h.cart_std), w = cast(h.w), ms = cast(h.mean_shape), ms_raw = cast(h.mean_shape);
  if (sizeof(Real) == 8) {
    const volatile double zero = 0.;
    for (auto& v : ms) v = (Real)((double)v + zero);   // RandomShape with zero shift, data.cpp:225-236
  }
  std::vector<uint8_t> cnorm(carts);
  for (size_t i = 0; i < carts; i++) cnorm[i] = !(cmean[i] == (Real)0 && cstd[i] == (Real)1);
  std::vector<Real> par0(carts * 4);             // {th, norm, mean, std} per cart (CartPar), packed for LDS staging
  for (size_t k = 0; k < carts; k++) { par0[4 * k] = cth[k]; par0[4 * k + 1] = cnorm[k] ? (Real)1 : (Real)0; par0[4 * k + 2] = cmean[k]; par0[4 * k + 3] = cstd[k]; }

  // level-major split copy of the nodes for k_finish (kernels.h: NodeOff, lm_index)
  std::vector<NodeOff<Real>> lm_off(nodes.size());
  std::vector<uint2> lm_meta(nodes.size());
  for (size_t t = 0; t < (size_t)h.T; t++)
    for (unsigned k = 0; k < (unsigned)h.K; k++)
      for (unsigned d = 0, n = 0; n < (unsigned)node_n; n++) {
        while (n >= (2u << d) - 1u) d++;
        const Node& s = nodes[(t * h.K + k) * node_n + n];
        const size_t o = t * (size_t)h.K * node_n + lm_index((unsigned)h.K, k, d, n);
        lm_off[o].o1x = s.o1x; lm_off[o].o1y = s.o1y; lm_off[o].o2x = s.o2x; lm_off[o].o2y = s.o2y;
        lm_meta[o].x = (uint32_t)s.lm1x2 | ((uint32_t)s.lm2x2 << 15) | ((uint32_t)s.scale << 30);
        lm_meta[o].y = (uint32_t)s.th;
      }

  // the last levels of deep trees once more, as whole records grouped under their ancestor on level split - 1
  // (kernels.h: lm_deep_index): level-major, each of those levels costs a wave of 64 carts one line per lane and array
  const unsigned levels = (unsigned)h.D - 1u;
  const unsigned split = (c->kn.lm_deep && levels >= 5u) ? 3u : levels;
  const size_t deep_per_cart = (size_t)node_n - ((1u << split) - 1u);
  std::vector<Node> lm_deep(deep_per_cart * (size_t)h.T * h.K);
  if (deep_per_cart)
    for (size_t t = 0; t < (size_t)h.T; t++)
      for (unsigned k = 0; k < (unsigned)h.K; k++)
        for (unsigned d = split; d < levels; d++)
          for (unsigned n = (1u << d) - 1u; n < (2u << d) - 1u; n++)
            lm_deep[(t * h.K) * deep_per_cart + lm_deep_index(k, d, n, levels, split)] = nodes[(t * h.K + k) * node_n + n];
  Carver sz(nullptr);
  sz.take<Node>(lm_deep.size());
  sz.take<NodeOff<Real>>(nodes.size()); sz.take<uint2>(nodes.size());
  sz.take<Node>(nodes.size()); sz.take<Real>(leaf.size()); sz.take<Real>(carts); sz.take<Real>(carts);
  sz.take<Real>(carts); sz.take<uint8_t>(carts); sz.take<Real>(w.size()); sz.take<Real>(dim); sz.take<Real>(dim); sz.take<Real>(par0.size());
  // k_finish's copy of the weight rows: every row on its own 128-byte lines (the file layout, c/jda.c:146, is what
  // k_stage and k_finish_wide stage whole carts of; a wave-per-window gather of single rows pays per line touched)
  const size_t w_rows_n = w.size() / (size_t)dim;
  const int line_elems = 128 / (int)sizeof(Real);
  const int w_pitch = c->kn.w_pad ? ((dim + line_elems - 1) / line_elems) * line_elems : dim;
  const bool padded = w_pitch != dim && w_rows_n * (size_t)w_pitch < (1ull << 32);      // (k_finish keeps row offsets in 32 bits)
  if (padded) sz.take<Real>(w_rows_n * (size_t)w_pitch);
  if (!mo.buf.reserve(sz.off + 256)) return false;
  Carver cv(mo.buf.p);
  Node* d_lm_deep = cv.take<Node>(lm_deep.size());
  if (!lm_deep.empty()) JDA_HIP(hipMemcpy(d_lm_deep, lm_deep.data(), lm_deep.size() * sizeof(Node), hipMemcpyHostToDevice));
  NodeOff<Real>* d_lm_off = cv.take<NodeOff<Real>>(nodes.size());
  uint2* d_lm_meta = cv.take<uint2>(nodes.size());
  JDA_HIP(hipMemcpy(d_lm_off, lm_off.data(), nodes.size() * sizeof(NodeOff<Real>), hipMemcpyHostToDevice));
  JDA_HIP(hipMemcpy(d_lm_meta, lm_meta.data(), nodes.size() * sizeof(uint2), hipMemcpyHostToDevice));
  Node* d_nodes = cv.take<Node>(nodes.size());
  Real* d_leaf = cv.take<Real>(leaf.size());
  Real* d_cth = cv.take<Real>(carts);
  Real* d_cmean = cv.take<Real>(carts);
  Real* d_cstd = cv.take<Real>(carts);
  uint8_t* d_cnorm = cv.take<uint8_t>(carts);
  Real* d_w = cv.take<Real>(w.size());
  Real* d_ms = cv.take<Real>(dim);
  Real* d_ms_raw = cv.take<Real>(dim);
  Real* d_par0 = cv.take<Real>(par0.size());
  Real* d_w_rows = padded ? cv.take<Real>(w_rows_n * (size_t)w_pitch) : d_w;
  JDA_HIP(hipMemcpy(d_nodes, nodes.data(), nodes.size() * sizeof(Node), hipMemcpyHostToDevice));
  JDA_HIP(hipMemcpy(d_leaf, leaf.data(), leaf.size() * sizeof(Real), hipMemcpyHostToDevice));
  JDA_HIP(hipMemcpy(d_cth, cth.data(), carts * sizeof(Real), hipMemcpyHostToDevice));
  JDA_HIP(hipMemcpy(d_cmean, cmean.data(), carts * sizeof(Real), hipMemcpyHostToDevice));
  JDA_HIP(hipMemcpy(d_cstd, cstd.data(), carts * sizeof(Real), hipMemcpyHostToDevice));
  JDA_HIP(hipMemcpy(d_cnorm, cnorm.data(), carts, hipMemcpyHostToDevice));
  JDA_HIP(hipMemcpy(d_w, w.data(), w.size() * sizeof(Real), hipMemcpyHostToDevice));
  if (padded) {
    JDA_HIP(hipMemset(d_w_rows, 0, w_rows_n * (size_t)w_pitch * sizeof(Real)));
    JDA_HIP(hipMemcpy2D(d_w_rows, (size_t)w_pitch * sizeof(Real), d_w, (size_t)dim * sizeof(Real), (size_t)dim * sizeof(Real), w_rows_n, hipMemcpyDeviceToDevice));
  }
  JDA_HIP(hipMemcpy(d_ms, ms.data(), dim * sizeof(Real), hipMemcpyHostToDevice));
  JDA_HIP(hipMemcpy(d_ms_raw, ms_raw.data(), dim * sizeof(Real), hipMemcpyHostToDevice));
  JDA_HIP(hipMemcpy(d_par0, par0.data(), par0.size() * sizeof(Real), hipMemcpyHostToDevice));
  DevModelT<Real>& m = mo.m;
  m.T = h.T; m.K = h.K; m.L = h.L; m.D = h.D; m.node_n = node_n; m.leaf_n = leaf_n; m.dim = dim;
  m.nodes = d_nodes; m.lm_off = d_lm_off; m.lm_meta = d_lm_meta; m.leaf = d_leaf; m.cth = d_cth; m.cmean = d_cmean; m.cstd = d_cstd;
  m.lm_deep = d_lm_deep; m.lm_split = (int)split;
  m.w_rows = d_w_rows; m.w_pitch = padded ? w_pitch : dim;
  m.w_stream = (c->kn.w_stream_mb > 0 && (size_t)h.K * leaf_n * (size_t)m.w_pitch * sizeof(Real) > (size_t)c->kn.w_stream_mb << 20) ? 1 : 0;
  m.cnorm = d_cnorm; m.w = d_w; m.mean_shape = d_ms; m.mean_shape_raw = d_ms_raw;
  m.similarity = (sizeof(Real) == 8) ? c->similarity : 0;
  m.par0 = d_par0;
  mo.ready = true;
  return true;
}

// ---------------------------------------------------------------- tiling of levels

// Chooses, per level, how k_scan covers it (DESIGN.md "LDS tiles"): the tile of windows (tw x th, at most
// 512) that share one LDS pixel tile, or the global-pixel mode for windows that do not fit LDS.
//
// Candidates are every tile shape whose workgroup fits LDS; each is priced with a small throughput model
// (CU clocks per frame, constants from the r02 kernel traces) and the cheapest wins:
//   per workgroup   c_fix + pix_bytes / c_bw + slots * c_win        (table load + barriers, tile load, cart walks)
//   per CU          divided by min(1, (waves per CU / w_sat)^alpha)  (latency hiding needs resident waves)
// slots = lanes the tile occupies in phase 0: windows rounded up to whole waves, or to the power of two the
// pair phases pad to for tiles of few windows.  Tile origins need not be multiples of 16 pixels: the
// LDS-DMA loader starts at the 16-byte chunk below and the pitch covers the lead-in.
// JDA_TILES="win:twxth,win:twxth" forces shapes (experiments); JDA_DEBUG_TILES=1 prints the choice.
struct TileChoice { int mode = 0, tw = 1, th = 1, pitch = 0, pix = 0, lds = 0, block = 256; double cost = 0; };

static TileChoice choose_tile(const Level& s, int width, const HostModel& hm, int real_bytes, int chunk, int cp_max,
                              bool ragged = false) {
  static const double c_win = (double)env_ll("JDA_TILE_CWIN", 23), c_bw = (double)env_ll("JDA_TILE_CBW", 32),
                      c_fix = (double)env_ll("JDA_TILE_CFIX", 1500), w_sat = (double)env_ll("JDA_TILE_WSAT", 20),
                      alpha = (double)env_ll("JDA_TILE_ALPHA_PCT", 70) / 100.0;
  const int lds_cu = 160 * 1024;
  static const int lds_max = (int)std::min<long long>(lds_cu, env_ll("JDA_SCAN_LDS_MAX", lds_cu));
  static const char* const tiles_env = std::getenv("JDA_TILES");      // (experiments; read once per process)
  int force_tw = 0, force_th = 0;
  if (const char* e = tiles_env) {
    for (const char* p = e; p && *p;) {
      int w = 0, a = 0, b = 0;
      if (std::sscanf(p, "%d:%dx%d", &w, &a, &b) == 3 && w == s.win) { force_tw = a; force_th = b; }
      p = std::strchr(p, ',');
      if (p) p++;
    }
  }
  TileChoice best;
  const int fixed256 = (int)scan_lds_bytes(0, chunk, hm.node_n(), hm.leaf_n(), real_bytes, true, 256);
  const int fixed512 = (int)scan_lds_bytes(0, chunk, hm.node_n(), hm.leaf_n(), real_bytes, true, 512);
  const int tw_hi = std::min(s.nx, 128), th_hi = std::min(s.ny, 128);
  for (int th = 1; th <= th_hi; th++) {
    for (int tw = 1; tw <= tw_hi; tw++) {
      if (force_tw && (tw != force_tw || th != force_th)) continue;
      const int n_tile = tw * th;
      if (n_tile > 512) break;
      // no point in tiles smaller than a pair-phase round unless the level itself is that small
      if (!force_tw && n_tile < 16 && n_tile < s.nx * s.ny && (long long)s.win * s.win < 64 * 1024) continue;
      const int tiles_x = (s.nx + tw - 1) / tw, tiles_y = (s.ny + th - 1) / th;
      const int pw = s.win + (tw - 1) * s.step, ph = s.win + (th - 1) * s.step;
      int xs = 0;
      if (ragged) {
        // images of any width share the tile's LDS pitch: the worst lead-in of a tile origin x0 = tx * tw * step
        // (and an image may re-cut the tile narrower, ragged_tile: any lead-in below 16 can occur)
        xs = 15;
      } else {
        for (int tx = 0; tx < tiles_x; tx++) xs = std::max(xs, (tx * tw * s.step) & 15);
        static const long long force_xs = env_ll("JDA_TILE_XS", -1);     // (experiment: the ragged chooser's worst-case lead-in)
        if (force_xs >= 0) xs = (int)force_xs;
      }
      int pitch = (xs + pw + 15) & ~15;
      if ((pitch & 127) == 0) pitch += 16;          // keep tile rows off a 32-bank multiple
      const long long pix = (long long)pitch * ph;
      const int block = n_tile > 256 ? 512 : 256;
      const long long lds = (block == 512 ? fixed512 : fixed256) + ((pix + 15) & ~15LL);
      if (lds > lds_max) continue;                   // (the pitch is not monotonic in tw: a wider tile can fit again)
      const long long max_off = (long long)(s.win - 1) * pitch + s.win - 1 + 15;
      if (max_off >= (1LL << kS0GlobalOffBits)) continue;
      const int mode = max_off <= 65535 ? 1 : 3;
      const int wgs = (int)std::min<long long>(lds_cu / lds, 32 / (block / 64));
      const double waves = (double)wgs * (block / 64);
      int slots = (n_tile + 63) & ~63;
      if (n_tile <= cp_max) { slots = 16; while (slots < n_tile) slots *= 2; }
      const double eff = std::min(1.0, std::pow(waves / w_sat, alpha));
      const double cost = (double)tiles_x * tiles_y * (c_fix + (double)pix / c_bw + (double)slots * c_win) / eff;
      if (best.mode == 0 || cost < best.cost) {
        best.mode = mode; best.tw = tw; best.th = th; best.pitch = pitch; best.pix = (int)pix; best.lds = (int)lds;
        best.block = block; best.cost = cost;
      }
    }
  }
  (void)width;
  return best;
}

// ragged: sp holds the global level list of a ragged batch with NOMINAL grids (the mean nx, ny over the images that
// have the level) and the common row pitch as its width; the shapes must suit every image
static void assign_tiles(const ScanPlan& sp, const HostModel& hm, const Knobs& kn, bool fast_scan, int real_bytes, PlanEntry* pe,
                         bool ragged = false) {
  DevPlan& hp = pe->hp;
  hp.n_levels = (int)sp.levels.size();
  hp.width = sp.width; hp.height = sp.height; hp.windows = (int)sp.windows;
  int table = 0;
  pe->any_untiled = false;
  const int handoff = (int)kn.handoff;
  const int chunk = std::min(std::min(hm.K, handoff), scan_handoff_cap(hm.node_n(), hm.leaf_n(), real_bytes));
  const int cp_max = (int)std::max<long long>(0, std::min<long long>(256, kn.cp_max));
  // a level's cost per window in global-pixel mode, in the units of choose_tile (r01: 0.38 ms for 952 k windows)
  const double glb_per_window = (double)kn.tile_cglb;
  for (int i = 0; i < hp.n_levels; i++) {
    const Level& s = sp.levels[i];
    DevLevel& d = hp.lv[i];
    d.win = s.win; d.step = s.step; d.nx = s.nx; d.ny = s.ny; d.base = (int)s.base;
    d.tiled = 0; d.tw = d.th = 1; d.tiles_x = d.tiles_y = 0; d.pitch = 0; d.s0_table = 0;
    const bool glb_ok = kn.no_global_scan == 0 &&
                        (long long)(s.win - 1) * sp.width + s.win - 1 < (1LL << kS0GlobalOffBits);
    if (fast_scan) {
      const TileChoice t = (kn.no_lds_scan || s.win > kn.lds_win_max) ? TileChoice() : choose_tile(s, sp.width, hm, real_bytes, chunk, cp_max, ragged);
      if (t.mode && (!glb_ok || t.cost <= glb_per_window * (double)s.nx * s.ny)) {
        d.tiled = t.mode; d.tw = t.tw; d.th = t.th; d.pitch = t.pitch;
      } else if (glb_ok) {
        // no LDS tile: k_scan reads the frame through L1/L2 (the offsets fit the packed node).  The tile is only a
        // grouping of up to 512 windows per workgroup here: the shape that wastes the fewest lane slots of the
        // first phase (a fixed 32 x 16 filled about half of them on the big-window levels of 640x480)
        d.tiled = 2; d.tw = 32; d.th = 16; d.pitch = sp.width;
        if (kn.glb_tile_fit) {
          long long best = -1;
          for (int th = 1; th <= std::min(s.ny, 512); th++)
            for (int tw = 1; tw <= std::min(s.nx, 512) && tw * th <= 512; tw++) {
              const int n_tile = tw * th;
              int slots = (n_tile + 63) & ~63;
              if (n_tile <= cp_max) { slots = 16; while (slots < n_tile) slots *= 2; }
              const long long tiles = (long long)((s.nx + tw - 1) / tw) * ((s.ny + th - 1) / th);
              const long long cost = tiles * (slots + 96);          // (+ a fixed cost per workgroup: table load, barriers)
              if (best < 0 || cost < best) { best = cost; d.tw = tw; d.th = th; }
            }
        }
      }
      if (kn.debug_tiles)
        std::fprintf(stderr, "[jda] level %d win %d step %d windows %dx%d: mode %d tile %dx%d pitch %d pix %d lds %d block %d cost/window %.0f\n",
                     i, s.win, s.step, s.nx, s.ny, d.tiled, d.tw, d.th, d.pitch, t.pix, t.lds, t.block,
                     t.mode ? t.cost / ((double)s.nx * s.ny) : 0.0);
    }
    if (!d.tiled) { pe->any_untiled = true; continue; }
    d.tiles_x = (s.nx + d.tw - 1) / d.tw;
    d.tiles_y = (s.ny + d.th - 1) / d.th;
    d.s0_table = table;
    table += hm.K * hm.node_n();
  }
}

// The plan of (frame size, call parameters), built on first use.  Caller holds c->mu.  The plan comes back PINNED
// (PlanEntry::pins): it is not evicted -- its device tables are not recycled -- until unpin_plan.
static bool get_plan(Cascador* c, const PlanKey& key, const ScanPlan& sp, int dialect, PlanEntry** out, bool ragged = false) {
  auto it = c->plans.find(key);
  if (it != c->plans.end()) { it->second.last_use = ++c->plan_clock; it->second.pins++; *out = &it->second; return true; }
  // bounded cache: a stream of differently sized images (FDDB) must not pile up device tables
  const size_t cap = (size_t)std::max<long long>(2, c->kn.plan_cache);
  while (c->plans.size() >= cap) {
    // least recently used plan that no submitted batch still runs on (PlanEntry::pins): a pending ticket's kernels
    // read the plan's device tables until its Wait
    auto victim = c->plans.end();
    for (auto p = c->plans.begin(); p != c->plans.end(); ++p)
      if (p->second.pins == 0 && (victim == c->plans.end() || p->second.last_use < victim->second.last_use)) victim = p;
    if (victim == c->plans.end()) break;        // every plan is in use: exceed the cap for now
    c->plan_pool.push_back({victim->second.dp, victim->second.table, victim->second.table_cap});
    c->plans.erase(victim);
  }
  if ((int)sp.levels.size() > kMaxLevels) { fail("too many pyramid levels"); return false; }
  if (!ragged && sp.windows * 1LL > 0x7fffffffLL) { fail("frame has too many windows"); return false; }
  if (sp.width > 65535 || sp.height > 65535) { fail("frames wider or taller than 65535 pixels are not supported"); return false; }
  PlanEntry pe;
  pe.sp = sp;
  // LDS-tiled stage-0 scan needs every stage-0 node to read the origin image
  bool s0_plain = true;
  const size_t n0 = (size_t)c->hm.K * c->hm.node_n();
  for (size_t i = 0; i < n0; i++) s0_plain = s0_plain && c->hm.nodes[i].scale == 0;
  pe.fast_scan = s0_plain && c->kn.no_fast_scan == 0;
  assign_tiles(sp, c->hm, c->kn, pe.fast_scan, dialect == JDA_DIALECT_C ? 4 : 8, &pe, ragged);
  size_t entries = 0;
  pe.lm_ok = true;
  for (int i = 0; i < pe.hp.n_levels; i++)
    if (pe.hp.lv[i].tiled) {
      entries += n0;
      if (pe.hp.lv[i].win > 2047) pe.lm_ok = false;      // (x, y) inside the window: 11 bits each
    }
  if (!c->plan_pool.empty()) {            // recycle an evicted plan's allocations
    Cascador::PlanBuffers b = c->plan_pool.back();
    c->plan_pool.pop_back();
    pe.dp = b.dp; pe.table = b.table; pe.table_cap = b.table_cap;
    if (pe.table_cap < entries) { if (pe.table) (void)hipFree(pe.table); pe.table = nullptr; pe.table_cap = 0; }
  }
  // (a failure below must not lose the device allocations: whatever the entry holds goes back to the pool)
  auto build = [&]() -> bool {
    if (!pe.dp) JDA_HIP(hipMalloc((void**)&pe.dp, sizeof(DevPlan)));
    JDA_HIP(hipMemcpy(pe.dp, &pe.hp, sizeof(DevPlan), hipMemcpyHostToDevice));
    if (entries) {
      if (!pe.table) {
        pe.table_cap = std::max(entries, (size_t)16 * n0);      // room for 16 levels: most recycled tables fit the next plan
        JDA_HIP(hipMalloc((void**)&pe.table, 2 * pe.table_cap * sizeof(S0Node)));   // cart-major tables + their level-major copy
      }
      const void* nodes = dialect == JDA_DIALECT_C ? c->mf.m.nodes : c->md.m.nodes;
      const void* ms = dialect == JDA_DIALECT_C ? (const void*)c->mf.m.mean_shape : (const void*)c->md.m.mean_shape;
      JDA_HIP(launch_prep_stage0(dialect, pe.dp, pe.hp, nodes, ms, c->hm.K, c->hm.node_n(), pe.table, pe.table + pe.table_cap, c->aux));
      // the scans that read the table run on the lanes' streams
      JDA_HIP(hipStreamSynchronize(c->aux));
    }
    return true;
  };
  if (!build()) {
    if (pe.dp || pe.table) c->plan_pool.push_back({pe.dp, pe.table, pe.table ? pe.table_cap : 0});
    return false;
  }
  pe.last_use = ++c->plan_clock;
  pe.pred_tail = c->pred_tail; pe.pred_out = c->pred_out;      // a new frame size starts from the cascador's last pass
  pe.dense_hint = c->last_dense;
  pe.pins = 1;
  auto ins = c->plans.emplace(key, std::move(pe));
  *out = &ins.first->second;
  return true;
}

static void unpin_plan(Cascador* c, PlanEntry* pe) {
  if (!pe) return;
  std::lock_guard<std::mutex> lk(c->mu);
  if (pe->pins > 0) pe->pins--;
}

// ---------------------------------------------------------------- workspace

template <typename Real>
static size_t bytes_per_window(int dim, bool trace) {
  size_t b = (4 + sizeof(Real) + 4 + 8) + 2 * (4 + sizeof(Real) + (size_t)dim * sizeof(Real)) + 8 + 4;
  if (trace) b += 4 + 4 + sizeof(Real) + 4 + (size_t)dim * sizeof(Real);
  return b;
}

// The lane's per-window arrays for `cap` windows of dialect Real (grow-only; the lane is idle: its holder has
// collected whatever ran on it).
template <typename Real>
static bool ensure_workspace(Lane* ln, size_t cap, bool trace, int dim) {
  if (ln->cap >= cap && (ln->trace || !trace) && ln->dim == dim && ln->real_bytes == (int)sizeof(Real)) return true;
  trace = trace || (ln->trace && ln->dim == dim && ln->real_bytes == (int)sizeof(Real));
  cap = std::max(cap, ln->real_bytes == (int)sizeof(Real) && ln->dim == dim ? ln->cap : (size_t)0);
  WorkT<Real>& w = Sel<Real>::work(ln);
  auto carve = [&](Carver& cv) {
    w.q_gid = cv.take<uint32_t>(cap);
    w.q_score = cv.take<Real>(cap);
    w.q_kstart = cv.take<uint32_t>(cap);
    w.q_xy = cv.take<uint32_t>(cap);
    w.q_wf = cv.take<uint32_t>(cap);
    w.q_hash = trace ? cv.take<uint32_t>(cap) : nullptr;
    w.m_gid = cv.take<uint32_t>(cap);
    w.m_score = cv.take<Real>(cap);
    w.m_hash = trace ? cv.take<uint32_t>(cap) : nullptr;
    w.m_shape = cv.take<Real>(cap * dim);
    w.m_xy = cv.take<uint32_t>(cap);
    w.m_wf = cv.take<uint32_t>(cap);
    w.st_carts = cv.take<int>(cap);
    w.out_gid = cv.take<uint32_t>(cap);
    w.out_score = cv.take<Real>(cap);
    w.out_shape = cv.take<Real>(cap * dim);
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
  };
  if (ln->stream) (void)hipStreamSynchronize(ln->stream);      // nothing may still use the old carving
  w = WorkT<Real>{};
  Carver sz(nullptr);
  carve(sz);
  if (!ln->ws.reserve(sz.off + 256)) {
    // the old allocation is gone: forget every pointer carved out of it
    ln->cap = 0; ln->trace = false; ln->real_bytes = 0;
    ln->wf = WorkT<float>{}; ln->wd = WorkT<double>{};
    return false;
  }
  Carver cv(ln->ws.p);
  carve(cv);
  w.cap = (unsigned)cap;
  ln->cap = cap; ln->trace = trace; ln->dim = dim; ln->real_bytes = (int)sizeof(Real);
  return true;
}

// ---------------------------------------------------------------- the pipeline

template <typename Real>
struct RawDets {               // survivors of a batch, sorted by gid (= frame, then scan order)
  std::vector<uint32_t> gid;
  std::vector<Real> score;
  std::vector<Real> shape;     // [n][dim]
};

template <typename Real>
struct TraceOut {              // host arrays, may be null
  int* carts_n; Real* score; unsigned* path_hash; Real* shapes;
};

struct RunStats {
  // Device-side spans (gpu_ms, scan_ms, scan_lds_ms) are wanted: the pass brackets its steps with events.  Off for
  // callers that did not ask for statistics -- each record is a marker packet the command processor works off
  // between the kernels, a few microseconds apiece, five per pass.
  bool timed = true;
  long long carts = 0, out = 0, carts_scan = 0, carts_scan_glb = 0, win_scan = 0, tail = 0;
  long long stage_done[kMaxStages] = {0};
  double gpu_ms = 0, scan_ms = 0, scan_lds_ms = 0;
  int scan_launches = 0;
  int dense_passes = 0;
};

// Copies a run of host frames to the staging buffer (frame i at dst + i*stride) on a stream.
// Frames that lie back to back in host memory (one array) go as ONE strided copy: 256 separate
// 300-KB copies from pageable memory reach ~15 GB/s, one copy of the batch 57 GB/s (tools/pcie_bw.py).
static bool copy_frames_h2d(uint8_t* dst, size_t stride, const unsigned char* const* frames, int n, size_t fbytes,
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

// ---- ragged batches: images of different sizes in one pass (kernels.h: RagSeg) ----
// Host tables of one chunk of a ragged job: what its pass uploads and launches.
struct RaggedChunk {
  int i0 = 0, n = 0;                    // images [i0, i0 + n) of the job
  long long windows = 0;                // candidate windows of the chunk
  size_t frame_bytes = 0;               // staged images (common row pitch)
  size_t raw_bytes = 0;                 // tight images (host staging; 0 when the images are already on the device)
  int max_h = 0, pitch = 0;
  std::vector<uint32_t> gid_base;       // [n + 1] first gid of every image inside the pass
  struct Launch { int mode, block, pix_bytes, blk_base, blk_n; };
  std::vector<Launch> launches;
  int n_segs = 0, n_blk = 0;
  size_t off_segs = 0, off_blk = 0, off_imgoff = 0, off_rimg = 0, table_bytes = 0;   // layout of the table buffer
  const unsigned char* const* host_imgs = nullptr;   // the chunk's images in host memory (tight), or
  const uint8_t* d_raw = nullptr;                    // the base their RagImg::src_off refer to on the device
  const int* widths = nullptr; const int* heights = nullptr;      // of the chunk's images
  bool host_contiguous = false;         // the host images lie back to back in memory in RagImg::src_off order
  const uint8_t* d_uploaded = nullptr;  // the chunk's tight images are (being) uploaded here by the job's helper thread
};

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
    w = Sel<Real>::work(l); cap = l->cap;
    std::lock_guard<std::mutex> lk(c->mu);
    hint_dense = pe->dense_hint; pred_tail = pe->pred_tail; pred_out = pe->pred_out; pred_mid = pe->pred_mid;
    busy_lanes = 0;
    for (auto& up : c->lanes) busy_lanes += up->busy ? 1 : 0;
  }
  WorkT<Real> w; size_t cap = 0;
  int f0 = 0, nf = 0;
  const unsigned char* const* host_frames = nullptr; size_t host_fbytes = 0;   // frames of this sub-batch still on the host
  const RaggedChunk* rag = nullptr;   // ragged pass: images of different sizes (w.segs / w.blk / w.img_off set by stage_ragged)
  // state between the steps
  bool dense = false, finished = false, lds_span = false;
  bool timed = true;               // RunStats::timed
  bool predicted = false;          // the finishing launches were sized from PlanEntry::pred_tail, no host wait in between
  bool counters_issued = false, results_pending = false;
  int p_launches = 0;              // k_scan_p launches of this pass so far (each deals its tiles from its own counter words)
  bool mid_direct = false;         // a scan launch of this pass put stage-0 survivors into the mid queue itself (k_scan_p up to cart K)
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
      if (!c->h2d) JDA_HIP(hipStreamCreateWithFlags(&c->h2d, hipStreamNonBlocking));
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
  bool scan_persistent(int level, hipStream_t s) {
    if constexpr (sizeof(Real) != 4) { (void)level; (void)s; return false; }
    else {
      if (!kn().scan_p || want_trace()) return false;
      const DevModelT<Real>& m = model();
      const DevLevel& lv = pe->hp.lv[level];
      if (lv.win > kn().scan_p_win_max) return false;
      const int K = std::min(m.K, (int)(kn().scan_p_handoff > 0 ? kn().scan_p_handoff : kn().handoff));
      PScanCfg cfg{};
      // all of stage 0 in this kernel: its survivors are what k_filter0 would leave in the mid queue (launch_finishers
      // then takes the k_filter0 + k_finish(survivors) form whatever the size of the hand-off queue)
      cfg.to_mid = (K == m.K && kn().scan_p_mid && filter0_ok()) ? 1 : 0;
      const long long bs[5] = {kn().scan_p_b0, kn().scan_p_b1, kn().scan_p_b2, kn().scan_p_b3, kn().scan_p_b4};
      int digits[kPScanMaxBuckets] = {6, 6, 6, 6, 6, 6}, nd = 0;
      { long long v = std::max<long long>(0, kn().scan_p_lg); int tmp[16]; int n = 0; while (v > 0 && n < 16) { tmp[n++] = (int)(v % 10); v /= 10; }
        for (int i = n - 1; i >= 0 && nd < kPScanMaxBuckets; i--) digits[nd++] = tmp[i]; }
      int last = 0;
      for (int i = 0; i < 5 && cfg.nb < kPScanMaxBuckets; i++) {
        const int b = (int)bs[i];
        if (b <= last || b >= K) continue;
        cfg.bound[cfg.nb] = b;
        const int d = digits[cfg.nb];
        cfg.lg[cfg.nb] = ((d == 4 || d == 5 || d == 7 || d == 8 || d == 9) && m.leaf_n <= 256) || d == 2 || d == 3 ? d : 6;
        cfg.nb++;
        last = b;
      }
      cfg.bound[cfg.nb] = K;
      cfg.bound_last = K;
      cfg.any_norm = stage0_any_norm(c->hm, K, true) ? 1 : 0;
      const int block = (int)std::max<long long>(64, std::min<long long>(1024, kn().scan_p_block)) & ~63;
      const int wgs = (int)std::max<long long>(1, std::min<long long>(8, kn().scan_p_wgs));
      cfg.ring_cap[0] = (int)std::max<long long>(64, std::min<long long>(4096, kn().scan_p_ring));
      scan_p_ring_caps(&cfg, block / 64);
      { const unsigned mg = ((1u << 20) + (unsigned)lv.tw - 1u) / (unsigned)lv.tw; bool ok = true;
        for (unsigned i = 0; i < (unsigned)(lv.tw * lv.th + 64) && ok; i++) ok = ((i * mg) >> 20) == i / (unsigned)lv.tw;
        cfg.tw_magic = ok ? (int)mg : 0; }
      cfg.opts = (int)kn().scan_p_opts;
      // the kernel's own cut of the tile in y (same row pitch and tile width: the resolved node offsets hold): small
      // tiles turn over faster and leave room for more slots.  Candidates are the heights whose windows fill their
      // waves to 90 % (or the best filled one); the tallest that keeps the pixel tile within scan_p_tile_kb, else the
      // smallest
      cfg.th = lv.th;
      if (kn().scan_p_tile_kb > 0) {
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
      cfg.slot_bytes = (lv.pitch * (lv.win + (cfg.th - 1) * lv.step) + 15) & ~15;
      cfg.slots = 0;
      const long long fixed = (long long)scan_p_lds_bytes(cfg, K, m.node_n, m.leaf_n, block / 64);
      const long long budget = std::max<long long>(16, std::min<long long>(160, kn().scan_p_lds_kb)) * 1024 / wgs;
      long long slots = (budget - fixed) / std::max(1, cfg.slot_bytes);
      // (the cap only where another batch's kernels are in flight next to this pass -- a second lane of this call,
      // other tickets or callers; alone, the workgroup takes every slot that fits)
      if (kn().scan_p_slots > 0 && (!solo || busy_lanes > 1)) slots = std::min<long long>(slots, kn().scan_p_slots);
      slots = std::min<long long>(slots, 8);
      if (slots < 2 || fixed + slots * cfg.slot_bytes >= (1 << 18)) return false;
      if (kn().scan_p == 1 && slots < kn().scan_p_min_slots) return false;     // few resident windows per wave: k_scan's closed tiles do better there
      cfg.slots = (int)slots;
      if (kn().scan_p == 1 && (long long)lv.tiles_x * cfg.tiles_y * nf < (long long)c->n_cus * wgs * 4) return false;   // too few tiles to keep persistent workgroups fed
      cfg.dyn_slot = (kn().scan_p_dyn && p_launches < kCntMidScan - kCntTotal) ? p_launches : -1;
      const int grid = kn().scan_p_grid > 0 ? (int)std::min<long long>(kn().scan_p_grid, 1 << 16) : c->n_cus * wgs;
      const hipError_t e = launch_scan_persistent(level, cfg, block, grid, pe->dp, pe->hp, m, pe->table, w, s);
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
    if (timed) JDA_HIP(hipEventRecord(ev[0], st));
    if (rag) return issue_scan_ragged();
    if (host_frames && !upload_frames(const_cast<uint8_t*>(w.frames), w.frame_stride, host_frames, nf, host_fbytes)) return false;
    if (multi) {
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
        if (mode == 1 && level >= 0 && scan_persistent(level, s)) { rs->scan_launches++; return true; }
        JDA_HIP(launch_scan<Real>(mode, level, want_trace(), handoff, cp_max, opts, pe->dp, pe->hp, m, pe->table, w, s));
        rs->scan_launches++;
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
        const bool side = any_glb && solo && kn().side_stream && ln->ensure_side();
        int fork_in = side ? (int)std::max<long long>(0, kn().side_after) : -1;
        if (fork_in == 0) { if (!fork_glb()) return false; fork_in = -1; }
        if (rev && any_glb) { if (!scan(2, -1, st)) return false; any_glb = false; }
        for (int li = 0; li < pe->hp.n_levels; li++) {
          const int l = rev ? pe->hp.n_levels - 1 - li : li;
          const int mode = pe->hp.lv[l].tiled;
          if (mode != 1 && mode != 3) continue;
          // big-window levels of a batch are short launches: merge all of them into one (at the first one met)
          if (mode == 3) { if (any_wide) { if (!scan(3, -1, st)) return false; any_wide = false; } continue; }
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
      const long long guess = std::min<long long>((long long)cap, (long long)(pred_tail * (double)nw * 1.1) + 64);
      if (!launch_finishers(guess)) return false;
      predicted = true;
      const double po = pred_out >= 0 ? pred_out : 0.0;
      const size_t to = std::min<size_t>(cap, (size_t)(po * (double)nw * 1.25) + 64);
      if (kn().kernel_d2h && dets && to > 0) return issue_results(0, to, true);      // counters + prefix in one launch
      return issue_counters() && issue_results(0, to);
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
    w.frames = (const uint8_t*)ln->rag_frames.p; w.frame_stride = 0; w.n_frames = ch.n;
    w.segs = (const RagSeg*)(tab + ch.off_segs); w.blk = (const RagBlk*)(tab + ch.off_blk);
    w.img_off = (const unsigned long long*)(tab + ch.off_imgoff);
    if (!clear_counters()) return false;
    if (timed) JDA_HIP(hipEventRecord(ev[1], st));
    const int handoff = (int)kn().handoff;
    const int cp_max = (int)std::max<long long>(0, std::min<long long>(256, kn().cp_max));
    const int opts = ((int)(std::max<long long>(4, std::min<long long>(64, kn().first_phase)) & ~3LL) << 8) |
                     (kn().scan_lean && !stage0_any_norm(c->hm, handoff, sizeof(Real) == 4) ? 2 : 0);
    for (const RaggedChunk::Launch& l : ch.launches) {
      JDA_HIP(launch_scan_ragged<Real>(l.mode, l.block, false, handoff, cp_max, opts, pe->dp, m, pe->table, w, l.pix_bytes,
                                       l.blk_base, l.blk_n, st));
      rs->scan_launches++;
    }
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
      const long long wg2 = std::min<long long>((long long)cap, std::max<long long>(std::max<long long>(2048, n_grid / std::max<long long>(1, kn().fin_grid_div)), nmid));
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
      JDA_HIP(launch_finish_wide<Real>(want_trace(), apply_th, th, pe->dp, model(), w, n_grid, s0_tbl(), st));
      finished = true;
      return true;
    }
    if (T == 1 || n_grid <= kn().finish_merge) {
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
    n_tail = (long long)std::min<unsigned long long>(h_cnt[0], cap);
    const long long n_alive = n_tail + (mid_direct ? (long long)std::min<unsigned long long>(h_cnt[kCntMid - kCntTail], cap) : 0);
    int pix_cap, lds_max;
    const double dense_frac = (double)kn().dense_pct / 100.0;
    if (dense_ok(&pix_cap, &lds_max) && (double)n_alive >= dense_frac * (double)windows() && n_alive > 4096) {
      // most windows are still alive after the scan: start over in dense mode (the scan's work
      // is a small part of T*K carts per window) and remember the choice for the next pass
      { std::lock_guard<std::mutex> lk(c->mu); pe->dense_hint = true; }
      if (rag) return launch_finishers(n_tail);   // (a ragged pass finishes window by window; the NEXT job runs image by image, dense)
      dense = true; finished = true;
      if (!clear_counters()) return false;
      return run_dense();
    }
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
    JDA_HIP(hipStreamSynchronize(st));
    results_pending = false;
    for (int shd = 1; shd < kCntShards; shd++) {   // fold the counter shards into shard 0
      for (int i = 0; i < kCntTotal; i++) h_cnt[i] += h_cnt[shd * kCntStride + i];
      h_cnt[kCntMidScan] += h_cnt[shd * kCntStride + kCntMidScan];
    }
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
    if (n_out > cap) { fail("internal: more detections than windows"); return false; }
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
    if (n_out > out_copied && !issue_results(out_copied, n_out)) return false;   // the prediction fell short (or there was none)
    return true;
  }

  // step 6: detections of this pass sorted back into scan order and appended; trace arrays
  bool collect() {
    const int dim = hm().dim();
    const long long wpf = rag ? 0 : pe->sp.windows;
    const double t_dbg = now_ms();
    if (n_out && dets) {
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

// test hook: JDA_TEST_WPF_SCALE pretends every frame has that many times more windows (the gid-overflow guard
// is otherwise only reachable with thousands of 4K frames)
static bool jda_gid_overflow(const Knobs& kn, long long n, long long wpf) {
  const long long scale = std::max<long long>(1, kn.test_wpf_scale);
  return (double)n * (double)wpf * (double)scale > 4294967295.0;
}

// Frames of a call that are still in host memory: run_device copies them sub-batch by sub-batch into the staging buffer
// of the call's first lane (the copies of one sub-batch then overlap the kernels of the other lane).
struct HostFrames {
  const unsigned char* const* ptrs = nullptr;
  size_t fbytes = 0;
};

// Runs the device pipeline over n frames in device memory (d_frames; with host.ptrs set they are copied there first,
// sub-batch by sub-batch).  `lanes` holds the call's first lane; a large batch takes a second one from the pool and
// is split into sub-batches that alternate between the two (streams with their own workspace), see Pass.
template <typename Real>
static bool run_device_impl(Cascador* c, LaneSet& lanes_held, PlanEntry* pe, const uint8_t* d_frames, size_t stride, int n,
                            bool apply_th, Real th, hipStream_t user_stream, RawDets<Real>* dets,
                            const TraceOut<Real>* trace, RunStats* rs, HostFrames host);

// A pass that fails half way (an allocation, a launch, a detection list beyond its capacity) leaves work queued on the
// lanes' streams: kernels that still read the caller's frames, copies out of the caller's host memory, writes into
// the lanes' pinned buffers.  The lanes go back to the pool and the caller may free its frames as soon as this
// returns, so everything queued is waited for first.
template <typename Real>
static bool run_device(Cascador* c, LaneSet& lanes_held, PlanEntry* pe, const uint8_t* d_frames, size_t stride, int n,
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
static bool run_device_impl(Cascador* c, LaneSet& lanes_held, PlanEntry* pe, const uint8_t* d_frames, size_t stride, int n,
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
  const size_t bpw = bytes_per_window<Real>(dim, want_trace);
  const long long budget = (c->kn.workspace_mb << 20) / lanes;
  long long fpp = std::max<long long>(1, budget / (long long)(bpw * (size_t)wpf));
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
  for (int l = 0; l < lanes; l++)
    if (!ensure_workspace<Real>(lanes_held.v[l], cap, want_trace, dim)) return false;

  int hw = 0, hh = 0, qw = 0, qh = 0;
  size_t hs = 0, qs = 0;
  if (multi) {
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

  std::vector<Pass<Real>> ps;
  for (int f0 = 0; f0 < n;) {
    // one round: up to `lanes` sub-batches in flight, their steps interleaved
    ps.clear();
    for (int l = 0; l < lanes && f0 < n; l++) {
      Pass<Real> p;
      p.c = c; p.pe = pe; p.trace = trace; p.dets = dets; p.rs = rs; p.apply_th = apply_th; p.th = th; p.multi = multi;
      p.solo = lanes == 1;
      p.bind(lanes_held.v[l], l, l == 0 ? user_stream : nullptr);
      p.cap = cap;
      p.f0 = f0; p.nf = std::min<int>((int)fpp, n - f0);
      p.w.frames = d_frames + (size_t)f0 * stride; p.w.frame_stride = stride; p.w.n_frames = p.nf;
      if (host_frames) { p.host_frames = host_frames + f0; p.host_fbytes = host_fbytes; }
      p.w.half = nullptr; p.w.quarter = nullptr; p.w.half_stride = p.w.quarter_stride = 0;
      p.w.hw = hw; p.w.hh = hh; p.w.qw = qw; p.w.qh = qh;
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

// window of a gid
struct WinRef { int frame, x, y, win; };
static WinRef locate(const ScanPlan& sp, uint32_t gid) {
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
  }

 private:
  struct Job {
    const std::function<void(int)>* fn = nullptr;   // valid until every chunk is done (run() waits for that)
    int n = 0, chunks = 0, chunk = 8;
    std::atomic<int> next{0}, done{0};
  };
  static void work(Job& j) {
    for (int c; (c = j.next.fetch_add(1)) < j.chunks;) {
      const int e = std::min(j.n, (c + 1) * j.chunk);
      for (int i = c * j.chunk; i < e; i++) (*j.fn)(i);
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

static void parallel_for(int n, const std::function<void(int)>& fn, bool small_job = false) {
  PostPool::get().run(n, fn, !small_job);
}

static void fill_stats(jdaStats* st, const RunStats& rs, long long patch_n, int T, int K, double host_ms) {
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
  st->scan_lds_ms = rs.scan_lds_ms; st->scan_lds_cart_n = rs.carts_scan - rs.carts_scan_glb;
}

static jdaResult empty_result(int landmark_n) {
  jdaResult r;
  r.n = 0; r.landmark_n = landmark_n;
  r.bboxes = (int*)std::malloc(sizeof(int));
  r.shapes = (float*)std::malloc(sizeof(float));
  r.scores = (float*)std::malloc(sizeof(float));
  return r;
}

// NMS, relocation and the jdaResult of every frame of a dialect-C batch from its raw detections
// (sorted by gid = frame, then scan order).  Returns the time it took (ms).
static double post_c(Cascador* c, const ScanPlan& sp, const RawDets<float>& dets, int n, const jdaDetectOptions* opt,
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
  parallel_for(n, [&](int f) {
    const size_t a = first[f], cnt = first[f + 1] - a;
    static thread_local std::vector<int> bb, keep;          // per-frame scratch, grown once per thread
    bb.resize(cnt * 3);
    for (size_t i = 0; i < cnt; i++) {
      const WinRef wr = locate(sp, dets.gid[a + i]);
      bb[3 * i] = wr.x; bb[3 * i + 1] = wr.y; bb[3 * i + 2] = wr.win;
    }
    if (do_nms) nms_dialect_c_into(bb.data(), &dets.score[a], (int)cnt, overlap, &keep);
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

// Validation + plan of a dialect-C call (shared by the synchronous and the submit/wait entries).  Takes c->mu for
// the shared parts (device, model, plan cache); the plan comes back pinned.
static bool plan_c_call(Cascador* c, size_t stride, int width, int height, float scale, int min_size, int max_size,
                        ScanPlan* sp, PlanEntry** pe) {
  std::string err;
  if (!plan_dialect_c(width, height, scale, min_size, max_size, sp, &err)) { fail(err); return false; }
  if (stride < (size_t)width * height) { fail("frame_stride smaller than a frame"); return false; }
  std::lock_guard<std::mutex> lk(c->mu);
  if (!ensure_device(c) || !upload_model<float>(c)) return false;
  unsigned sb; std::memcpy(&sb, &scale, 4);
  PlanKey key{width, height, JDA_DIALECT_C, (int)sb, std::max(min_size, 24), max_size <= 0 ? -1 : max_size, 0ull};
  return get_plan(c, key, *sp, JDA_DIALECT_C, pe);
}

struct PlanPin {           // unpins on scope exit
  Cascador* c; PlanEntry* pe;
  ~PlanPin() { unpin_plan(c, pe); }
};

// The shared part of an entry, under c->mu: device, the model of dialect Real on the device, the plan (pinned).
template <typename Real>
static bool begin_call(Cascador* c, const PlanKey& key, const ScanPlan& sp, int dialect, PlanEntry** pe) {
  std::lock_guard<std::mutex> lk(c->mu);
  if (!ensure_device(c) || !upload_model<Real>(c)) return false;
  return get_plan(c, key, sp, dialect, pe);
}

// Dialect CPP walks stages [0, current_stage_idx) and then carts [0, current_cart_idx] of the next one
// (cascador.cpp:178,199-209): header ints 5, 6 of the model file (cascador.cpp:93-104).  A complete model carries
// (T, -1); the float files of the C library carry (T+1, -1) (c/jda.c:662-665), which the reference's own C++ loader
// would walk out of bounds with -- both mean "every stage" here.  Training snapshots (fewer stages, or a stage cut
// at a cart) are refused by the dialect-CPP entries instead of being silently run to the end; dialect C ignores the
// header like c/jda.c:499-505 does.
static bool cpp_model_complete(const Cascador* c) {
  const HostModel& h = c->hm;
  if ((h.hdr_stage == h.T || h.hdr_stage == h.T + 1) && h.hdr_cart == -1) return true;
  fail("partial model (training snapshot: header says stage " + std::to_string(h.hdr_stage) + ", cart " + std::to_string(h.hdr_cart) +
       " of T=" + std::to_string(h.T) + "): the dialect-CPP entries run complete models only");
  return false;
}

// ... without a plan (entries that only need a lane)
static bool begin_device(Cascador* c) {
  std::lock_guard<std::mutex> lk(c->mu);
  return ensure_device(c);
}

// Reserves the lane's staging buffer for n host frames.  defer = false: copies them now and waits; defer = true:
// leaves the copies to run_device (per sub-batch).
static bool stage_frames(Lane* ln, const unsigned char* const* frames, int n, size_t fbytes, size_t* stride,
                         bool defer = false) {
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
static int detect_c_device(Cascador* c, const uint8_t* d_frames, size_t stride, int n, int width, int height,
                           float scale, int min_size, int max_size, float th, const jdaDetectOptions* opt,
                           jdaResult* out, const unsigned char* const* host_frames = nullptr) {
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

// ---- submit / wait: batches in flight on one cascador, driven by one host thread ----
// Submit queues a batch (no host wait) on a lane of its own; Wait collects and post-processes it.  A caller that
// submits batch i+1 before it waits for batch i keeps the GPU busy with batch i+1 while the host parts of batch i run.
static int submit_c_device(Cascador* c, const uint8_t* d_frames, size_t stride, int n, int width, int height,
                           float scale, int min_size, int max_size, float th, const jdaDetectOptions* opt,
                           const unsigned char* const* host_frames = nullptr) {
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
  if (!ensure_workspace<float>(ln, cap, false, c->hm.dim())) return -1;
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
  p.bind(ln, 0, nullptr);
  p.f0 = 0; p.nf = n;
  p.w.frames = d_frames; p.w.frame_stride = stride; p.w.n_frames = n;
  p.w.half = nullptr; p.w.quarter = nullptr; p.w.half_stride = p.w.quarter_stride = 0;
  p.w.hw = p.w.hh = p.w.qw = p.w.qh = 0;
  if (host_frames) {
    pb.host_ptrs.assign(host_frames, host_frames + n);       // (the helper thread reads them after Submit has returned)
    p.host_frames = pb.host_ptrs.data(); p.host_fbytes = (size_t)width * height;
  }
  auto give_up = [&]() { std::lock_guard<std::mutex> lk(c->mu); pb.reserved = false; return -1; };
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
      if (hipSetDevice(dev) != hipSuccess || !pbp->pass.issue_scan(nullptr, 0, nullptr, 0, nullptr)) {
        pbp->issue_ok = false;
        pbp->issue_err = g_err.empty() ? std::string("issuing the batch failed") : g_err;
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

static int wait_c_device(Cascador* c, int slot, jdaStats* stats, jdaResult* out) {
  if (!c || slot < 0 || slot >= kTickets || !out) { fail("no pending batch in this slot"); return -1; }
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->pending || !c->pending[slot].active || c->pending[slot].waiting) { fail("no pending batch in this slot"); return -1; }
    c->pending[slot].waiting = true;           // (two threads waiting for one ticket: the second is refused)
    if (!ensure_device(c)) { c->pending[slot].waiting = false; return -1; }   // the waiting thread's current device may differ
  }
  PendingBatch& pb = c->pending[slot];
  const int L = c->hm.L, n = pb.n;
  for (int i = 0; i < n; i++) { out[i].n = 0; out[i].landmark_n = L; out[i].bboxes = nullptr; out[i].shapes = nullptr; out[i].scores = nullptr; }
  Pass<float>& p = pb.pass;
  pb.join_issuer();
  bool ok = pb.issue_ok;
  if (!ok) fail(pb.issue_err);
  p.dets = &pb.dets; p.rs = &pb.rs;
  ok = ok && p.after_tail() && p.after_mid() && p.issue_counters() && p.after_counters() && p.collect();
  if (!ok) (void)hipStreamSynchronize(pb.lane->stream);
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
  }
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (pb.pe && pb.pe->pins > 0) pb.pe->pins--;
    pb.pe = nullptr;
    if (pb.lane) { pb.lane->busy = false; c->lane_cv.notify_all(); }
    pb.lane = nullptr;
    pb.active = false; pb.waiting = false;
  }
  return ok ? 0 : -1;
}

// ---------------------------------------------------------------- ragged batches (images of different sizes)
//
// The reference's FDDB loop calls Detect once per image (src/test.cpp:100-170), the C API once per jdaDetect; on a
// GPU that is one latency-bound pass per image.  A ragged job runs a list of differently sized images as a few
// passes: the window sizes of c/jda.c:331-333 are the same series for every image (an image uses the prefix that fits
// it), so the levels, their tile shapes and stage-0 tables are shared, and the images are staged with ONE row pitch.
// Per image the results are those of jdaDetect on that image.

struct RaggedJob {
  int n = 0;
  const int* widths = nullptr; const int* heights = nullptr;
  const unsigned char* const* host_imgs = nullptr;     // tight images in host memory, or
  const uint8_t* d_base = nullptr; const size_t* d_offsets = nullptr;   // ... on the device at d_base + d_offsets[i]
  int pitch = 0;                    // common row pitch of the staged images (multiple of 16)
  ScanPlan levels;                  // global level list; nx, ny = nominal (mean) grids, width = pitch
  std::vector<int> n_lv;            // levels image i has (a prefix of the global list)
  PlanEntry* pe = nullptr;
  uint8_t* d_job_raw = nullptr;     // host job with a helper thread: every chunk's tight images go here ...
  std::vector<size_t> raw_off;      // ... chunk k at d_job_raw + raw_off[k]
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Levels of image (w, h): the prefix of the job's global list whose windows fit (c/jda.c:321-322,332).
static int ragged_levels_of(const RaggedJob& job, int w, int h) {
  const int lim = std::min(w, h);
  int k = 0;
  while (k < (int)job.levels.levels.size() && job.levels.levels[k].win <= lim) k++;
  return k;
}

// Geometry of a ragged call: common pitch, global levels with nominal grids, the plan (tile shapes + tables).
// Returns 0 = ok, 1 = this job needs the per-image fallback, -1 = error.
static int ragged_prepare(Cascador* c, RaggedJob* job, float scale, int min_size, int max_size) {
  int max_w = 0, max_min = 0;
  for (int i = 0; i < job->n; i++) {
    if (job->widths[i] <= 0 || job->heights[i] <= 0) { fail("image " + std::to_string(i) + " has no pixels"); return -1; }
    if (job->widths[i] > 65535 || job->heights[i] > 65535) { fail("images wider or taller than 65535 pixels are not supported"); return -1; }
    max_w = std::max(max_w, job->widths[i]);
    max_min = std::max(max_min, std::min(job->widths[i], job->heights[i]));
  }
  std::string err;
  if (!plan_dialect_c(max_min, max_min, scale, min_size, max_size, &job->levels, &err)) { fail(err); return -1; }
  const int nl = (int)job->levels.levels.size();
  if (nl > kMaxLevels) return 1;
  int pitch = (max_w + 15) & ~15;
  if ((pitch & 255) == 0) pitch += 16;            // keep rows of neighbouring tiles off one memory channel
  job->pitch = pitch;
  job->levels.width = pitch; job->levels.height = max_min;
  // nominal grids: the mean over the images that have the level (tile shapes are chosen for them)
  std::vector<double> sx(nl, 0.0), sy(nl, 0.0);
  std::vector<long long> cnt(nl, 0);
  job->n_lv.resize(job->n);
  for (int i = 0; i < job->n; i++) {
    const int k = ragged_levels_of(*job, job->widths[i], job->heights[i]);
    job->n_lv[i] = k;
    for (int l = 0; l < k; l++) {
      const Level& lv = job->levels.levels[l];
      sx[l] += (job->widths[i] - lv.win) / lv.step + 1; sy[l] += (job->heights[i] - lv.win) / lv.step + 1; cnt[l]++;
    }
  }
  unsigned long long h = 1469598103934665603ull;
  for (int l = 0; l < nl; l++) {
    Level& lv = job->levels.levels[l];
    // quantised, so that jobs over similar image sets share a plan
    // (rounded UP: a nominal grid one window narrower than the images' cuts every row of tiles in two)
    const int qx = cnt[l] ? std::max(1, (int)std::ceil(sx[l] / (double)cnt[l] / 4.0) * 4) : 1;
    const int qy = cnt[l] ? std::max(1, (int)std::ceil(sy[l] / (double)cnt[l] / 4.0) * 4) : 1;
    lv.nx = qx; lv.ny = qy; lv.base = 0;
    h = (h ^ (unsigned long long)(qx * 65536 + qy)) * 1099511628211ull;
  }
  job->levels.windows = 0;
  unsigned sb; std::memcpy(&sb, &scale, 4);
  PlanKey key{pitch, nl, 3 /* ragged, dialect C */, (int)sb, std::max(min_size, 24), max_size <= 0 ? -1 : max_size, h};
  std::lock_guard<std::mutex> lk(c->mu);
  if (!get_plan(c, key, job->levels, JDA_DIALECT_C, &job->pe, true)) return -1;      // (pinned; detect_ragged unpins)
  if (job->pe->dense_hint && !c->last_dense) job->pe->dense_hint = false;   // the per-image passes since then rejected most windows again
  if (!job->pe->fast_scan || job->pe->any_untiled || job->pe->dense_hint || c->kn.dense == 2) return 1;
  for (int l = 0; l < nl; l++) if (job->pe->hp.lv[l].tw * job->pe->hp.lv[l].th > 512) return 1;
  return 0;
}

// The level's tile re-cut for an image's own grid of nx x ny windows: as few tiles per row as the level's widest tile
// allows, evenly wide; the slack of a narrower tile goes into its height (up to 512 windows and the LDS the level's
// launch may use), again evenly.  An FDDB-sized image (77 x 64 windows of 46 pixels) gets 2 x 5 tiles of 39 x 13 windows
// (99 % of a 512-lane first phase) instead of 2 x 7 of 50 x 10 (60 %).
static void ragged_tile(const DevLevel& d, int nx, int ny, int th_lds, int* tw, int* th) {
  if (d.tiled == 2) th_lds = 512;                                  // global-pixel "tiles" are only window groups: no LDS limit
  const int tx = (nx + d.tw - 1) / d.tw;
  *tw = (nx + tx - 1) / tx;
  const int cap = std::max(1, std::min(th_lds, 512 / *tw));
  const int ty = (ny + cap - 1) / cap;
  *th = (ny + ty - 1) / ty;
}
// rows of windows a tile of level d may hold within lds_budget bytes of pixels
static int ragged_th_lds(const DevLevel& d, int pix_budget) {
  const int rows = pix_budget / std::max(1, d.pitch);
  return std::max(d.th, (rows - d.win) / std::max(1, d.step) + 1);
}

// Tables of images [i0, i0 + n) into the lane's pinned table buffer (and, for host images that do not lie back to
// back, the images into the lane's pinned staging buffer).
static bool ragged_build_chunk(Cascador* c, const RaggedJob& job, int i0, int n, Lane* ln, RaggedChunk* ch) {
  const DevPlan& hp = job.pe->hp;
  const int nl = hp.n_levels;
  ch->i0 = i0; ch->n = n; ch->pitch = job.pitch;
  ch->widths = job.widths + i0; ch->heights = job.heights + i0;
  ch->host_imgs = job.host_imgs ? job.host_imgs + i0 : nullptr;
  ch->d_raw = job.d_base;
  // ---- counts ----
  // How far a re-cut tile's pixels may outgrow the level's nominal tile (taller, narrower tiles for narrow images): as
  // far as the workgroups per CU stay what the nominal tile allows -- measured, a flat 1.4x took the 71/88-pixel levels
  // from 2 workgroups per CU to 1 and cost more than the fuller first phase gained.
  int th_lds[kMaxLevels];
  {
    const HostModel& hm = c->hm;
    const int chunk = std::min(std::min(hm.K, (int)c->kn.handoff), scan_handoff_cap(hm.node_n(), hm.leaf_n(), 4));
    for (int l = 0; l < nl; l++) {
      const DevLevel& d = hp.lv[l];
      if (d.tiled == 2) { th_lds[l] = d.th; continue; }
      const int nominal = d.pitch * (d.win + (d.th - 1) * d.step);
      const int fixed = (int)scan_lds_bytes(0, chunk, hm.node_n(), hm.leaf_n(), 4, false, d.tw * d.th > 256 ? 512 : 256);
      const int per_cu = std::max(1, (160 * 1024) / (fixed + nominal));
      const int room = (160 * 1024) / per_cu - fixed - 64;
      th_lds[l] = ragged_th_lds(d, std::max(nominal, std::min(room, nominal * (int)c->kn.ragged_tile_grow_pct / 100)));
    }
  }
  int n_segs = 0;
  long long n_blk = 0;
  for (int i = 0; i < n; i++) {
    const int W = job.widths[i0 + i], H = job.heights[i0 + i];
    n_segs += job.n_lv[i0 + i];
    for (int l = 0; l < job.n_lv[i0 + i]; l++) {
      const DevLevel& d = hp.lv[l];
      const int nx = (W - d.win) / d.step + 1, ny = (H - d.win) / d.step + 1;
      int tw, th;
      ragged_tile(d, nx, ny, th_lds[l], &tw, &th);
      n_blk += (long long)((nx + tw - 1) / tw) * ((ny + th - 1) / th);
    }
  }
  if (n_blk > 0x7fffffffLL) { fail("ragged chunk has too many tiles"); return false; }
  ch->n_segs = n_segs; ch->n_blk = (int)n_blk;
  size_t o = 0;
  ch->off_segs = o; o = align_up(o + (size_t)n_segs * sizeof(RagSeg), 256);
  ch->off_blk = o; o = align_up(o + (size_t)n_blk * sizeof(RagBlk), 256);
  ch->off_imgoff = o; o = align_up(o + (size_t)n * sizeof(unsigned long long), 256);
  ch->off_rimg = o; o = align_up(o + (size_t)n * sizeof(RagImg), 256);
  ch->table_bytes = o;
  if (!ln->h_tab.reserve(o) || !ln->rag_tab.reserve(o)) return false;
  uint8_t* tab = (uint8_t*)ln->h_tab.p;
  RagSeg* segs = (RagSeg*)(tab + ch->off_segs);
  RagBlk* blk = (RagBlk*)(tab + ch->off_blk);
  unsigned long long* img_off = (unsigned long long*)(tab + ch->off_imgoff);
  RagImg* rimg = (RagImg*)(tab + ch->off_rimg);
  // ---- images and segments ----
  ch->gid_base.assign(n + 1, 0);
  std::vector<int> seg_first(n + 1, 0);
  size_t dst = 0, src = 0;
  long long gid = 0;
  int max_h = 0, si = 0;
  bool contiguous = job.host_imgs != nullptr;
  for (int i = 0; i < n; i++) {
    const int W = job.widths[i0 + i], H = job.heights[i0 + i];
    max_h = std::max(max_h, H);
    img_off[i] = dst;
    rimg[i].dst_off = dst; rimg[i].w = W; rimg[i].h = H;
    if (job.host_imgs) {
      if (!job.host_imgs[i0 + i]) { fail("null image pointer"); return false; }
      if (i > 0 && job.host_imgs[i0 + i] != job.host_imgs[i0 + i - 1] + (size_t)job.widths[i0 + i - 1] * job.heights[i0 + i - 1]) contiguous = false;
      rimg[i].src_off = src;                         // tight, back to back in the staging copy
      src += (size_t)W * H;
    } else {
      rimg[i].src_off = job.d_offsets[i0 + i];
    }
    dst += align_up((size_t)H * job.pitch, 256);
    ch->gid_base[i] = (uint32_t)gid;
    seg_first[i] = si;
    for (int l = 0; l < job.n_lv[i0 + i]; l++) {
      const DevLevel& d = hp.lv[l];
      RagSeg& sg = segs[si++];
      sg.img_off = img_off[i]; sg.gid_base = (uint32_t)gid;
      sg.nx = (uint16_t)((W - d.win) / d.step + 1); sg.ny = (uint16_t)((H - d.win) / d.step + 1);
      int tw, th;
      ragged_tile(d, sg.nx, sg.ny, th_lds[l], &tw, &th);
      sg.tw = (uint16_t)tw; sg.th = (uint16_t)th;
      sg.tiles_x = (uint16_t)((sg.nx + tw - 1) / tw);
      sg.level = (uint16_t)l; sg.image = (uint16_t)i; sg.pad0 = 0; sg.pad1 = 0;
      sg.win = d.win; sg.step = d.step; sg.pitch = d.pitch; sg.s0_table = d.s0_table; sg.tiled = d.tiled;
      sg.pad2 = sg.pad3 = sg.pad4 = 0;
      gid += (long long)sg.nx * sg.ny;
    }
  }
  seg_first[n] = si;
  ch->gid_base[n] = (uint32_t)gid;
  if (gid > 0x7fffffffLL) { fail("ragged chunk has too many windows"); return false; }
  ch->windows = gid; ch->frame_bytes = dst + 256; ch->max_h = max_h;
  ch->raw_bytes = job.host_imgs ? src : 0;
  ch->host_contiguous = contiguous;
  if (!ln->rag_frames.reserve(ch->frame_bytes)) return false;
  if (job.host_imgs && !job.d_job_raw) {
    if (!ln->rag_raw.reserve(src + 16)) return false;
    if (!contiguous) {
      if (!ln->h_raw.reserve(src + 16)) return false;
      uint8_t* hr = (uint8_t*)ln->h_raw.p;
      for (int i = 0; i < n; i++) std::memcpy(hr + rimg[i].src_off, job.host_imgs[i0 + i], (size_t)rimg[i].w * rimg[i].h);
    }
  }
  // ---- block map and launches: one launch per LDS-tiled level (all of them in one when the chunk is small), one for
  //      the big-window LDS levels, one for the global-pixel levels.  Inside a launch the tiles of 8 images interleave,
  //      so that an image's tiles mostly land on one XCD's L2 (block b -> XCD b % 8). ----
  ch->launches.clear();
  int bi = 0;
  auto emit_group = [&](int l, int g0) {
    {
      int tiles[8], seg[8], most = 0;
      const int ge = std::min(n, g0 + 8);
      for (int i = g0; i < ge; i++) {
        tiles[i - g0] = 0; seg[i - g0] = -1;
        if (l < job.n_lv[i0 + i]) {
          const RagSeg& sg = segs[seg_first[i] + l];
          tiles[i - g0] = (int)sg.tiles_x * ((sg.ny + sg.th - 1) / sg.th);
          seg[i - g0] = seg_first[i] + l;
          most = std::max(most, tiles[i - g0]);
        }
      }
      for (int t = 0; t < most; t++)
        for (int j = 0; j < ge - g0; j++)
          if (t < tiles[j]) { blk[bi].seg = (uint32_t)seg[j]; blk[bi].tile = (uint32_t)t; bi++; }
    }
  };
  auto emit_level = [&](int l) { for (int g0 = 0; g0 < n; g0 += 8) emit_group(l, g0); };
  // pixel bytes / windows of the largest tile any image of the chunk cut from level l
  int th_max[kMaxLevels], win_max[kMaxLevels];
  for (int l = 0; l < nl; l++) { th_max[l] = 1; win_max[l] = 1; }
  long long lds_blocks = 0;
  for (int i = 0; i < n; i++)
    for (int l = 0; l < job.n_lv[i0 + i]; l++) {
      const RagSeg& sg = segs[seg_first[i] + l];
      th_max[l] = std::max<int>(th_max[l], sg.th); win_max[l] = std::max<int>(win_max[l], (int)sg.tw * sg.th);
      if (hp.lv[l].tiled == 1) lds_blocks += (long long)sg.tiles_x * ((sg.ny + sg.th - 1) / sg.th);
    }
  auto pix_of = [&](int l) { const DevLevel& d = hp.lv[l]; return d.pitch * (d.win + (th_max[l] - 1) * d.step); };
  const bool small = lds_blocks <= c->kn.merge_blocks;
  auto merged = [&](int mode) {
    RaggedChunk::Launch L{mode, 256, 0, bi, 0};
    // (group-major: the eight images of a group go through every level of the launch before the next group starts, so
    // that they stay in L2 from level to level -- level-major order cost the global-pixel launch 47 %)
    for (int g0 = 0; g0 < n; g0 += 8)
      for (int l = 0; l < nl; l++) if (hp.lv[l].tiled == mode) emit_group(l, g0);
    for (int l = 0; l < nl; l++) if (hp.lv[l].tiled == mode && mode != 2) L.pix_bytes = std::max(L.pix_bytes, pix_of(l));
    L.blk_n = bi - L.blk_base;
    if (L.blk_n > 0) ch->launches.push_back(L);
  };
  if (small) merged(1);
  else
    for (int l = 0; l < nl; l++)
      if (hp.lv[l].tiled == 1) {
        RaggedChunk::Launch L{1, win_max[l] > 256 ? 512 : 256, pix_of(l), bi, 0};
        emit_level(l);
        L.blk_n = bi - L.blk_base;
        if (L.blk_n > 0) ch->launches.push_back(L);
      }
  merged(3);
  merged(2);
  if (bi != ch->n_blk) { fail("internal: ragged block map size"); return false; }
  return true;
}

// NMS, relocation and the jdaResult of every image of a ragged chunk (dets sorted by gid = image, level, y, x).
static double post_ragged(Cascador* c, const RaggedJob& job, const RaggedChunk& ch, const RawDets<float>& dets,
                          const jdaDetectOptions* opt, jdaResult* out) {
  const double t0 = now_ms();
  const int L = c->hm.L, dim = c->hm.dim();
  const bool do_nms = !opt || opt->nms;
  const float overlap = opt ? opt->nms_overlap : 0.3f;
  const DevPlan& hp = job.pe->hp;
  std::vector<size_t> first(ch.n + 1, dets.gid.size());
  {
    size_t i = 0;
    for (int f = 0; f < ch.n; f++) {
      first[f] = i;
      while (i < dets.gid.size() && dets.gid[i] < ch.gid_base[f + 1]) i++;
    }
    first[ch.n] = i;
  }
  parallel_for(ch.n, [&](int f) {
    const size_t a = first[f], cnt = first[f + 1] - a;
    static thread_local std::vector<int> bb, keep;
    bb.resize(cnt * 3);
    const int W = ch.widths[f], H = ch.heights[f];
    int l = 0;
    uint32_t lbase = ch.gid_base[f];
    int nx = 0, cntl = 0;
    auto level_grid = [&](int lv) {
      const DevLevel& d = hp.lv[lv];
      nx = (W - d.win) / d.step + 1;
      cntl = nx * ((H - d.win) / d.step + 1);
    };
    if (cnt) level_grid(0);
    for (size_t i = 0; i < cnt; i++) {          // gids ascend: levels are walked once
      const uint32_t g = dets.gid[a + i];
      while (g >= lbase + (uint32_t)cntl) { lbase += (uint32_t)cntl; l++; level_grid(l); }
      const uint32_t rel = g - lbase;
      const DevLevel& d = hp.lv[l];
      bb[3 * i] = (int)(rel % (uint32_t)nx) * d.step; bb[3 * i + 1] = (int)(rel / (uint32_t)nx) * d.step; bb[3 * i + 2] = d.win;
    }
    if (do_nms) nms_dialect_c_into(bb.data(), &dets.score[a], (int)cnt, overlap, &keep);
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

static void add_stats(RunStats* a, const RunStats& b) {
  a->carts += b.carts; a->out += b.out; a->carts_scan += b.carts_scan; a->carts_scan_glb += b.carts_scan_glb;
  a->win_scan += b.win_scan; a->tail += b.tail; a->gpu_ms += b.gpu_ms; a->scan_ms += b.scan_ms;
  a->scan_launches += b.scan_launches; a->dense_passes += b.dense_passes;
  for (int t = 0; t < kMaxStages; t++) a->stage_done[t] += b.stage_done[t];
}

// A ragged job: images of different sizes, in host memory (host_imgs) or on the device (d_base + d_offsets).
static int detect_ragged(Cascador* c, const unsigned char* const* host_imgs, const uint8_t* d_base, const size_t* d_offsets,
                         const int* widths, const int* heights, int n, float scale, int min_size, int max_size, float th,
                         const jdaDetectOptions* opt, jdaResult* out) {
  const double t_call = now_ms();
  if (!c || !out || n < 0 || !widths || !heights || (!host_imgs && !(d_base && d_offsets))) { fail("bad arguments"); return -1; }
  const int L = c->hm.L;
  for (int i = 0; i < n; i++) { out[i].n = 0; out[i].landmark_n = L; out[i].bboxes = nullptr; out[i].shapes = nullptr; out[i].scores = nullptr; }
  if (n == 0) return 0;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!ensure_device(c) || !upload_model<float>(c)) return -1;
  }
  RunStats total;
  long long patch_n = 0;
  double post_ms = 0;
  jdaDetectOptions o1;
  jdaStats st1;
  auto finish = [&]() {
    fill_stats(opt ? opt->stats : nullptr, total, patch_n, c->hm.T, c->hm.K, post_ms);
    if (opt && opt->stats) opt->stats->call_ms = now_ms() - t_call;
    return 0;
  };
  // per-image passes: models the ragged scan does not cover (multi-scale split nodes, levels without a tile), and
  // cascades that reject so little that the dense kernel is the right tool
  auto fallback = [&]() -> int {
    for (int i = 0; i < n; i++) {
      if (opt) o1 = *opt; else jdaDetectOptionsInit(&o1);
      o1.stats = &st1; o1.hip_stream = nullptr;
      const int W = widths[i], H = heights[i];
      if (W <= 0 || H <= 0) { fail("image " + std::to_string(i) + " has no pixels"); return -1; }
      int rc;
      if (host_imgs) {
        const unsigned char* one[1] = {host_imgs[i]};
        rc = detect_c_device(c, nullptr, 0, 1, W, H, scale, min_size, max_size, th, &o1, out + i, one);
      } else {
        rc = detect_c_device(c, d_base + d_offsets[i], (size_t)W * H, 1, W, H, scale, min_size, max_size, th, &o1, out + i);
      }
      if (rc != 0) return -1;
      total.carts += st1.cart_total_n; total.out += st1.face_patch_n; total.carts_scan += st1.scan_cart_n;
      total.win_scan += st1.scan_patch_n; total.tail += st1.handoff_n; total.gpu_ms += st1.gpu_ms; total.scan_ms += st1.scan_ms;
      total.scan_launches += st1.scan_launches; total.dense_passes += st1.dense_passes;
      for (int t = 0; t < 16 && t < kMaxStages; t++) total.stage_done[t] += st1.stage_done_n[t];
      patch_n += st1.patch_n; post_ms += st1.host_ms;
    }
    return finish();
  };
  if (c->hm.multi_scale()) return fallback();
  RaggedJob job;
  job.n = n; job.widths = widths; job.heights = heights; job.host_imgs = host_imgs; job.d_base = d_base; job.d_offsets = d_offsets;
  const int prep = ragged_prepare(c, &job, scale, min_size, max_size);
  PlanPin pin{c, job.pe};
  if (prep < 0) return -1;
  if (prep > 0) return fallback();
  if (job.levels.levels.empty()) {                                // no image holds a window: n empty results
    for (int i = 0; i < n; i++) out[i] = empty_result(L);
    return finish();
  }

  // ---- chunks: as many images as make ragged_chunk_windows windows (<= 65535 images, the queues pack the index
  //      in 16 bits), walked through up to three lanes as a software pipeline: while the GPU works on chunks i-1 and i-2
  //      the host builds and issues chunk i and post-processes chunk i-3 ----
  const DevPlan& hp = job.pe->hp;
  // host images: uploaded by a helper thread, below (packed = every image right behind the one before in memory)
  bool helper = host_imgs != nullptr && c->kn.ragged_uploader != 0, packed = helper;
  std::vector<size_t> tight(helper ? (size_t)n + 1 : 0, 0);       // image i at tight[i] of the job's tight image buffer
  for (int i = 0; i < n && helper; i++) {
    if (!host_imgs[i] || widths[i] <= 0 || heights[i] <= 0) { helper = false; break; }
    if (i > 0 && host_imgs[i] != host_imgs[i - 1] + (size_t)widths[i - 1] * heights[i - 1]) packed = false;
    tight[i + 1] = tight[i] + (size_t)widths[i] * heights[i];
  }
  std::vector<int> starts;
  {
    long long wsum = 0; int cnt = 0;
    const long long full = std::max<long long>(1, c->kn.ragged_chunk_windows);
    long long target = full;
    starts.push_back(0);
    for (int i = 0; i < n; i++) {
      long long wi = 0;
      for (int l = 0; l < job.n_lv[i]; l++)
        wi += (long long)((widths[i] - hp.lv[l].win) / hp.lv[l].step + 1) * ((heights[i] - hp.lv[l].win) / hp.lv[l].step + 1);
      if (wi > 0x7fffffffLL) { fail("image has too many windows"); return -1; }
      // (a job whose pixels still have to come over the link starts with a quarter and a half chunk: the GPU has work
      // after a quarter of a chunk's upload time instead of a whole one)
      if (helper) target = starts.size() == 1 ? full / 4 : (starts.size() == 2 ? full / 2 : full);
      if (cnt > 0 && (wsum + wi > target || cnt >= 65535)) { starts.push_back(i); wsum = 0; cnt = 0; }
      wsum += wi; cnt++;
    }
    starts.push_back(n);
  }
  const int n_chunks = (int)starts.size() - 1;
  const int lanes = std::min(std::min(kRaggedLanes, n_chunks), (int)std::max<long long>(1, c->kn.max_lanes));
  struct Slot { bool busy = false; RaggedChunk ch; Pass<float> pass; RawDets<float> dets; RunStats rs; };
  std::vector<Slot> slots(lanes);
  LaneSet held(c);
  if (!held.take(lanes, n_chunks > 1 ? (size_t)c->kn.ragged_chunk_windows : 0, true)) return -1;
  bool ok = true;

  // ---- host images, several chunks: a helper thread brings chunk after chunk into a buffer of the job (the first
  //      lane's) on the cascador's upload stream and waits for each upload on the host; this thread builds tables,
  //      enqueues passes and post-processes meanwhile, and only waits for a chunk's pixels right before it enqueues that
  //      chunk.  (Issued here, every pageable upload blocked this thread for a millisecond, and staging 2,845 separate
  //      arrays with one memcpy loop took longer than the GPU needs for the job.)  Images that lie back to back go up
  //      straight from the caller's memory; separate arrays are gathered into two pinned buffers (the first two lanes')
  //      by `ragged_stage_threads` copy threads, chunk k+1 while chunk k is on the link. ----
  struct Uploader {
    std::thread th;
    std::mutex mu; std::condition_variable cv;
    int ready = 0;                 // chunks [0, ready) are on the device
    bool failed = false, stop = false;
    std::string err;
    ~Uploader() { { std::lock_guard<std::mutex> lk(mu); stop = true; } if (th.joinable()) th.join(); }
  } up;
  if (helper && n_chunks > 1 && lanes > 1 && held.v[0]->rag_raw.reserve(tight[n] + 16)) {
    size_t most = 0;
    job.raw_off.assign(n_chunks + 1, 0);
    for (int k = 0; k < n_chunks; k++) { job.raw_off[k + 1] = tight[starts[k + 1]]; most = std::max(most, job.raw_off[k + 1] - job.raw_off[k]); }
    if (packed || (held.v[0]->h_raw.reserve(most + 16) && held.v[1]->h_raw.reserve(most + 16))) {
      job.d_job_raw = (uint8_t*)held.v[0]->rag_raw.p;
      const int dev = c->device;
      const int copy_threads = (int)std::max<long long>(1, std::min<long long>(16, c->kn.ragged_stage_threads));
      uint8_t* stage[2] = {(uint8_t*)held.v[0]->h_raw.p, (uint8_t*)held.v[1]->h_raw.p};
      auto uploader = [&, dev, copy_threads, stage]() {
        bool good = hipSetDevice(dev) == hipSuccess;
        auto publish = [&](int k_done) {
          std::lock_guard<std::mutex> lk(up.mu);
          if (good) up.ready = k_done;
          else { up.failed = true; up.err = std::string("upload of a ragged chunk failed: ") + hipGetErrorString(hipGetLastError()); }
          up.cv.notify_all();
        };
        auto drain = [&]() {          // the uploads queued so far are on the device
          std::lock_guard<std::mutex> lk(c->h2d_mu);
          good = good && hipStreamSynchronize(c->h2d) == hipSuccess;
        };
        for (int k = 0; k < n_chunks && good; k++) {
          { std::lock_guard<std::mutex> lk(up.mu); if (up.stop) return; }
          const double t_up = now_ms();
          const size_t bytes = job.raw_off[k + 1] - job.raw_off[k];
          const uint8_t* src = host_imgs[0] + job.raw_off[k];
          if (!packed) {
            // gather the chunk's images into pinned buffer k & 1 (its last upload, chunk k-2, was drained one round ago)
            uint8_t* dst = stage[k & 1];
            const int a = starts[k], b = starts[k + 1];
            auto copy_range = [&](int i0, int i1) {
              for (int i = i0; i < i1; i++) std::memcpy(dst + (tight[i] - tight[a]), host_imgs[i], tight[i + 1] - tight[i]);
            };
            std::vector<std::thread> ts;
            int i0 = a;
            for (int t = 0; t < copy_threads && i0 < b; t++) {
              const size_t upto = tight[a] + bytes * (size_t)(t + 1) / (size_t)copy_threads;
              int i1 = i0;
              while (i1 < b && (tight[i1 + 1] <= upto || t == copy_threads - 1)) i1++;
              if (i1 == i0) continue;
              if (t == copy_threads - 1 || i1 == b) { copy_range(i0, b); i0 = b; }
              else {
                try { ts.emplace_back(copy_range, i0, i1); } catch (...) { copy_range(i0, i1); }   // (no thread to be had: copy here)
                i0 = i1;
              }
            }
            if (i0 < b) copy_range(i0, b);
            for (auto& t : ts) t.join();
            src = dst;
          }
          if (k > 0 && !packed) { drain(); publish(k); }           // chunk k-1 has arrived while this one was gathered
          {
            std::lock_guard<std::mutex> lk(c->h2d_mu);
            if (!c->h2d) good = hipStreamCreateWithFlags(&c->h2d, hipStreamNonBlocking) == hipSuccess;
            good = good && hipMemcpyAsync(job.d_job_raw + job.raw_off[k], src, bytes, hipMemcpyHostToDevice, c->h2d) == hipSuccess;
          }
          if (packed || k == n_chunks - 1) { drain(); publish(k + 1); }
          if (c->kn.debug_times) fprintf(stderr, "[jda] ragged upload %d: %.3f MB at %.3f..%.3f ms\n", k, bytes / 1e6, t_up - t_call, now_ms() - t_call);
        }
        if (!good) publish(0);
      };
      try { up.th = std::thread(uploader); }
      catch (...) { job.d_job_raw = nullptr; }          // (no thread to be had: the chunks upload themselves, as without a helper)
    }
  }
  auto collect = [&](Slot& sl) -> bool {
    Pass<float>& p = sl.pass;
    sl.busy = false;
    if (!p.after_tail() || !p.issue_counters() || !p.after_counters() || !p.collect()) return false;
    float ms_scan = 0, ms_all = 0;
    if (p.timed) {
      (void)hipEventElapsedTime(&ms_scan, p.ev[1], p.ev[2]);
      (void)hipEventElapsedTime(&ms_all, p.ev[0], p.ev[3]);
    }
    sl.rs.scan_ms += ms_scan; sl.rs.gpu_ms += ms_all;
    post_ms += post_ragged(c, job, sl.ch, sl.dets, opt, out + sl.ch.i0);
    add_stats(&total, sl.rs);
    patch_n += sl.ch.windows;
    return true;
  };
  for (int ci = 0; ci < n_chunks && ok; ci++) {
    const int lane = ci % lanes;
    Slot& sl = slots[lane];
    if (sl.busy && !collect(sl)) { ok = false; break; }
    sl.dets = RawDets<float>(); sl.rs = RunStats(); sl.rs.timed = opt && opt->stats;
    Lane* ln = held.v[lane];
    if (!ragged_build_chunk(c, job, starts[ci], starts[ci + 1] - starts[ci], ln, &sl.ch)) { ok = false; break; }
    if (sl.ch.windows == 0) {                        // images too small for any window
      post_ms += post_ragged(c, job, sl.ch, sl.dets, opt, out + sl.ch.i0);
      continue;
    }
    // workspace: every lane holds a whole chunk (its previous chunk has been collected above)
    {
      const size_t want = n_chunks > 1 ? std::max<size_t>((size_t)sl.ch.windows, (size_t)std::min<long long>(c->kn.ragged_chunk_windows, 0x7fffffffLL))
                                       : (size_t)sl.ch.windows;
      if (!ensure_workspace<float>(ln, want, false, c->hm.dim())) { ok = false; break; }
    }
    Pass<float>& p = sl.pass;
    p = Pass<float>();
    p.c = c; p.pe = job.pe; p.trace = nullptr; p.dets = &sl.dets; p.rs = &sl.rs; p.apply_th = true; p.th = th; p.multi = false;
    p.solo = lanes == 1;
    p.bind(ln, lane, nullptr);
    p.f0 = 0; p.nf = sl.ch.n; p.rag = &sl.ch;
    p.w.half = nullptr; p.w.quarter = nullptr; p.w.half_stride = p.w.quarter_stride = 0;
    p.w.hw = p.w.hh = p.w.qw = p.w.qh = 0;
    if (job.d_job_raw) {
      const double t_w = now_ms();
      std::unique_lock<std::mutex> lk(up.mu);
      up.cv.wait(lk, [&] { return up.ready > ci || up.failed; });
      if (c->kn.debug_times) fprintf(stderr, "[jda] ragged chunk %d: waited for its pixels %.3f..%.3f ms\n", ci, t_w - t_call, now_ms() - t_call);
      if (up.failed) { fail(up.err); ok = false; break; }
      sl.ch.d_uploaded = job.d_job_raw + job.raw_off[ci];
    }
    if (!p.issue_scan(nullptr, 0, nullptr, 0, nullptr)) { ok = false; break; }
    sl.busy = true;
  }
  if (c->kn.debug_times) fprintf(stderr, "[jda] ragged job: all chunks issued at %.3f ms\n", now_ms() - t_call);
  // drain in chunk order
  for (int k = 0; k < lanes && ok; k++) {
    Slot& sl = slots[(n_chunks + k) % lanes];
    if (sl.busy && !collect(sl)) ok = false;
  }
  if (!ok) {
    for (Lane* l : held.v) (void)hipStreamSynchronize(l->stream);
    return -1;
  }
  for (int i = 0; i < n; i++)
    if (!out[i].bboxes) out[i] = empty_result(L);     // (chunks fill every image; belt and braces)
  return finish();
}

}  // namespace jda

// =============================================================================
// C ABI
// =============================================================================

using namespace jda;

extern "C" {

const char* jdaGetLastError(void) { return g_err.c_str(); }

static void* create_impl(const char* path, int real_bytes) {
  g_err.clear();
  Cascador* c = new (std::nothrow) Cascador();
  if (!c) return nullptr;
  c->kn.load();
  std::string err;
  if (!load_model(path, real_bytes, &c->hm, &err)) {
    g_err = err;   // reference returns NULL silently (c/jda.c:487-488); keep the reason retrievable
    delete c;
    return nullptr;
  }
  return c;
}

void* jdaCascadorCreateDouble(const char* model) { return create_impl(model, 8); }
void* jdaCascadorCreateFloat(const char* model) { return create_impl(model, 4); }
void* jdaCascadorCreate(const char* model) { return create_impl(model, 0); }

void jdaCascadorSerializeTo(void* cascador, const char* model) {
  if (!cascador) return;
  (void)save_model_f32(((Cascador*)cascador)->hm, model);
}

void jdaCascadorRelease(void* cascador) {
  Cascador* c = (Cascador*)cascador;
  if (!c) return;
  // Submitted batches nobody waited for are drained here (their helper threads joined, their streams synchronised by
  // Lane::destroy).  A call still running on another thread is the caller's error, as with the reference, whose
  // release frees what jdaDetect reads (c/jda.c:718-720); such a call is given ten seconds to return before the
  // lanes go -- a race at shutdown then ends in a late but orderly release instead of a use-after-free.
  for (int i = 0; c->pending && i < kTickets; i++) c->pending[i].join_issuer();
  {
    std::unique_lock<std::mutex> lk(c->mu);
    for (int i = 0; c->pending && i < kTickets; i++)
      if (c->pending[i].active && c->pending[i].lane) { c->pending[i].lane->busy = false; c->pending[i].active = false; }
    c->lane_cv.wait_for(lk, std::chrono::seconds(10), [&]() {
      for (auto& l : c->lanes) if (l->busy) return false;
      return true;
    });
  }
  if (c->dev_init) {
    (void)hipSetDevice(c->device);
    for (auto& l : c->lanes) l->destroy();
    if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
    if (c->h2d) { (void)hipStreamSynchronize(c->h2d); (void)hipStreamDestroy(c->h2d); }
    for (auto& kv : c->plans) { if (kv.second.dp) (void)hipFree(kv.second.dp); if (kv.second.table) (void)hipFree(kv.second.table); }
    for (auto& b : c->plan_pool) { if (b.dp) (void)hipFree(b.dp); if (b.table) (void)hipFree(b.table); }
    c->mf.buf.release(); c->md.buf.release();
  }
  delete[] c->pending;
  delete c;
}

int jdaCascadorInfo(void* cascador, jdaModelInfo* info) {
  if (!cascador || !info) return -1;
  const HostModel& h = ((Cascador*)cascador)->hm;
  info->T = h.T; info->K = h.K; info->landmark_n = h.L; info->tree_depth = h.D;
  info->multi_scale = h.multi_scale() ? 1 : 0; info->source_real_bytes = h.real_bytes;
  return 0;
}

int jdaSetSimilarityTransform(void* cascador, int on) {
  Cascador* c = (Cascador*)cascador;
  if (!c) return -1;
  std::lock_guard<std::mutex> lock(c->mu);
  for (auto& l : c->lanes)
    if (l->busy) { fail("jdaSetSimilarityTransform while a call is running on this cascador"); return -1; }
  on = on ? 1 : 0;
  if (c->similarity != on) {
    c->similarity = on;
    c->md.ready = false;          // the fp64 node table depends on it (stage-0 offsets carry the transform)
  }
  return 0;
}

int jdaSetDevice(void* cascador, int device) {
  Cascador* c = (Cascador*)cascador;
  if (!c) return -1;
  std::lock_guard<std::mutex> lock(c->mu);
  if (c->dev_init && c->device != device) { fail("jdaSetDevice after first use"); return -1; }
  c->device = device;
  return 0;
}

int jdaSetOption(void* cascador, const char* key, long long value) {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !key) { fail("jdaSetOption: null cascador or key"); return -1; }
  std::lock_guard<std::mutex> lock(c->mu);
  for (auto& l : c->lanes)
    if (l->busy) { fail("jdaSetOption while a call is running or a submitted batch is pending on this cascador"); return -1; }
  for (auto& kv : c->plans)
    if (kv.second.pins) { fail("jdaSetOption while a call is running on this cascador"); return -1; }
  if (!c->kn.set(key, value)) { fail(std::string("jdaSetOption: unknown option or value out of range: '") + key + "'"); return -1; }
  // scan plans (tile shapes, table chunking) depend on the knobs: rebuild them on next use (no lane is busy, so
  // nothing runs on the old ones)
  for (auto& kv : c->plans) c->plan_pool.push_back({kv.second.dp, kv.second.table, kv.second.table_cap});
  c->plans.clear();
  return 0;
}

long long jdaGetOption(void* cascador, const char* key) {
  Cascador* c = (Cascador*)cascador;
  long long v = 0;
  if (!c || !key || !c->kn.get(key, &v)) { fail("jdaGetOption: unknown option"); return -1; }
  return v;
}

int jdaCountWindows(int width, int height, float scale, int min_size, int max_size,
                    long long* n_windows, int* n_levels) {
  ScanPlan sp; std::string err;
  if (!plan_dialect_c(width, height, scale, min_size, max_size, &sp, &err)) { g_err = err; return -1; }
  if (n_windows) *n_windows = sp.windows;
  if (n_levels) *n_levels = (int)sp.levels.size();
  return 0;
}

void jdaDetectOptionsInit(jdaDetectOptions* opt) {
  if (!opt) return;
  std::memset(opt, 0, sizeof(*opt));
  opt->dialect = JDA_DIALECT_C; opt->nms = 1; opt->nms_overlap = 0.3f; opt->cpp_step = 5;
}

int jdaDetectBatchDevice(void* cascador, const unsigned char* d_frames, size_t frame_stride, int n,
                         int width, int height, float scale, float step, int min_size, int max_size,
                         float th, const jdaDetectOptions* opt, jdaResult* out) {
  (void)step;  // ignored like the reference (c/jda.c:333)
  g_err.clear();
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchDevice runs dialect C; use jdaDetectBatchCpp"); return -1; }
  if (!cascador) { fail("null cascador"); return -1; }
  return detect_c_device((Cascador*)cascador, d_frames, frame_stride, n, width, height, scale, min_size, max_size, th, opt, out);
}

int jdaDetectBatchSubmit(void* cascador, const unsigned char* d_frames, size_t frame_stride, int n,
                         int width, int height, float scale, float step, int min_size, int max_size,
                         float th, const jdaDetectOptions* opt) {
  (void)step;
  g_err.clear();
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchSubmit runs dialect C"); return -1; }
  if (!cascador) { fail("null cascador"); return -1; }
  Cascador* c = (Cascador*)cascador;
  return submit_c_device(c, d_frames, frame_stride, n, width, height, scale, min_size, max_size, th, opt);
}

int jdaDetectBatchSubmitHost(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                             float scale, float step, int min_size, int max_size, float th,
                             const jdaDetectOptions* opt) {
  (void)step;
  g_err.clear();
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchSubmitHost runs dialect C"); return -1; }
  if (!cascador || !frames) { fail("null cascador or frames"); return -1; }
  if (width <= 0 || height <= 0) { fail("frame has no pixels"); return -1; }
  Cascador* c = (Cascador*)cascador;
  return submit_c_device(c, nullptr, 0, n, width, height, scale, min_size, max_size, th, opt, frames);
}

int jdaDetectBatchWait(void* cascador, int ticket, jdaStats* stats, jdaResult* out) {
  g_err.clear();
  if (!cascador) { fail("null cascador"); return -1; }
  Cascador* c = (Cascador*)cascador;
  return wait_c_device(c, ticket, stats, out);
}

int jdaDetectBatch(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                   float scale, float step, int min_size, int max_size, float th,
                   const jdaDetectOptions* opt, jdaResult* out) {
  (void)step;
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !frames || !out || n < 0) { fail("bad arguments"); return -1; }
  if (width <= 0 || height <= 0) { fail("frame has no pixels"); return -1; }
  for (int i = 0; i < n; i++) if (!frames[i]) { fail("null frame pointer"); return -1; }
  return detect_c_device(c, nullptr, 0, n, width, height, scale, min_size, max_size, th, opt, out, frames);
}

int jdaDetectBatchRagged(void* cascador, const unsigned char* const* images, const int* widths, const int* heights, int n,
                         float scale, float step, int min_size, int max_size, float th,
                         const jdaDetectOptions* opt, jdaResult* out) {
  (void)step;
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !images || !widths || !heights || !out || n < 0) { fail("bad arguments"); return -1; }
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchRagged runs dialect C"); return -1; }
  return detect_ragged(c, images, nullptr, nullptr, widths, heights, n, scale, min_size, max_size, th, opt, out);
}

int jdaDetectBatchRaggedDevice(void* cascador, const unsigned char* d_base, const size_t* offsets, const int* widths,
                               const int* heights, int n, float scale, float step, int min_size, int max_size, float th,
                               const jdaDetectOptions* opt, jdaResult* out) {
  (void)step;
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !d_base || !offsets || !widths || !heights || !out || n < 0) { fail("bad arguments"); return -1; }
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchRaggedDevice runs dialect C"); return -1; }
  return detect_ragged(c, nullptr, d_base, offsets, widths, heights, n, scale, min_size, max_size, th, opt, out);
}

jdaResult jdaDetect(void* cascador, unsigned char* data, int width, int height,
                    float scale, float step, int min_size, int max_size, float th) {
  Cascador* c = (Cascador*)cascador;
  jdaResult r;
  r.n = 0; r.landmark_n = c ? c->hm.L : 0; r.bboxes = nullptr; r.shapes = nullptr; r.scores = nullptr;
  if (!c || !data) { fail("jdaDetect: null cascador or image"); return empty_result(r.landmark_n); }
  const unsigned char* frames[1] = {data};
  if (jdaDetectBatch(cascador, frames, 1, width, height, scale, step, min_size, max_size, th, nullptr, &r) != 0) {
    jdaResultRelease(r);
    return empty_result(c->hm.L);
  }
  return r;
}

void jdaResultRelease(jdaResult result) {
  std::free(result.bboxes);
  std::free(result.shapes);
  std::free(result.scores);
}

int jdaTraceBatch(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                  float scale, int min_size, int max_size, int* carts_n, float* score,
                  unsigned int* path_hash, float* shapes) {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !frames || n < 0) { fail("bad arguments"); return -1; }
  ScanPlan sp; std::string err;
  if (!plan_dialect_c(width, height, scale, min_size, max_size, &sp, &err)) { fail(err); return -1; }
  unsigned sb; std::memcpy(&sb, &scale, 4);
  PlanKey key{width, height, JDA_DIALECT_C, (int)sb, std::max(min_size, 24), max_size <= 0 ? -1 : max_size, 0ull};
  PlanEntry* pe = nullptr;
  if (!begin_call<float>(c, key, sp, JDA_DIALECT_C, &pe)) return -1;
  PlanPin pin{c, pe};
  LaneSet lanes(c);
  size_t stride = 0;
  if (!lanes.take(1) || !stage_frames(lanes.v[0], frames, n, (size_t)width * height, &stride, true)) return -1;
  TraceOut<float> tr{carts_n, score, path_hash, shapes};
  RunStats rs;
  if (!run_device<float>(c, lanes, pe, (const uint8_t*)lanes.v[0]->frames.p, stride, n, false, 0.f, nullptr, nullptr, &tr, &rs,
                         HostFrames{frames, (size_t)width * height})) return -1;
  return 0;
}

int jdaBuildPyramid(void* cascador, const unsigned char* data, int width, int height,
                    unsigned char* half, int* hw, int* hh, unsigned char* quarter, int* qw, int* qh) {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !data || width <= 0 || height <= 0) { fail("bad arguments"); return -1; }
  const float r = 1.f / sqrtf(2.f);
  const int w1 = (int)((float)width * r), h1 = (int)((float)height * r), w2 = width / 2, h2 = height / 2;
  if (hw) *hw = w1; if (hh) *hh = h1; if (qw) *qw = w2; if (qh) *qh = h2;
  if (!half && !quarter) return 0;
  if (!begin_device(c)) return -1;
  LaneSet lanes(c);
  if (!lanes.take(1)) return -1;
  Lane* ln = lanes.v[0];
  const unsigned char* frames[1] = {data};
  size_t stride = 0;
  if (!stage_frames(ln, frames, 1, (size_t)width * height, &stride)) return -1;
  auto one = [&](unsigned char* dst, int dw, int dh) -> bool {
    if (!dst || dw < 1 || dh < 1) return true;
    if (!ln->pyr.reserve((size_t)dw * dh + 256)) return false;
    JDA_HIP(launch_resize((const uint8_t*)ln->frames.p, stride, 1, width, height, (uint8_t*)ln->pyr.p,
                          (size_t)dw * dh, dw, dh, (float)(width - 1) / dw, (float)(height - 1) / dh, ln->stream));
    JDA_HIP(hipMemcpyAsync(dst, ln->pyr.p, (size_t)dw * dh, hipMemcpyDeviceToHost, ln->stream));
    JDA_HIP(hipStreamSynchronize(ln->stream));
    return true;
  };
  if (!one(half, w1, h1) || !one(quarter, w2, h2)) return -1;
  return 0;
}

int jdaTraceBatchCpp(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                     int minimum_size, int step, double factor, int* carts_n, double* score,
                     unsigned int* path_hash, double* shapes) {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !frames || n < 0) { fail("bad arguments"); return -1; }
  if (!cpp_model_complete(c)) return -1;
  ScanPlan sp; std::string err;
  if (!plan_dialect_cpp(width, height, minimum_size, step, factor, &sp, &err)) { fail(err); return -1; }
  unsigned long long fb; std::memcpy(&fb, &factor, 8);
  PlanKey key{width, height, JDA_DIALECT_CPP, minimum_size, step, c->similarity, fb};
  PlanEntry* pe = nullptr;
  if (!begin_call<double>(c, key, sp, JDA_DIALECT_CPP, &pe)) return -1;
  PlanPin pin{c, pe};
  LaneSet lanes(c);
  size_t stride = 0;
  if (!lanes.take(1) || !stage_frames(lanes.v[0], frames, n, (size_t)width * height, &stride, true)) return -1;
  TraceOut<double> tr{carts_n, score, path_hash, shapes};
  RunStats rs;
  if (!run_device<double>(c, lanes, pe, (const uint8_t*)lanes.v[0]->frames.p, stride, n, false, 0.0, nullptr, nullptr, &tr, &rs,
                          HostFrames{frames, (size_t)width * height})) return -1;
  return 0;
}

int jdaResizeCv(void* cascador, const unsigned char* data, int width, int height, unsigned char* out, int ow, int oh) {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !data || !out || width <= 0 || height <= 0 || ow <= 0 || oh <= 0) { fail("bad arguments"); return -1; }
  if (!begin_device(c)) return -1;
  LaneSet lanes(c);
  if (!lanes.take(1)) return -1;
  Lane* ln = lanes.v[0];
  const unsigned char* frames[1] = {data};
  size_t stride = 0;
  if (!stage_frames(ln, frames, 1, (size_t)width * height, &stride)) return -1;
  auto run = [&]() -> bool {
    if (!ln->pyr.reserve((size_t)ow * oh + 256)) return false;
    JDA_HIP(launch_resize_cv((const uint8_t*)ln->frames.p, stride, 1, width, height, (uint8_t*)ln->pyr.p,
                             (size_t)ow * oh, ow, oh, ln->stream));
    JDA_HIP(hipMemcpyAsync(out, ln->pyr.p, (size_t)ow * oh, hipMemcpyDeviceToHost, ln->stream));
    JDA_HIP(hipStreamSynchronize(ln->stream));
    return true;
  };
  return run() ? 0 : -1;
}

int jdaDetectBatchCppPyramid(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                             int origin_size, int step, double factor, double overlap, int nms,
                             jdaStats* stats, jdaResultD* out) {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !frames || !out || n < 0) { fail("bad arguments"); return -1; }
  const int L = c->hm.L, dim = c->hm.dim();
  for (int i = 0; i < n; i++) { out[i].n = 0; out[i].landmark_n = L; out[i].rects = nullptr; out[i].shapes = nullptr; out[i].scores = nullptr; }
  if (origin_size < 1 || step < 1 || !(factor > 1.0)) { fail("origin_size/step must be positive and factor > 1"); return -1; }
  if (c->hm.multi_scale()) { fail("method 0 supports only scale==0 split nodes (its per-window half/quarter patches are not reproduced)"); return -1; }
  if (!cpp_model_complete(c)) return -1;
  if (!begin_device(c)) return -1;
  LaneSet lanes(c);
  if (!lanes.take(1)) return -1;
  Lane* ln = lanes.v[0];
  size_t stride0 = 0;
  if (!stage_frames(ln, frames, n, (size_t)width * height, &stride0)) return -1;

  // per level: rects (already scaled back), scores, normalised shapes, per frame, in scan order
  struct Cand { int rect[4]; double score; size_t shape_at; };
  std::vector<std::vector<Cand>> per_frame(n);
  std::vector<double> shape_pool;
  RunStats rs_total;
  long long patch_total = 0;
  // level images ping-pong inside one buffer; level 0 is the staged input
  DevBuf levels;
  const size_t lvl_stride = ((size_t)width * height + 255) & ~(size_t)255;
  auto body = [&]() -> bool {
    if (!levels.reserve(2 * lvl_stride * (size_t)std::max(n, 1))) return false;
    const uint8_t* cur = (const uint8_t*)ln->frames.p;
    size_t cur_stride = stride0;
    int w = width, h = height, li = 0;
    double scale = 1.;
    while (w >= origin_size && h >= origin_size) {               // cascador.cpp:283
      ScanPlan sp; std::string err;
      if (!plan_single_level(w, h, origin_size, step, &sp, &err)) { fail(err); return false; }
      PlanKey key{w, h, 2 /* method 0 level */, origin_size, step, c->similarity, 0ull};
      PlanEntry* pe = nullptr;
      if (!begin_call<double>(c, key, sp, JDA_DIALECT_CPP, &pe)) return false;
      PlanPin pin{c, pe};
      RawDets<double> dets;
      RunStats rs;
      if (!run_device<double>(c, lanes, pe, cur, cur_stride, n, false, 0.0, nullptr, &dets, nullptr, &rs)) return false;
      rs_total.carts += rs.carts; rs_total.out += rs.out; rs_total.gpu_ms += rs.gpu_ms; rs_total.scan_ms += rs.scan_ms;
      rs_total.carts_scan += rs.carts_scan; rs_total.win_scan += rs.win_scan; rs_total.scan_launches += rs.scan_launches;
      rs_total.tail += rs.tail;
      for (int t = 0; t < c->hm.T; t++) rs_total.stage_done[t] += rs.stage_done[t];
      patch_total += sp.windows * n;
      for (size_t i = 0; i < dets.gid.size(); i++) {
        const WinRef wr = locate(sp, dets.gid[i]);
        Cand cd;
        int rx = wr.x, ry = wr.y, rw = wr.win, rh = wr.win;
        rx = (int)(rx * scale); ry = (int)(ry * scale); rw = (int)(rw * scale); rh = (int)(rh * scale);   // cascador.cpp:292-294
        cd.rect[0] = rx; cd.rect[1] = ry; cd.rect[2] = rw; cd.rect[3] = rh;
        cd.score = dets.score[i];
        cd.shape_at = shape_pool.size();
        shape_pool.insert(shape_pool.end(), dets.shape.begin() + i * dim, dets.shape.begin() + (i + 1) * dim);
        per_frame[wr.frame].push_back(cd);
      }
      scale *= factor;                                            // cascador.cpp:299
      const int nw = (int)(w / factor), nh = (int)(h / factor);   // cascador.cpp:300-301
      if (nw < 1 || nh < 1) break;
      uint8_t* nxt = (uint8_t*)levels.p + (size_t)(li & 1) * lvl_stride * (size_t)n;
      JDA_HIP(launch_resize_cv(cur, cur_stride, n, w, h, nxt, lvl_stride, nw, nh, ln->stream));   // cascador.cpp:302
      JDA_HIP(hipStreamSynchronize(ln->stream));
      cur = nxt; cur_stride = lvl_stride; w = nw; h = nh; li++;
    }
    return true;
  };
  const bool ok = body();
  levels.release();
  if (!ok) return -1;

  const double t0 = now_ms();
  size_t total = 0;
  for (auto& v : per_frame) total += v.size();
  parallel_for(n, [&](int f) {
    const std::vector<Cand>& cs = per_frame[f];
    const size_t cnt = cs.size();
    std::vector<int> rc(cnt * 4);
    std::vector<double> sc(cnt);
    for (size_t i = 0; i < cnt; i++) { std::memcpy(&rc[4 * i], cs[i].rect, 16); sc[i] = cs[i].score; }
    std::vector<int> pick;
    if (nms) pick = nms_dialect_cpp(rc.data(), sc.data(), (int)cnt, overlap);
    else { pick.resize(cnt); std::iota(pick.begin(), pick.end(), 0); }
    jdaResultD& r = out[f];
    r.n = (int)pick.size(); r.landmark_n = L;
    r.rects = (int*)std::malloc(std::max<size_t>(1, pick.size() * 4) * sizeof(int));
    r.scores = (double*)std::malloc(std::max<size_t>(1, pick.size()) * sizeof(double));
    r.shapes = (double*)std::malloc(std::max<size_t>(1, pick.size() * dim) * sizeof(double));
    for (size_t i = 0; i < pick.size(); i++) {
      const int k = pick[i];
      std::memcpy(r.rects + 4 * i, &rc[4 * k], 4 * sizeof(int));
      r.scores[i] = sc[k];
      double* sh = r.shapes + i * dim;
      std::memcpy(sh, &shape_pool[cs[k].shape_at], dim * sizeof(double));
      relocate_dialect_cpp(sh, L, rc[4 * k], rc[4 * k + 1], rc[4 * k + 2], rc[4 * k + 3]);
    }
  }, total < 6000);
  fill_stats(stats, rs_total, patch_total, c->hm.T, c->hm.K, now_ms() - t0);
  return 0;
}

int jdaNmsC(const int* bboxes, const float* scores, int n, float overlap, unsigned char* keep) {
  if (n < 0 || (n > 0 && (!bboxes || !scores || !keep))) return -1;
  std::vector<int> k = nms_dialect_c(bboxes, scores, n, overlap);
  std::memset(keep, 0, (size_t)n);
  for (int i : k) keep[i] = 1;
  return (int)k.size();
}

int jdaNmsCpp(const int* rects, const double* scores, int n, double overlap, int* picked) {
  if (n < 0 || (n > 0 && (!rects || !scores || !picked))) return -1;
  std::vector<int> k = nms_dialect_cpp(rects, scores, n, overlap);
  std::copy(k.begin(), k.end(), picked);
  return (int)k.size();
}

int jdaResultsPack(const jdaResult* results, int n, int frame_offset, float* rows, int capacity_rows) {
  if (!results || n < 0) return -1;
  long long total = 0;
  for (int i = 0; i < n; i++) total += results[i].n;
  if (!rows) return (int)total;
  if (total > capacity_rows) return -1;
  float* o = rows;
  for (int i = 0; i < n; i++) {
    const jdaResult& r = results[i];
    const int dim = 2 * r.landmark_n;
    for (int j = 0; j < r.n; j++) {
      o[0] = (float)(frame_offset + i);
      o[1] = (float)r.bboxes[3 * j]; o[2] = (float)r.bboxes[3 * j + 1]; o[3] = (float)r.bboxes[3 * j + 2];
      o[4] = r.scores[j];
      std::memcpy(o + 5, r.shapes + (size_t)j * dim, dim * sizeof(float));
      o += 5 + dim;
    }
  }
  return (int)total;
}

void jdaResultsRelease(jdaResult* results, int n) {
  if (!results) return;
  for (int i = 0; i < n; i++) {
    std::free(results[i].bboxes); std::free(results[i].shapes); std::free(results[i].scores);
    results[i].bboxes = nullptr; results[i].shapes = nullptr; results[i].scores = nullptr; results[i].n = 0;
  }
}

// Tile plan of a dialect-C call without touching a device (tests, tools): per level 10 ints
// {win, step, nx, ny, mode, tw, th, pitch, tiles_x, tiles_y}.  Returns the number of levels.
int jdaDebugPlanTiles(void* cascador, int width, int height, float scale, int min_size, int max_size, int* out, int cap_levels) {
  Cascador* c = (Cascador*)cascador;
  if (!c) return -1;
  ScanPlan sp; std::string err;
  if (!plan_dialect_c(width, height, scale, min_size, max_size, &sp, &err)) { fail(err); return -1; }
  if ((int)sp.levels.size() > kMaxLevels) return -1;
  PlanEntry pe;
  bool s0_plain = true;
  const size_t n0 = (size_t)c->hm.K * c->hm.node_n();
  for (size_t i = 0; i < n0; i++) s0_plain = s0_plain && c->hm.nodes[i].scale == 0;
  assign_tiles(sp, c->hm, c->kn, s0_plain, 4, &pe);
  for (int i = 0; i < pe.hp.n_levels && i < cap_levels && out; i++) {
    const DevLevel& d = pe.hp.lv[i];
    const int v[10] = {d.win, d.step, d.nx, d.ny, d.tiled, d.tw, d.th, d.pitch, d.tiles_x, d.tiles_y};
    std::memcpy(out + 10 * i, v, sizeof v);
  }
  return pe.hp.n_levels;
}

long long jdaModelStreamBytes(int T, int K, int landmark_n, int tree_depth, int real_bytes) {
  return model_stream_bytes(T, K, landmark_n, tree_depth, real_bytes);
}

#ifdef JDA_SCAN_TIMING
// timing build only: shader-clock stamps of the k_scan workgroups of the last float pass
__attribute__((visibility("default"))) int jdaDebugScanTiming(void* cascador, unsigned long long* out) {
  Cascador* c = (Cascador*)cascador;
  if (!c || c->lanes.empty() || !c->lanes[0]->wf.dbg) return -1;
  return hipMemcpy(out, c->lanes[0]->wf.dbg, sizeof(unsigned long long) * 65536 * 32, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

void jdaResultDRelease(jdaResultD result) {
  std::free(result.rects);
  std::free(result.shapes);
  std::free(result.scores);
}

int jdaDetectBatchCpp(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                      int minimum_size, int step, double factor, double overlap, int nms,
                      jdaStats* stats, jdaResultD* out) {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !frames || !out || n < 0) { fail("bad arguments"); return -1; }
  const int L = c->hm.L, dim = c->hm.dim();
  for (int i = 0; i < n; i++) { out[i].n = 0; out[i].landmark_n = L; out[i].rects = nullptr; out[i].shapes = nullptr; out[i].scores = nullptr; }
  if (!cpp_model_complete(c)) return -1;
  ScanPlan sp; std::string err;
  if (!plan_dialect_cpp(width, height, minimum_size, step, factor, &sp, &err)) { fail(err); return -1; }
  unsigned long long fb; std::memcpy(&fb, &factor, 8);
  PlanKey key{width, height, JDA_DIALECT_CPP, minimum_size, step, c->similarity, fb};
  PlanEntry* pe = nullptr;
  if (!begin_call<double>(c, key, sp, JDA_DIALECT_CPP, &pe)) return -1;
  PlanPin pin{c, pe};
  LaneSet lanes(c);
  size_t stride = 0;
  if (!lanes.take(1) || !stage_frames(lanes.v[0], frames, n, (size_t)width * height, &stride, true)) return -1;
  RawDets<double> dets;
  RunStats rs;
  if (!run_device<double>(c, lanes, pe, (const uint8_t*)lanes.v[0]->frames.p, stride, n, false, 0.0, nullptr, &dets, nullptr, &rs,
                          HostFrames{frames, (size_t)width * height})) return -1;
  const double t0 = now_ms();
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
    std::vector<int> rc(cnt * 4);
    for (size_t i = 0; i < cnt; i++) {
      const WinRef wr = locate(sp, dets.gid[a + i]);
      rc[4 * i] = wr.x; rc[4 * i + 1] = wr.y; rc[4 * i + 2] = wr.win; rc[4 * i + 3] = wr.win;
    }
    std::vector<int> pick;
    if (nms) pick = nms_dialect_cpp(rc.data(), &dets.score[a], (int)cnt, overlap);
    else { pick.resize(cnt); std::iota(pick.begin(), pick.end(), 0); }
    jdaResultD& r = out[f];
    r.n = (int)pick.size(); r.landmark_n = L;
    r.rects = (int*)std::malloc(std::max<size_t>(1, pick.size() * 4) * sizeof(int));
    r.scores = (double*)std::malloc(std::max<size_t>(1, pick.size()) * sizeof(double));
    r.shapes = (double*)std::malloc(std::max<size_t>(1, pick.size() * dim) * sizeof(double));
    for (size_t i = 0; i < pick.size(); i++) {
      const int k = pick[i];
      std::memcpy(r.rects + 4 * i, &rc[4 * k], 4 * sizeof(int));
      r.scores[i] = dets.score[a + k];
      double* sh = r.shapes + i * dim;
      std::memcpy(sh, &dets.shape[(a + k) * dim], dim * sizeof(double));
      relocate_dialect_cpp(sh, L, rc[4 * k], rc[4 * k + 1], rc[4 * k + 2], rc[4 * k + 3]);
    }
  }, dets.gid.size() < 6000);
  fill_stats(stats, rs, sp.windows * n, c->hm.T, c->hm.K, now_ms() - t0);
  return 0;
}

}  // extern "C"
