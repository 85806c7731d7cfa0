// Host side of libjda.so, shared by its translation units: error channel, options, device buffers, the cascador with
// its scan plans and lanes, and the types a pass over a batch works with.
//   lanes.cpp      device / lane / workspace management            plans.cpp    tile chooser, scan plans
//   model_dev.cpp  the model's device copies                       pass.h       one sub-batch through the device pipeline
//   run.h          a call's sub-batches over its lanes              detect.cpp   dialect-C batch entry
//   post_host.cpp  sort, NMS, relocation, jdaResult, statistics     tickets.cpp  submit / wait
//   ragged.cpp     images of different sizes as one job             abi.cpp      the extern "C" entry points of include/jda.h
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <map>
#include <mutex>
#include <numeric>
#include <new>
#include <stdexcept>
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

inline thread_local std::string g_err;     // (one per thread, whatever translation unit sets it)

inline void fail(const std::string& msg) {
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

inline double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// LDS of a gfx950 CU is handed out in granules of 1,280 bytes, 128 of them (160 KB): a workgroup of 54,272 bytes takes 43
// granules and only TWO of them fit, although 3 x 54,272 < 163,840 and hipOccupancyMaxActiveBlocksPerMultiprocessor says
// three (measured: tools/experiments/lds_occupancy.hip, profiles/r06_lds_granule.txt).
constexpr int kLdsGranule = 1280, kLdsGranules = 128;
inline int lds_wgs_per_cu(long long lds_bytes) { return lds_bytes <= 0 ? kLdsGranules : (int)(kLdsGranules / ((lds_bytes + kLdsGranule - 1) / kLdsGranule)); }
inline int lds_bytes_for_wgs(int wgs_per_cu) { return (kLdsGranules / std::max(1, wgs_per_cu)) * kLdsGranule; }   // the largest workgroup that still fits that many times

inline long long env_ll(const char* name, long long dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoll(v) : dflt;
}

// ---------------------------------------------------------------- knobs
// Tuning values of a cascador.  Read ONCE, when the cascador is created, from the JDA_* environment variables of
// DESIGN.md section 8 (experiments set them before jdaCascadorCreate*); jdaSetOption changes the documented ones
// afterwards.  Nothing on the call path touches the environment.
#define JDA_KNOBS(X)                                                                                   \
  X(handoff, "JDA_HANDOFF", 128)            /* carts of stage 0 k_scan evaluates before k_finish takes over */ \
  X(no_fast_scan, "JDA_NO_FAST_SCAN", 0)                                                               \
  X(plan_cache, "JDA_PLAN_CACHE", 64)       /* scan plans kept per cascador */                          \
  X(dense, "JDA_DENSE", 1)                  /* 0 off, 1 auto, 2 always */                              \
  X(merge_blocks, "JDA_MERGE_BLOCKS", 2048) /* workgroups below which the LDS-tiled levels share one launch */ \
  X(side_small, "JDA_SIDE_SMALL", 1)                                                                   \
  X(side_stream, "JDA_SIDE_STREAM", 1)                                                                 \
  X(wide_max, "JDA_WIDE_MAX", 1024)         /* ... and below which a window gets a whole workgroup (k_finish_wide) */ \
  X(wide_conc, "JDA_WIDE_CONC", 1)          /* ... in which a stage's score replay (wave 0) and its regression (waves 1..) run side by side (0: one after the other, rows through LDS) */ \
  X(h2d_stream, "JDA_H2D_STREAM", 1)        /* host frames go up on ONE stream per cascador, batch after batch, not lane by lane */ \
  X(h2d_min_bytes, "JDA_H2D_MIN_BYTES", 8 << 20) /* ... for uploads of at least this many bytes */ \
  X(kernel_d2h, "JDA_KERNEL_D2H", 1)        /* counters and detections -> pinned host memory by a kernel, not the copy engine */ \
  X(filter0, "JDA_FILTER0", 1)              /* large hand-off queues: k_filter0 + k_finish(survivors) instead of two k_finish passes */ \
  X(predict, "JDA_PREDICT", 1)              /* size the finishing launches from the previous pass (no host round trip) */ \
  X(debug_times, "JDA_DEBUG_TIMES", 0)                                                                 \
  X(test_wpf_scale, "JDA_TEST_WPF_SCALE", 1) /* test hook of the 32-bit window-id guard */             \
  X(test_throw, "JDA_TEST_THROW", 0)        /* test hook of the C ABI's exception barrier (abi.cpp): 1 = std::bad_alloc, 2 = std::runtime_error inside jdaDetectBatch */ \
  X(lanes, "JDA_LANES", 2)                  /* sub-batch lanes of one synchronous call */               \
  X(lanes_min_windows, "JDA_LANES_MIN_WINDOWS", 2000000)                                               \
  X(workspace_mb, "JDA_WORKSPACE_MB", 24 * 1024)                                                       \
  X(ws_bound, "JDA_WS_BOUND", 1)            /* queues of a pass sized from the fractions earlier passes left in them (0: for the worst case, every window in every queue) */ \
  X(ws_factor_pct, "JDA_WS_FACTOR_PCT", 400) /* ... times this safety factor, in percent */ \
  X(ws_min_entries, "JDA_WS_MIN_ENTRIES", 65536) /* ... and never fewer entries than this */ \
  X(ragged_chunk_windows, "JDA_RAGGED_CHUNK_WINDOWS", 4000000) /* windows per chunk of a ragged batch at most */ \
  X(ragged_chunk_windows_cpp, "JDA_RAGGED_CHUNK_WINDOWS_CPP", 8000000) /* ... of a dialect-CPP ragged batch */ \
  X(ragged_chunk_min_windows, "JDA_RAGGED_CHUNK_MIN_WINDOWS", 1500000) /* ... and at least, where a small job is cut into ragged_split chunks */ \
  X(ragged_split, "JDA_RAGGED_SPLIT", 3)    /* chunks a job smaller than that many full chunks is cut into */ \
  X(ragged_single_windows, "JDA_RAGGED_SINGLE_WINDOWS", 5000000) /* a ragged job of at most this many windows (a rank's shard of a sharded job) runs as ONE chunk, its global-pixel launch on the lane's side stream; 0: always cut into ragged_split chunks */ \
  X(hwq_place, "JDA_HWQ_PLACE", 1)          /* the cascador's streams are placed on the hardware queues by probing which of them share one (StreamPool); 0: where the runtime deals them */ \
  X(max_lanes, "JDA_MAX_LANES", 16)         /* lanes (stream + workspace + staging) a cascador creates at most; further concurrent callers wait for one */ \
  X(lane_idle_calls, "JDA_LANE_IDLE_CALLS", 256) /* lane hand-outs a free lane sits out before its workspace and staging buffers are released (0: never) */ \
  X(device_post, "JDA_DEVICE_POST", 1)      /* per-frame sort, NMS and relocation of a dialect-C batch on the device (k_post) instead of on the host (0): synchronous batch calls and tickets */ \
  X(device_post_min_frames, "JDA_DEVICE_POST_MIN_FRAMES", 16) /* ... for batches of at least this many frames */ \
  X(w_pad, "JDA_W_PAD", 1)                  /* k_finish gathers its weight rows from a copy whose rows start on 128-byte lines (0: from the tight table) */ \
  X(lm_deep, "JDA_LM_DEEP", 1)              /* trees of five or more node levels: k_finish reads the levels from the fourth on as whole records grouped per path (0: every level from the level-major split copy) */ \
  X(w_stream_mb, "JDA_W_STREAM_MB", 8)      /* ... with non-temporal loads when one stage's rows exceed this many MB (they would only push the stage's nodes out of L2); 0: never */ \
  X(scan_p, "JDA_SCAN_P", 1)                /* persistent scan kernel (k_scan_p): 0 off, 1 for the levels of large uniform batches it suits, 2 whenever it fits */ \
  X(scan_p_ragged, "JDA_SCAN_P_RAGGED", 1)  /* ... also for the single-level launches of a ragged chunk (tiles from the chunk's block map, re-cut per image) */ \
  X(scan_p_block, "JDA_SCAN_P_BLOCK", 768)  /* ... threads per workgroup */                             \
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
  X(scan_p_grid, "JDA_SCAN_P_GRID", 0)      /* ... workgroups of a launch (0: one per CU x scan_p_wgs) */ \
  X(scan_p_mid, "JDA_SCAN_P_MID", 1)        /* ... with scan_p_handoff >= K: windows that pass stage 0 go straight to the mid queue */

struct Knobs {
  // ---- former options, fixed at their measured values (r06 pruning): every A/B behind them is recorded as decided in
  //      profiles/DEAD_ENDS.md / DESIGN.md; they are compile-time constants now -- not settable, not read from the
  //      environment, the branches for other values fold away -- and keep their names where the code reads them ----
  static constexpr long long first_phase = 16;   // carts before k_scan's first compaction
  static constexpr long long cp_max = 128;   // windows per tile at or below which a phase spreads (window, cart) pairs
  static constexpr long long lds_win_max = 100;   // largest window that gets an LDS pixel tile
  static constexpr long long tile_cglb = 800;   // cost per window of the global-pixel mode (tile chooser)
  static constexpr long long glb_tile_fit = 1;
  static constexpr long long no_global_scan = 0;
  static constexpr long long no_lds_scan = 0;
  static constexpr long long debug_tiles = 0;
  static constexpr long long fin_s0 = 1;
  static constexpr long long dense_lds_max = 160 * 1024;
  static constexpr long long dense_pix = 16 * 1024;
  static constexpr long long dense_pct = 50;
  static constexpr long long side_after = 0;   // ... forked after this many LDS-tiled launches have been queued (the persistent scan takes its CUs first, the global-pixel workgroups fill what it leaves)
  static constexpr long long lanes_reverse = 1;
  static constexpr long long finish_merge = 4096;   // hand-off count below which one k_finish launch does all stages
  static constexpr long long wide_busy_max = 2;   // ... unless more than this many lanes of the cascador are in use
  static constexpr long long ragged_uploader = 1;   // ragged job from one packed host buffer: a helper thread uploads chunk after chunk
  static constexpr long long ragged_stage_threads = 4;   // ... and this many threads gather separate host arrays into its pinned buffers
  static constexpr long long fin_gm = 0;   // speculative 64-cart groups per k_finish round (0: from K)
  static constexpr long long fin_g1 = 1;
  static constexpr long long fin_g2 = 0;
  static constexpr long long fin_tile = -1;   // k_finish LDS window tile: -1 auto, 0 off, n pixels
  static constexpr long long fin_tile1 = 0;
  static constexpr long long fin_grid_div = 4;
  static constexpr long long host_chunk = 128;   // frames per sub-batch when the frames come from host memory
  static constexpr long long host_submit_thread = 1;
  static constexpr long long ragged_side = 1;   // ... (0: every launch of the single chunk on the lane's own stream)
  static constexpr long long ragged_lanes = 3;   // chunks of a ragged job in flight (lanes it takes), 1..8
  static constexpr long long ragged_merge = -1;   // LDS-tiled levels of a ragged chunk: one launch per occupancy class (1) or per level (0); -1: per class for dialect CPP, per level for dialect C (k_scan_p takes single-level launches)
  static constexpr long long ragged_tile_grow_pct = 150;   // pixel bytes of a re-cut tile, % of the level's nominal tile
  static constexpr long long scan_lean = 1;   // scan kernels without the per-cart test of the normalisation flag where no cart of the scanned range normalises
  static constexpr long long scan_p_min_slots = 4;   // ... pixel-tile slots a level's workgroup must have room for (scan_p = 1)
  static constexpr long long scan_p_wgs = 1;   // ... workgroups per CU
  static constexpr long long scan_p_lds_kb = 160;   // ... LDS a workgroup may take: what it leaves of the CU's 160 KB is where the other batch's kernels (global-pixel scan: 23.1 KB per workgroup, k_finish: 7.5 KB) find room next to it
  static constexpr long long scan_p_win_max = 100000;   // ... largest window of a level it takes
  static constexpr long long scan_p_dyn = 1;   // ... tiles dealt to the workgroups at run time (a workgroup that starts late takes fewer) instead of in fixed shares
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
    static const char* const non_negative[] = {"workspace_mb", "handoff", "plan_cache", "lanes", "ragged_chunk_windows", "ragged_chunk_windows_cpp",
                                               "ragged_chunk_min_windows", "h2d_min_bytes", "merge_blocks", "wide_max", "lanes_min_windows", "ragged_single_windows",
                                               "scan_p_handoff", "scan_p_slots", "max_lanes", "lane_idle_calls", "scan_p_tile_kb", "scan_p_grid",
                                               "ws_min_entries", "ws_factor_pct"};
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
inline bool stage0_any_norm(const HostModel& hm, int K, bool fp32) {
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
  bool building = false;        // its device tables are being made by the thread that inserted it (outside Cascador::mu): others wait on plan_cv
  bool failed = false;          // ... and that failed: waiters give up, the last one removes the entry
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

// The streams of a cascador, placed on the device's hardware queues (lanes.cpp).
// The HIP runtime deals the streams of a process to GPU_MAX_HW_QUEUES (four) hardware queues -- the queue that has the
// fewest streams so far, the last such -- and the packets of one queue run strictly one after the other.  Where a new
// stream lands therefore depends on every stream the process has created before, and lanes that land on one queue do not
// overlap at all: the same FDDB-sized ragged job takes 7.4 or 8.7 ms (dialect CPP: 30.1 or 35.7 ms) depending on how
// many streams the host program happened to create first (profiles/r06_hwq.txt).  The pool finds out which of its
// streams share a queue by PROBING (k_hwq_spin on the streams it knows, k_hwq_stamp on the new one) and hands them out
// by queue: the main streams of the lanes spread over the queues, a lane's side stream and the upload stream on queues
// that carry as few main streams as possible (never the lane's own).  Streams that landed where nothing was needed stay
// in the pool for later requests.  hwq_place = 0: plain hipStreamCreate, the runtime's deal.
struct StreamPool {
  enum Role { kMain, kSide, kAux };
  static constexpr int kNone = -2;           // "no queue to keep away from"; class -1 = a queue of the stream's own
  struct Item { hipStream_t s; int cls; bool used; Role role; };
  std::mutex mu;
  std::vector<Item> items;
  std::vector<hipStream_t> rep;              // the first stream seen on every known queue
  std::vector<int> mains, others;            // per queue: main streams / side and upload streams handed out
  unsigned long long* stamps = nullptr;      // mapped pinned memory the probes write to, kSlots x kRow
  unsigned probe_no = 0;
  int dry = 0;                               // streams created in a row that found no new queue (4: all queues known)
  bool place = true;                         // hwq_place
  int hw_queues = 4;                         // hardware queues the runtime deals streams to (its GPU_MAX_HW_QUEUES)
  int created = 0, probes = 0;               // (jdaGetOption "hwq_streams" / "hwq_probes")
  static constexpr int kMaxCls = 8, kRow = 16, kSlots = 32;
  hipStream_t take(Role role, int avoid, int* cls_out);
  void give_back(hipStream_t s);
  void destroy();
 private:
  bool create_one(int* cls);
  int classify(hipStream_t s);
  int best_class(Role role, int avoid) const;
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
  StreamPool* pool_ = nullptr;               // the cascador's: where both streams come from
  int q_main = -2, q_side = -2;              // hardware-queue class of either stream (StreamPool)
  hipEvent_t ev[5] = {};
  hipEvent_t ev_side[2] = {};
  hipEvent_t ev_user = nullptr;
  hipEvent_t ev_h2d[2] = {};                 // staging buffer free / frames uploaded (Cascador::h2d)
  unsigned long long* h_cnt = nullptr;       // pinned copy of the work counters
  HostPinned h_gid, h_score, h_shape;        // detections of the lane's pass
  HostPinned h_pn, h_pbb, h_psc, h_psh;      // ... post-processed on the device (k_post): per-frame count / first row (+ flag), boxes, scores, shapes
  DevBuf ws;                                 // per-window arrays, carved for one dialect at a time
  size_t cap = 0; bool trace = false; int dim = 0, real_bytes = 0;
  size_t cap_q = 0, cap_m = 0; bool dense_ws = false;   // entries of the hand-off queue / of the mid queue and the detection list; k_stage's per-window state is there
  WorkT<float> wf{};
  WorkT<double> wd{};
  DevBuf frames;                             // staging of host frames (the call's first lane holds the whole batch)
  DevBuf pyr;                                // half + quarter images (multi-scale models), method-0 levels
  // ragged passes: images at the common pitch, tight images, tables (segments, block map, image records)
  DevBuf rag_frames, rag_raw, rag_tab;
  HostPinned h_tab, h_raw;
  bool create(StreamPool* pool) {
    pool_ = pool;
    if (!pool_ || !(stream = pool_->take(StreamPool::kMain, StreamPool::kNone, &q_main))) return false;
    for (auto& e : ev) JDA_HIP(hipEventCreate(&e));
    JDA_HIP(hipEventCreateWithFlags(&ev_user, hipEventDisableTiming));
    for (auto& e : ev_h2d) JDA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    JDA_HIP(hipHostMalloc((void**)&h_cnt, sizeof(unsigned long long) * kCntShards * kCntStride, hipHostMallocDefault));
    return true;
  }
  bool ensure_side() {
    if (side) return true;
    // a hardware queue that is not the lane's own and, where there is one, carries no other lane's main stream either
    // (StreamPool).  (A side stream of another PRIORITY was tried in r06: the headline lost 8 %, 1.41 -> 1.53 ms per
    // step, the shard job gained nothing: profiles/DEAD_ENDS.md)
    if (!(side = pool_->take(StreamPool::kSide, q_main, &q_side))) return false;
    for (auto& e : ev_side) JDA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return true;
  }
  // The memory of a lane nobody has used for a while (a burst of concurrent callers leaves lanes behind, each with a
  // workspace of up to workspace_mb): everything that is re-created on demand.  The lane is free and its holder has
  // collected what ran on it, so nothing is in flight.
  // With a bag, the buffers are only MOVED out (pointer swaps: the caller holds Cascador::mu) and released when the
  // bag goes out of scope, after the mutex -- hipFree / hipHostFree of gigabytes synchronise the device and take
  // milliseconds, in which every other caller would stand in front of the plan cache and the lane pool.
  struct Bag {
    std::vector<void*> dev, host;
    Bag() = default;
    Bag(const Bag&) = delete;
    Bag& operator=(const Bag&) = delete;
    ~Bag() { for (void* p : dev) (void)hipFree(p); for (void* p : host) (void)hipHostFree(p); }
  };
  void trim(Bag* bag = nullptr) {
    DevBuf* db[] = {&ws, &frames, &pyr, &rag_frames, &rag_raw, &rag_tab};
    HostPinned* hb[] = {&h_gid, &h_score, &h_shape, &h_tab, &h_raw, &h_pn, &h_pbb, &h_psc, &h_psh};
    for (DevBuf* b : db) { if (bag && b->p) { bag->dev.push_back(b->p); b->p = nullptr; b->bytes = 0; } else b->release(); }
    for (HostPinned* b : hb) { if (bag && b->p) { bag->host.push_back(b->p); b->p = nullptr; b->bytes = 0; } else b->release(); }
    cap = 0; cap_q = 0; cap_m = 0; trace = false; dense_ws = false; dim = 0; real_bytes = 0;
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
    if (pool_) { pool_->give_back(stream); pool_->give_back(side); }     // (the pool destroys its streams with the cascador)
    stream = side = nullptr;
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
  StreamPool streams;          // every stream of the cascador, placed on the hardware queues
  hipStream_t aux = nullptr;   // stage-0 table builds (under mu)
  // Frame uploads of every lane, in the order they are issued (under h2d_mu).  Uploads issued lane by lane run
  // CONCURRENTLY on the copy engines, each at a fraction of the link: two batches then both arrive late, and their
  // kernels collide afterwards.  One after the other, batch i+1 goes up while batch i computes.
  hipStream_t h2d = nullptr;
  std::mutex h2d_mu;
  std::vector<std::unique_ptr<Lane>> lanes;
  std::condition_variable plan_cv;           // a plan's device tables are ready (or failed), with mu
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

// ---------------------------------------------------------------- device init, lanes (lanes.cpp)

// Makes the cascador's device current for the calling thread; first use picks the device (caller holds c->mu then).
bool ensure_device(Cascador* c);
// A free lane (caller holds c->mu), see lanes.cpp.
Lane* acquire_lane_locked(Cascador* c, size_t want_cap, bool* exhausted, Lane::Bag* trimmed);

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
    Lane::Bag trimmed;                     // (declared before the lock: released after it, see Lane::trim)
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
      Lane* l = acquire_lane_locked(c, want_cap, &exhausted, &trimmed);
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
    // unwinding (an exception on its way to the C ABI's barrier): what the call has queued on its lanes is waited for
    // HERE, while the lanes are still this call's -- once they are back in the pool another caller may take them
    if (std::uncaught_exceptions() > 0)
      for (Lane* l : v) { if (l->stream) (void)hipStreamSynchronize(l->stream); if (l->side) (void)hipStreamSynchronize(l->side); }
    { std::lock_guard<std::mutex> lk(c->mu); for (Lane* l : v) l->busy = false; }
    c->lane_cv.notify_all();
  }
};

// The model of dialect Real on the device (model_dev.cpp; caller holds c->mu).
template <typename Real> bool upload_model(Cascador* c);

// ---------------------------------------------------------------- scan plans (plans.cpp)

// The plan of (frame size, call parameters), built on first use.  The caller holds c->mu through `lk`; a miss builds the
// plan's device tables with the lock RELEASED (allocation, a blocking copy, a kernel and a stream wait: concurrent
// callers on other frame sizes -- a per-image loop over differently sized images from several threads -- go on meanwhile,
// callers on the same key wait for it).  The plan comes back PINNED (PlanEntry::pins): it is not evicted -- its device
// tables are not recycled -- until unpin_plan.
bool get_plan(Cascador* c, std::unique_lock<std::mutex>& lk, const PlanKey& key, const ScanPlan& sp, int dialect, PlanEntry** out,
              bool ragged = false);
void unpin_plan(Cascador* c, PlanEntry* pe);
// How k_scan covers every level of a plan (tile shapes, pixel modes, offsets of the stage-0 tables); ragged: see plans.cpp.
void assign_tiles(const ScanPlan& sp, const HostModel& hm, const Knobs& kn, bool fast_scan, int real_bytes, PlanEntry* pe,
                  bool ragged = false);

// ---------------------------------------------------------------- workspace (lanes.cpp)

template <typename Real> size_t bytes_per_window(int dim, bool trace);
// The lane's arrays for a pass over `cap` windows of dialect Real (grow-only; the lane is idle): cap_q entries of the
// hand-off queue, cap_m of the mid queue and the detection list (0: cap, the worst case), dense: k_stage's per-window state.
template <typename Real> bool ensure_workspace(Lane* ln, size_t cap, bool trace, int dim, size_t cap_q = 0, size_t cap_m = 0, bool dense = true);
template <typename Real> size_t workspace_bytes(size_t cap, size_t cap_q, size_t cap_m, bool trace, bool dense, int dim);

// How many entries the queues of a pass over `windows` windows get (r06; before: every queue held every window).  From the
// fractions earlier passes on the plan left in them (PlanEntry::pred_*, negative: none yet) times ws_factor_pct, never
// below ws_min_entries; without a prediction an eighth / a thirty-second of the windows.  A pass that outgrows them is
// noticed by its counters and run again with room (Pass::recover_overflow): bounding costs time in that case, never results.
struct QueueCaps { size_t q, m; };
inline QueueCaps queue_caps(const Knobs& kn, size_t windows, double pred_tail, double pred_mid, double pred_out, bool full) {
  if (full || kn.ws_bound == 0) return {windows, windows};
  const double f = (double)std::max<long long>(100, kn.ws_factor_pct) / 100.0;
  const size_t floor_n = (size_t)std::max<long long>(1, kn.ws_min_entries);
  const double fq = pred_tail >= 0 ? pred_tail * f : 0.125;
  const double pm = std::max(pred_mid, pred_out);
  const double fm = pm >= 0 ? pm * f : 1.0 / 32.0;
  auto clampn = [&](double frac) {
    const double v = frac * (double)windows + 64.0;
    return std::min(windows, std::max(floor_n, v >= (double)windows ? windows : (size_t)v));
  };
  return {clampn(fq), clampn(fm)};
}

// ---------------------------------------------------------------- the pipeline

template <typename Real>
struct RawDets {               // survivors of a batch, sorted by gid (= frame, then scan order)
  std::vector<uint32_t> gid;
  std::vector<Real> score;
  std::vector<Real> shape;     // [n][dim]
  // frames whose pass was post-processed on the device (k_post; dialect C only): p_n[f] >= 0 detections kept after NMS,
  // already relocated, rows [p_first[f], p_first[f] + p_n[f]) of p_bb (x, y, size) / p_sc / p_sh; p_n[f] < 0: the frame's
  // detections are in the raw lists above.  Empty unless the caller asked for it (HostFrames::device_post)
  std::vector<int> p_n, p_first, p_bb;
  std::vector<Real> p_sc, p_sh;
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
  int ws_regrows = 0;          // passes run again because a queue sized from earlier passes was too small (Pass::recover_overflow)
  int scan_fallbacks = 0;      // passes run again with k_scan because k_scan_p's launch tripped a watchdog / came back short
};


// Copies a run of host frames to the staging buffer (frame i at dst + i*stride) on a stream (lanes.cpp).
bool copy_frames_h2d(uint8_t* dst, size_t stride, const unsigned char* const* frames, int n, size_t fbytes, hipStream_t st);

// ---- ragged batches: images of different sizes in one pass (kernels.h: RagSeg) ----
// Host tables of one chunk of a ragged job: what its pass uploads and launches.
struct RaggedChunk {
  int i0 = 0, n = 0;                    // images [i0, i0 + n) of the job
  long long windows = 0;                // candidate windows of the chunk
  size_t frame_bytes = 0;               // staged images (common row pitch)
  size_t raw_bytes = 0;                 // tight images (host staging; 0 when the images are already on the device)
  int max_h = 0, pitch = 0;
  std::vector<uint32_t> gid_base;       // [n + 1] first gid of every image inside the pass
  struct Launch { int mode, block, pix_bytes, blk_base, blk_n; int level = -1; };   // level: the one level of the launch (-1: several, merged)
  std::vector<Launch> launches;
  int n_segs = 0, n_blk = 0;
  size_t off_segs = 0, off_blk = 0, off_imgoff = 0, off_rimg = 0, off_gidbase = 0, table_bytes = 0;   // layout of the table buffer
  size_t images_bytes = 0;              // ... whose first bytes are the image records (RagImg, image offsets, gid ranges)
  bool images_issued = false;           // the image records' upload and k_repack are already on the lane's stream (device-resident
                                        // images: queued before the block map was built, ragged.cpp); the pass uploads the rest
  const unsigned char* const* host_imgs = nullptr;   // the chunk's images in host memory (tight), or
  const uint8_t* d_raw = nullptr;                    // the base their RagImg::src_off refer to on the device
  const int* widths = nullptr; const int* heights = nullptr;      // of the chunk's images
  bool host_contiguous = false;         // the host images lie back to back in memory in RagImg::src_off order
  const uint8_t* d_uploaded = nullptr;  // the chunk's tight images are (being) uploaded here by the job's helper thread
};

}  // namespace jda
