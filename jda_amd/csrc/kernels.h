// Device-side data layout and kernel launchers of the detect path (gfx950).
//
// Vocabulary (follows the reference): a *window* is one candidate (level,x,y);
// a *cart* is one depth-D tree of the cascade; a *stage* is K carts followed
// by one global shape regression.  A window's *gid* is
// frame*windows_per_frame + its scan-order index inside the frame.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace jda {

constexpr int kMaxLevels = 64;
constexpr int kMaxStages = 16;

// How the stage-0 scan covers one pyramid level.
enum TileClass : int {
  kTileWide = 0,    // 256-thread workgroups, up to 512 windows share one LDS tile
  kTileNarrow = 1,  // 64-thread workgroups (one wave), up to 64 windows per LDS tile
  kTileNone = 2     // window too large for an LDS tile: generic walker reads HBM/L2
};

struct DevLevel {
  int win, step, nx, ny;
  int base;              // first window of the level inside a frame (scan order)
  int tile_class;        // TileClass
  int tw, th;            // windows per tile in x / y
  int tiles_x, tiles_y;
  int pitch;             // LDS bytes per tile row
  int s0_table;          // first entry of this level in the stage-0 offset table (units: nodes)
};

struct DevPlan {
  int n_levels;
  int width, height;
  int windows;            // per frame
  DevLevel lv[kMaxLevels];
};

// Split node as the generic walker reads it: 32 bytes, two 16-byte loads
// (the reference's jdaNode is also 32 bytes, c/jda.c:114-127).
struct NodeF {
  int scale;
  int lm1x2, lm2x2;   // landmark index * 2, like c/jda.c:521-523
  int th;
  float o1x, o1y, o2x, o2y;
};
struct NodeD {         // dialect CPP: fp64 offsets (already passed through the identity STParameter)
  int scale;
  int lm1x2, lm2x2;
  int th;
  double o1x, o1y, o2x, o2y;
};

// Stage-0 node with its pixel offsets resolved for one level (see DESIGN.md
// "stage-0 hoist"): in stage 0 every window holds the mean shape, so the
// feature coordinates depend only on (node, window size).
struct S0Node {
  uint32_t offs;   // off1 | off2 << 16, byte offsets from the window origin inside the LDS tile
  int32_t th;      // feature threshold clamped to [-256, 255]
};

template <typename Real>
struct DevModelT {
  int T, K, L, D, node_n, leaf_n, dim;
  const void* nodes;       // NodeF / NodeD  [T*K*node_n]
  const Real* leaf;        // [T*K*leaf_n]
  const Real* cth;         // [T*K]
  const Real* cmean;       // [T*K]
  const Real* cstd;        // [T*K]
  const uint8_t* cnorm;    // [T*K] 1 where (mean,std) != (0,1)
  const Real* w;           // [T][K*leaf_n][dim]
  const Real* mean_shape;  // [dim]
};

// One pipeline's device buffers for a sub-batch of frames.
template <typename Real>
struct WorkT {
  // frames
  const uint8_t* frames; size_t frame_stride; int n_frames;
  const uint8_t* half; size_t half_stride; int hw, hh;        // pyramid, only for multi-scale models
  const uint8_t* quarter; size_t quarter_stride; int qw, qh;
  // survivor queues, ping-pong by stage parity
  uint32_t* q_gid[2]; Real* q_score[2]; uint32_t* q_src[2]; uint32_t* q_hash[2];
  Real* shape[2];          // [cap][dim]
  // queue of windows the stage-0 scan does not cover (generic walker, t = 0)
  uint32_t* qg_gid;
  // counters (device): see Counter enum
  unsigned long long* counters;
  // slots (in queue T-1) of the windows that pass the final threshold
  uint32_t* out_slot;
  // trace (optional, all NULL when off)
  int* tr_carts; Real* tr_score; uint32_t* tr_hash; Real* tr_shape;
  unsigned cap;            // queue capacity (windows)
};

enum Counter : int {
  kCntQueue0 = 0,          // kCntQueue0 + t : windows that passed every cart of stage t
  kCntGeneric = kMaxStages,        // windows queued for the generic stage-0 walker
  kCntOut = kMaxStages + 1,        // final detections
  kCntCarts = kMaxStages + 2,      // carts evaluated (reference counting)
  kCntCartsScan = kMaxStages + 3,  // the part of kCntCarts evaluated by the LDS-tiled stage-0 scan
  kCntWinScan = kMaxStages + 4,    // windows the stage-0 scan covered
  kCntTotal = kMaxStages + 5
};

// ---- launchers (kernels.hip) ------------------------------------------------

// Pyramid images of reference c/jda.c:450-457 for n frames.
hipError_t launch_resize(const uint8_t* src, size_t src_stride, int n, int sw, int sh,
                         uint8_t* dst, size_t dst_stride, int dw, int dh, float rx, float ry,
                         hipStream_t stream);

// Resolves stage-0 node offsets for every tiled level (dialect 0 = C, 1 = CPP).
hipError_t launch_prep_stage0(int dialect, const DevPlan* d_plan, const DevPlan& h_plan,
                              const void* nodes, const void* mean_shape, int K, int node_n,
                              S0Node* table, hipStream_t stream);

// Table/LDS sizing shared by host planner and kernels.
int scan_chunk_max(int node_n, int leaf_n);       // carts per LDS table chunk
size_t scan_lds_bytes(int pix_bytes, int node_n, int leaf_n, int real_bytes, bool trace, int tile_class);

// Stage-0 scan of ONE tiled level (h_plan.lv[level].tile_class is wide or narrow).
template <typename Real>
hipError_t launch_scan(int level, bool trace, const DevPlan* d_plan, const DevPlan& h_plan,
                       const DevModelT<Real>& m, const S0Node* table, const WorkT<Real>& w,
                       hipStream_t stream);

template <typename Real>
hipError_t launch_enqueue_generic(const DevPlan* d_plan, const DevPlan& h_plan, bool all_levels,
                                  const WorkT<Real>& w, hipStream_t stream);

// Generic walker for stage t: reads queue `in_counter`, appends survivors to queue kCntQueue0+t.
template <typename Real>
hipError_t launch_walk(int dialect, bool trace, int t, const DevPlan* d_plan, const DevModelT<Real>& m,
                       const WorkT<Real>& w, hipStream_t stream);

// Stage regression for the survivors of stage t (+ final threshold/emit when t == T-1).
template <typename Real>
hipError_t launch_update(int dialect, bool trace, int t, bool apply_final_th, Real final_th,
                         const DevPlan* d_plan, const DevModelT<Real>& m, const WorkT<Real>& w,
                         hipStream_t stream);

template <typename Real>
hipError_t launch_pack(const WorkT<Real>& w, int T, int dim, hipStream_t stream);

template <typename Real>
hipError_t launch_trace_fill(const DevModelT<Real>& m, const WorkT<Real>& w, unsigned n_windows,
                             hipStream_t stream);

}  // namespace jda
