// Device-side data layout and kernel launchers of the detect path (gfx950).
//
// Vocabulary (follows the reference): a *window* is one candidate (level,x,y);
// a *cart* is one depth-D tree of the cascade; a *stage* is K carts followed
// by one global shape regression.  A window's *gid* is
// frame*windows_per_frame + its scan-order index inside the frame.
//
// Two kernels carry the cascade (DESIGN.md "pipeline"):
//   k_scan    lane = window, LDS pixel tile, the first `handoff` carts of
//             stage 0, survivors compacted by ballot/prefix-sum
//   k_finish  wave = window: lanes = carts for the tree walks of a stage,
//             lanes = shape coordinates for the regression gather; runs a
//             survivor through every remaining cart, stage and the final cut
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace jda {

constexpr int kMaxLevels = 64;
constexpr int kMaxStages = 16;

struct DevLevel {
  int win, step, nx, ny;
  int base;              // first window of the level inside a frame (scan order)
  int tiled;             // k_scan pixel mode -- 1: LDS pixel tile, 16-bit node offsets; 3: LDS pixel tile, 21-bit
                         // node offsets (big windows, few per tile); 2: pixels through L1/L2 (windows that do
                         // not fit LDS); 0: not scanned, k_finish takes its windows from cart 0
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

// Ragged batches (images of different sizes in one pass, the FDDB loop of reference src/test.cpp:100-170 as ONE
// job): every image is staged with the same row pitch (DevPlan::width), the pyramid levels are a global list
// (DevPlan::lv: window size, step, tile shape, stage-0 table -- the window sizes of c/jda.c:331-333 do not depend on
// the image, an image only uses the prefix that fits it), and what does depend on the image sits in one record per
// (image, level).  A workgroup of k_scan finds its tile through RagBlk: blockIdx -> (segment, tile).
struct RagSeg {
  unsigned long long img_off;   // byte offset of the image inside WorkT::frames
  uint32_t gid_base;            // gid of this level's first window of this image (scan order: image, level, y, x)
  uint16_t nx, ny;              // windows per row / column (c/jda.c:335-336)
  uint16_t tiles_x;             // tiles per row of windows
  uint16_t level;               // index into DevPlan::lv
  uint16_t image;               // image index inside the pass (the queues pack it in 16 bits)
  uint16_t tw, th;              // windows per tile in x / y for THIS image: the level's tile (DevLevel::tw x th) re-cut to
                                // the image's own grid, inside the level's LDS row pitch (the node offsets depend on
                                // the pitch only)
  uint16_t pad0; uint32_t pad1;
  // the level's own record (copied from DevPlan::lv, so that a workgroup's prologue is blk -> seg and not
  // blk -> seg -> level: three dependent loads in front of the tile load cost the scan 15 % on identical geometry)
  int win, step, pitch, s0_table, tiled; uint32_t pad2, pad3, pad4;
};
static_assert(sizeof(RagSeg) == 64, "RagSeg is read as four 16-byte words");
struct RagBlk { uint32_t seg, tile; };
struct RagImg {                 // k_repack: one image of a ragged batch, tight rows -> common row pitch
  unsigned long long src_off, dst_off;
  int w, h;
};

// Split node as k_finish reads it: 32 bytes, two 16-byte loads (the
// reference's jdaNode is also 32 bytes, c/jda.c:114-127).
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

// k_finish walks 64 carts per wave, one cart per lane.  In cart-major order every lane's node sits in its own
// cache line (a cart's nodes are 7 x 32 bytes apart) and the two 16-byte halves of a node are two accesses; the
// L1 tag rate is what bounds k_finish (TA busy 70-75 %).  k_finish therefore reads a second copy of the nodes:
// level-major (the roots of all carts of a stage together, then the second level, ...) and split into the four
// offsets and the integer fields, so that the lanes of a wave read consecutive records.
template <typename Real>
struct NodeOff { Real o1x, o1y, o2x, o2y; };
// index of heap node `node` (on level d) of cart k inside one stage's K*node_n records
__host__ __device__ inline unsigned lm_index(unsigned K, unsigned k, unsigned d, unsigned node) {
  const unsigned first = (1u << d) - 1u;
  return K * first + (k << d) + (node - first);
}

// Levels d >= split of a tree with `levels` node levels: the nodes below one ancestor on level split - 1 form a group of
// lm_deep_group(levels, split) records in level order; the groups of a cart are consecutive, carts follow each other.
__host__ __device__ inline unsigned lm_deep_group(unsigned levels, unsigned split) { return (2u << (levels - split)) - 2u; }
__host__ __device__ inline unsigned lm_deep_index(unsigned k, unsigned d, unsigned node, unsigned levels, unsigned split) {
  const unsigned rel = node - ((1u << d) - 1u);               // position on its level
  const unsigned below = d - split + 1u;                      // levels between the ancestor and this node
  const unsigned anc = rel >> below, within = rel & ((1u << below) - 1u);
  return ((k << (split - 1u)) + anc) * lm_deep_group(levels, split) + ((1u << below) - 2u) + within;
}

// Stage-0 node with its pixel offsets resolved for one level (DESIGN.md
// "stage-0 hoist"): in stage 0 every window holds the mean shape, so the
// feature coordinates depend only on (node, window size).
// Packings of the same 8 bytes:
//   k_scan, tiled == 1    : lo = off1 | off2 << 16 (byte offsets inside the LDS tile), hi = th in [-256,255]
//   k_scan, tiled == 2, 3 : off1 : 21 | off2 : 21 | th + 256 : 10 (byte offsets inside the frame, row pitch = width,
//                           or inside a big LDS tile)
//   k_finish (level-major copy of the table, every level): x1 : 11 | y1 : 11 | x2 : 11 | y2 : 11 | th + 256 : 10,
//                           pixel coordinates inside the window (read from the frame or from the window's LDS copy)
struct S0Node {
  uint32_t lo;
  uint32_t hi;
};
constexpr int kS0GlobalOffBits = 21;

template <typename Real>
struct DevModelT {
  int T, K, L, D, node_n, leaf_n, dim;
  const void* nodes;       // NodeF / NodeD  [T*K*node_n]
  const void* lm_off;      // NodeOff<Real> [T*K*node_n], level-major (lm_index): k_finish's copy of the offsets
  const uint2* lm_meta;    // the same nodes' {lm1x2 | lm2x2 << 15 | scale << 30, th}
  const void* lm_deep;     // NodeF / NodeD of the levels from lm_split on, grouped per (cart, ancestor on level lm_split - 1) in level
  int lm_split;            // order (lm_deep_index): a deep tree's last levels of one path share a line or two instead of one line
                           // per level and array.  lm_split = D - 1: no such levels (depth <= 5), everything level-major
  const Real* leaf;        // [T*K*leaf_n]
  const Real* cth;         // [T*K]
  const Real* cmean;       // [T*K]
  const Real* cstd;        // [T*K]
  const uint8_t* cnorm;    // [T*K] 1 where (mean,std) != (0,1)
  const void* par0;        // {th, norm, mean, std} per cart [T*K], packed for LDS staging (k_scan, k_stage)
  const Real* w;           // [T][K*leaf_n][dim]
  const Real* w_rows;      // the same rows w_pitch elements apart, each starting on a 128-byte line (k_finish gathers one row
  int w_pitch;             // per cart: a tight 216-byte row straddles a third line in most positions); = w, dim when not padded
  int w_stream;            // a stage's weight rows are far larger than an XCD's L2: k_finish reads them with non-temporal loads, so that
                           // they pass through L2 without evicting the stage's split nodes (3 MB for K=2000, D=6)
  const Real* mean_shape;  // [dim]  (dialect CPP: + 0., the zero random shift of RandomShape)
  const Real* mean_shape_raw;  // [dim]  as stored (second argument of STParameter::Calc)
  int similarity;          // dialect CPP: Config::with_similarity_transform -- 0 off, 1 on, 2 + s: on, and stage s of a trainer snapshot is the one in training (it walks with the previous stage's parameter)
};

// Device buffers of one pass over a sub-batch of frames.
template <typename Real>
struct WorkT {
  const uint8_t* frames; size_t frame_stride; int n_frames;
  const uint8_t* half; size_t half_stride; int hw, hh;        // pyramid images, only for multi-scale models
  const uint8_t* quarter; size_t quarter_stride; int qw, qh;
  // dialect CPP, method 0 on a multi-scale model (cascador.cpp:243-245): no half / quarter IMAGE -- every window has its own
  // half_size^2 and quarter_size^2 patches, resized from its ROI: window i of frame f at half + f * half_stride + i * patch_hs^2
  // (quarter alike).  0: the images above
  int patch_hs, patch_qs;
  // hand-off queue k_scan -> k_finish (plus the windows k_scan does not cover)
  uint32_t* q_gid; Real* q_score; uint32_t* q_hash; uint32_t* q_kstart;
  uint32_t* q_xy; uint32_t* q_wf;          // x | y << 16 and win | frame << 16 of the queued window
  unsigned long long* counters;                                // see Counter
  // mid queue: windows alive after stage 0, with their regressed shape (k_finish pass 1 -> pass 2)
  uint32_t* m_gid; Real* m_score; uint32_t* m_hash; Real* m_shape; uint32_t* m_xy; uint32_t* m_wf;
  // dense mode (k_stage): per-window state indexed by gid -- m_score / m_hash / m_shape are reused as
  // score / hash / shape, st_carts holds carts evaluated (-1 = still alive)
  int* st_carts;
  // final detections: windows that passed every cart and the final threshold
  uint32_t* out_gid; Real* out_score; Real* out_shape;
  // per-window trace (all null when off), indexed by gid
  int* tr_carts; Real* tr_score; uint32_t* tr_hash; Real* tr_shape;
  unsigned cap;                                                // windows of the pass at most: entries of the per-window arrays (trace; dense mode's state)
  // The queues are sized from what earlier passes left in them, not for the worst case (r06): cap_q entries of the hand-off
  // queue (q_*), cap_m of the mid queue (m_*) and of the detection list (out_*).  A kernel never writes past them; the
  // counters keep counting, so the host sees an overflow as counter > capacity and runs the pass again with room
  // (Pass::recover_overflow).  Dense mode uses m_* as per-window state: the host gives it cap_m >= cap.
  unsigned cap_q, cap_m;
  // ragged batch (all null otherwise): (image, level) segments, the block map of the scan launches, image offsets
  const RagSeg* segs; const RagBlk* blk; const unsigned long long* img_off;
#ifdef JDA_SCAN_TIMING
  unsigned long long* dbg;                                     // [65536][16] shader-clock stamps of k_scan workgroups
#endif
#ifdef JDA_BOUNDS_CHECK
  const uint8_t* bc_lo; const uint8_t* bc_hi;                  // bounds-check build: the device range [lo, hi) the pass's frames occupy
#endif
};

// Work counters live in kCntShards copies, one 256-byte line apart, so that the
// workgroups of a launch do not serialise on one L2 atomic unit; the host sums
// the shards.  kCntTail / kCntOut are queue allocators and exist once (shard 0).
constexpr int kCntShards = 64;
constexpr int kCntStride = 32;   // 64-bit words per shard

enum Counter : int {
  kCntStage0 = 0,                  // kCntStage0 + t : windows that completed stage t
  kCntTail = kMaxStages,           // length of the hand-off queue
  kCntOut = kMaxStages + 1,        // final detections
  kCntCarts = kMaxStages + 2,      // carts evaluated, reference counting (Validate's n)
  kCntCartsScan = kMaxStages + 3,  // carts evaluated inside k_scan
  kCntWinScan = kMaxStages + 4,    // windows k_scan covered
  kCntMid = kMaxStages + 5,        // length of the mid queue (allocator, shard 0 only)
  kCntCartsScanGlb = kMaxStages + 6,  // carts evaluated inside k_scan's global-pixel launches
  kCntTotal = kMaxStages + 7,
  // spare words of a shard: [kCntTotal, kCntMidScan) deal k_scan_p's tiles (PScanCfg::dyn_slot, shards 0..7)
  kCntMidScan = kCntStride - 1,    // windows k_scan_p put into the mid queue itself (they count as handed off)
  kCntPostCursor = kCntTotal,      // shard 8 only: rows k_post has allotted (shards 0..7 use this word to deal tiles)
  kCntScanErr = kCntTotal          // shard 9 only: k_scan_p's watchdog word (bit 0: a wave gave up waiting for work that never came,
                                   // bit 1: a ring commit timed out) -- nonzero: the host runs the pass again with k_scan
};
constexpr int kCntScanErrShard = 9;
static_assert(kCntTotal <= kCntStride, "counter shard too small");

// ---- launchers (k_misc.hip, k_scan.hip, k_finish.hip, k_stage.hip) ------------------------------------------------

// Pyramid images of reference c/jda.c:450-457 for n frames.
hipError_t launch_resize(const uint8_t* src, size_t src_stride, int n, int sw, int sh,
                         uint8_t* dst, size_t dst_stride, int dw, int dh, float rx, float ry,
                         hipStream_t stream);

// cv::resize(INTER_LINEAR, 8UC1) restatement for dialect CPP (half/quarter images, method-0 pyramid).
hipError_t launch_resize_cv(const uint8_t* src, size_t src_stride, int n, int sw, int sh,
                            uint8_t* dst, size_t dst_stride, int dw, int dh, hipStream_t stream);

// The per-window patches of method 0 (cascador.cpp:243-245): cv::resize(INTER_LINEAR) of the win x win ROI of every window
// of a one-level plan (nx x ny windows, `step` apart, in frames of row pitch lw) to ds x ds, window i of frame f at
// dst + f * dst_stride + i * ds * ds.
hipError_t launch_resize_cv_patches(const uint8_t* src, size_t src_stride, int n, int lw, int nx, int ny, int step, int win,
                                    uint8_t* dst, size_t dst_stride, int ds, hipStream_t stream);

// Resolves stage-0 node offsets for every tiled level (dialect 0 = C, 1 = CPP): cart-major into table (k_scan),
// level-major (lm_index) into table_lm (k_finish).
hipError_t launch_prep_stage0(int dialect, const DevPlan* d_plan, const DevPlan& h_plan,
                              const void* nodes, const void* mean_shape, int K, int node_n,
                              S0Node* table, S0Node* table_lm, hipStream_t stream);

int scan_handoff_cap(int node_n, int leaf_n, int real_bytes);   // carts per LDS table chunk of k_scan
size_t scan_lds_bytes(int pix_bytes, int carts, int node_n, int leaf_n, int real_bytes, bool trace, int block);

// Windows of untiled levels (or of every level) -> head of the hand-off queue, k_start = 0.
template <typename Real>
hipError_t launch_enqueue(const DevPlan* d_plan, const DevPlan& h_plan, bool all_levels,
                          const WorkT<Real>& w, hipStream_t stream);

// Stage-0 scan, carts [0, handoff) of stage 0, for the levels of pixel mode `mode` (DevLevel::tiled):
// level >= 0 = that level; level < 0 = every level of the mode in one launch (small batches).
// cp_max = windows per tile at or below which a phase spreads (window, cart) pairs over the lanes.
// opts: bit 0 = 8 trees in flight per lane instead of 4; bits 8..15 = carts before the first compaction.
template <typename Real>
hipError_t launch_scan(int mode, int level, bool trace, int handoff, int cp_max, int opts, const DevPlan* d_plan,
                       const DevPlan& h_plan, const DevModelT<Real>& m, const S0Node* table,
                       const WorkT<Real>& w, hipStream_t stream, int lv_lo = 0, int lv_hi = kMaxLevels);
// (level < 0: ONE launch for every level of pixel mode `mode` among levels [lv_lo, lv_hi))

// The same for a ragged batch: one launch over blocks [blk_base, blk_base + blk_n) of WorkT::blk, all of pixel mode
// `mode` with workgroups of `block` threads and pixel tiles of at most pix_bytes.
template <typename Real>
hipError_t launch_scan_ragged(int mode, int block, bool trace, int handoff, int cp_max, int opts, const DevPlan* d_plan,
                              const DevModelT<Real>& m, const S0Node* table, const WorkT<Real>& w, int pix_bytes,
                              int blk_base, int blk_n, hipStream_t stream);

// The persistent form of the LDS-tiled scan (k_scan_p.hip; dialect C, pixel mode 1, uniform batches): one workgroup per
// CU walks its share of the level's tiles through `slots` pixel-tile slots; items that have completed bound[b] carts
// wait in ring b; lg[b] = 6: a bucket task is lane = window over 64 items, 4 / 5: pair task over 16 / 32 items.
// bound[nb] = carts done by this kernel (the hand-off).  opts bit 0 / 1: 8 trees in flight per lane in fresh / bucket tasks.
constexpr int kPScanMaxBuckets = 6;
struct PScanCfg {
  int nb;
  int bound[kPScanMaxBuckets + 1];
  int lg[kPScanMaxBuckets];
  int slots, slot_bytes, opts;
  int bound_last;                      // = bound[nb]
  int any_norm;                        // a cart of [0, bound_last) normalises its score (mean, std != 0, 1)
  int tw_magic;                        // ceil(2^20 / tile width in windows) where i / tw == (i * magic) >> 20 for every window index of a tile, else 0
  int ring_cap[kPScanMaxBuckets], ring_off[kPScanMaxBuckets], ring_items;     // items per ring / first item / all rings (scan_p_ring_caps)
  int th, tiles_y;                     // the kernel's own cut of the level in y: rows of windows per tile, tiles per column (the
                                       // row pitch and the tile width are the plan's: the resolved node offsets depend on them)
  int dyn_slot;                        // >= 0: tiles are dealt at run time from the counters' spare word kCntTotal + dyn_slot of shards 0..7 (zeroed with the counters); -1: fixed shares
  int to_mid;                          // bound_last == K: a window that passes every cart of stage 0 goes straight to the mid queue
};
void scan_p_ring_caps(PScanCfg* cfg, int waves);
size_t scan_p_lds_bytes(const PScanCfg& cfg, int carts, int node_n, int leaf_n, int waves);
// rag_blk_n >= 0: a ragged pass -- the launch covers blocks [rag_blk_base, rag_blk_base + rag_blk_n) of WorkT::blk, all of
// `level` (cfg.slot_bytes = the largest tile any image of the chunk cut from it; cfg.th / tiles_y / tw_magic unused).
hipError_t launch_scan_persistent(int level, const PScanCfg& cfg, int block, int grid_max, const DevPlan* d_plan,
                                  const DevPlan& h_plan, const DevModelT<float>& m, const S0Node* table,
                                  const WorkT<float>& w, hipStream_t stream, int rag_blk_base = 0, int rag_blk_n = -1);

// Tight images (row stride = width) at raw + src_off -> rows of `pitch` bytes at dst + dst_off, n images of at most
// max_h rows.
hipError_t launch_repack(const uint8_t* raw, uint8_t* dst, const RagImg* imgs, int n, int max_h, int pitch, hipStream_t stream);

// Stages [t_begin, t_end) for every queued window: t_begin == 0 reads the hand-off queue,
// t_begin > 0 the mid queue; survivors go to the mid queue (t_end < T) or the detection list.
template <typename Real>
hipError_t launch_finish(bool trace, int t_begin, int t_end, bool apply_final_th, Real final_th,
                         const DevPlan* d_plan, const DevModelT<Real>& m, const WorkT<Real>& w,
                         int groups, long long n_hint, const S0Node* s0_table, int tile_win, hipStream_t stream,
                         bool survivors = false);
// survivors: the input is the mid queue as launch_filter0 leaves it (windows that passed every cart of stage 0, still
// holding the mean shape); t_begin must be 0.

// The rest of stage 0 for every window of the hand-off queue, four windows per workgroup and nothing but the filtering
// (k_finish.hip: k_filter0): rejected windows are final, survivors go to the mid queue.  Needs the resolved stage-0
// table of every level.
template <typename Real>
hipError_t launch_filter0(bool trace, const DevPlan* d_plan, const DevModelT<Real>& m, const WorkT<Real>& w, long long n_hint,
                          const S0Node* s0_table, hipStream_t stream);
// s0_table: the plan's resolved stage-0 tables (or null): stage 0 of windows from levels that have one walks from it
// tile_win: windows up to this side copy their pixels to LDS first (0: every pixel is read from the frame,
// < 0: the largest side that leaves the workgroup within kFinishLdsPerGroup)
constexpr size_t kFinishLdsPerGroup = 7680;

// The finishing path of small jobs: one WORKGROUP per queued window (k_wide.hip), every remaining cart, stage and the
// final cut; reads the hand-off queue.  finish_wide_ok: the model fits (single-scale nodes, no similarity transform,
// a weight row within the LDS row buffer).
bool finish_wide_ok(int dim, int K, int leaf_n, int real_bytes, bool multi, bool similarity);
template <typename Real>
hipError_t launch_finish_wide(bool trace, bool apply_final_th, Real final_th, const DevPlan* d_plan,
                              const DevModelT<Real>& m, const WorkT<Real>& w, long long n_hint, const S0Node* s0_table,
                              hipStream_t stream, bool conc = true);
// conc: the stage's score replay (one wave) and its regression (a wave per 64 shape coordinates) run side by side

// Dense mode: stage t for every window of one level, a 16 x 8 tile of windows per workgroup.
// pix_cap = largest pixel tile that may live in LDS (larger windows read the frame through L1/L2).
template <typename Real>
hipError_t launch_stage(bool trace, int level, int t, bool apply_final_th, Real final_th, const DevPlan* d_plan,
                        const DevPlan& h_plan, const DevModelT<Real>& m, const WorkT<Real>& w, int pix_cap,
                        int lds_max, hipStream_t stream);
size_t stage_lds_bytes(int dim, int node_n, int leaf_n, int real_bytes);

// Device -> mapped pinned host memory by a kernel on `stream` (instead of the copy engine): up to 4 segments, 16-byte
// aligned.
hipError_t launch_copy_out(const void* const* src, void* const* dst, const size_t* bytes, int n, hipStream_t stream);

// Per-frame post-processing of a dialect-C pass on the device (k_post.hip): scan order, score order, NMS, relocation --
// one workgroup per frame, results into mapped pinned host memory.  n[f] = detections kept for frame f of the pass
// (-1: the kernel declined, flag[0] = 1: the host post-processes the pass from its raw detections), first[f] = their first
// row in bb (x, y, size) / score / shape (dim floats, relocated); rows are allotted from *cursor (device, zeroed
// with the counters).
struct PostOut {
  int* n; int* first; int* bb; float* score; float* shape; int* flag;
  unsigned long long* cursor;
  unsigned cap_rows;
};
// rag_gid / rag_img: a ragged pass (first gid of every image, n + 1 entries; the images' sizes), else null.
hipError_t launch_post(const DevPlan* d_plan, const WorkT<float>& w, int dim, int n_frames, bool do_nms, float overlap,
                       const PostOut& o, hipStream_t stream, const uint32_t* rag_gid = nullptr, const RagImg* rag_img = nullptr);

template <typename Real>
hipError_t launch_trace_fill(const DevModelT<Real>& m, const WorkT<Real>& w, unsigned n_windows,
                             hipStream_t stream);

// Hardware-queue probe (k_misc.hip): a one-wave spin of `ticks` wall-clock ticks (100 MHz) that leaves its end time in
// *out (mapped pinned host memory), and a kernel that leaves the time it ran.
hipError_t launch_hwq_spin(long long ticks, unsigned long long* out, hipStream_t stream);
hipError_t launch_hwq_stamp(unsigned long long* out, hipStream_t stream);

}  // namespace jda
