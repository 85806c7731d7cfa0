// k_scan_p: the first `handoff` carts of stage 0 as ONE PERSISTENT workgroup per CU whose waves run free of each other
// (no workgroup barrier after the prologue).  Replaces, for large uniform batches in dialect C, k_scan's closed
// tile-per-workgroup form, where every tile went 500 -> 286 -> 160 -> 47 -> 3 live windows behind phase barriers while
// its workgroup kept 8 waves and 46 KB of LDS (stamps: half of a workgroup's clocks for the last third of its windows).
// Reference loop being replaced: c/jda.c:357-402 (stage 0; the offsets are resolved per (node, level) by k_prep_stage0).
//
//   * the workgroup owns S pixel-tile SLOTS in LDS and walks its share of the level's tiles through them; a wave that
//     finds a free slot claims it, brings the next tile in by LDS-DMA and publishes it
//   * work is cut into wave TASKS, taken by whichever wave is free:
//       fresh   64 consecutive windows of a published tile, carts [0, bound[0]), lane = window
//       bucket  items (window, score) that have completed bound[b] carts wait in ring b (LDS) -- items of ALL resident
//               tiles together, so that deep cart ranges still find full waves.  A bucket task takes 64 of them
//               (lane = window, carts [bound[b], bound[b+1])) or, in the pair form, 16 / 32 of them with the
//               (window, 8 carts) pairs spread over the lanes and the scores replayed in cart order from the leaf
//               indices in a per-wave LDS scratch (k_scan's pair phases, per wave instead of per workgroup)
//     survivors go to the next ring, or to the hand-off queue after the last range; every item holds a reference on
//     its slot, and the slot is free again when its last window has died or has been handed off
//   * rings are multi-producer / multi-consumer: producers reserve with one LDS atomic, write, and commit IN ORDER
//     (a short spin on the commit counter); consumers take with a compare-and-swap on the pop counter.  A wave prefers
//     the deepest ring that holds a full task, then fresh windows, then whatever is left (partial tasks: only when a
//     tile is late or at the end of the launch), so a ring never holds more than a task plus what the waves in flight
//     can add -- the ring capacity 64 * (waves + 2)
// Same arithmetic in the same order as k_scan: bit-identical results (reject cart, score).
#include "kernels_common.h"
#include "scan_walk.h"

namespace jda {

namespace {

constexpr int kPSlotsMax = 8;          // slot index: 3 bits of an item
constexpr int kPWidxBits = 11;         // window index inside the tile: up to 2048 windows per tile
constexpr int kPBaseBits = 18;         // LDS byte offset of the window's first pixel
// Watchdogs of the two wait loops (a wait is microseconds; these are a large fraction of a second).  Neither loop can
// hang the device, and neither ends quietly: a trip sets a bit of the counters' error word (kCntScanErr) and the host,
// which also compares the windows covered with the plan's, runs the pass again with k_scan (pass.h: recover_scan).
// A wave that trips never publishes anything half done: an idle wave just leaves; a ring commit that times out is NOT
// committed out of order (its items are lost to the launch, later reservations time out behind it, the slots they
// reference never drain and the remaining waves leave through the idle watchdog) -- so a tripped launch loses
// windows, it never hands a corrupt item to the finishing kernels.
constexpr int kPSpinMax = 1 << 20;     // ring commit
#ifndef JDA_SCAN_P_IDLE_MAX
#define JDA_SCAN_P_IDLE_MAX (1 << 20)  // (tests build a variant with 0: every wave that idles once leaves -> the fallback runs)
#endif
constexpr int kPIdleMax = JDA_SCAN_P_IDLE_MAX;

struct ThNormF { float th, norm; };

// Control block in LDS: 16-byte records, so that a wave picks its next task from ONE ds_read_b128 -- lane i reads
// record i, evaluates "its" slot or ring, and ballots turn the candidates into masks.  (The first version walked rings
// and slots with dependent reads and scalar branches: ~400 instructions per pick.  A wave issues an instruction every
// 5-10 clocks at best, so a pick cost 4 k clocks and -- with lost compare-and-swaps repeating it -- 75 % of all wave
// clocks, although the LDS round trip itself is ~60 clocks: stamps in profiles/r04_scan_p_stamps.txt.)
typedef int v4i __attribute__((ext_vector_type(4)));
struct PCtl {
  v4i rec[32];
  //   rec[s], s < 8        slot:  x claim generation << 2 | state (0 free, 1 being loaded, 2 published), y generation << 12 | next
  //                               fresh batch, z batches, w references (fresh batches not finished + items alive).  The state
  //                               word carries a generation so that a claim's compare-and-swap, made on a snapshot that has
  //                               gone stale, cannot succeed on a slot another wave has claimed AND republished meanwhile
  //   rec[8 + b], b < 6    ring:  x committed, y popped, z reserved
  //   rec[14]              x next tile (sequence number inside this workgroup's share) to bring in
  //   rec[16 + s]          slot geometry: x twe | the << 16, y xshift, z frame, w wx0 | wy0 << 16
  //   rec[24 + s]          ragged passes: the tile's image -- x gid of its level's first window, y windows per row (nx),
  //                        z windows per tile row (tw: the level's tile re-cut for this image)
  int cfgw[64];                        // PScanCfg as words (see kCf*): a wave keeps word i in lane i and reads it with v_readlane
  unsigned long long dbg[20];
};
constexpr int kRecRing = 8, kRecMisc = 14, kRecGeo = 16, kRecGeo2 = 24;
// (dynamic indexing of the by-value kernel argument is a scalar memory load per access, ~200 clocks each and serialised)
constexpr int kCfNb = 0, kCfBound = 1, kCfLg = kCfBound + kPScanMaxBuckets + 1, kCfCap = kCfLg + kPScanMaxBuckets,
              kCfOff = kCfCap + kPScanMaxBuckets, kCfEnd = kCfOff + kPScanMaxBuckets;
static_assert(kCfEnd <= 64, "cfg words fit a wave");

struct PLds {
  int nodes, leaf, par, ctl, rings, lfbuf, slots, total;
  __host__ __device__ PLds(int carts, int node_n, int leaf_n, int ring_items, int waves, int n_slots, int slot_bytes) {
    int o = 0;
    nodes = o; o += carts * node_n * (int)sizeof(S0Node); o = (o + 15) & ~15;
    leaf = o; o += carts * leaf_n * 4; o = (o + 15) & ~15;
    par = o; o += carts * (int)sizeof(CartPar<float>); o = (o + 15) & ~15;
    ctl = o; o += (int)sizeof(PCtl); o = (o + 15) & ~15;
    rings = o; o += ring_items * 8; o = (o + 15) & ~15;
    lfbuf = o; o += waves * 512;
    slots = o; o += n_slots * slot_bytes;
    total = o;
  }
};

__device__ __forceinline__ int ld_relaxed(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_relaxed(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// every earlier LDS operation of this wave has completed (and the compiler moves no memory access across)
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// global -> LDS by LDS-DMA with a run-time workgroup size (dma_to_lds, kernels_common.h, takes it as a template
// parameter); the tail and unaligned sources go through registers.  The caller waits vmcnt(0) + barrier.
__device__ __forceinline__ void dma_to_lds_rt(unsigned char* lds_dst, const void* __restrict__ src, int nbytes, int tid, int nthreads) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  const unsigned char* g = (const unsigned char*)src;
  int done = 0;
  if ((((uintptr_t)g) & 15) == 0) {
    const int chunks = nbytes >> 4;
    const int lane = tid & 63, wv = tid >> 6;
    for (int base = wv * 64; base < chunks; base += nthreads) {
      if (base + lane < chunks)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + ((size_t)(base + lane) << 4)), (lds_ptr_t)(lds_dst + (base << 4)), 16, 0, 0);
    }
    done = chunks << 4;
  }
  for (int i = done + tid; i < nbytes; i += nthreads) lds_dst[i] = g[i];
}

struct PCtx {
  Bc bc;                               // bounds-check build: the LDS bytes of the pixel-tile slots (empty otherwise)
  const uint8_t* lds;                  // pixels are addressed by absolute LDS offsets
  const S0Node* t_nodes; const float* t_leaf; const CartPar<float>* t_par;
  int node_n, leaf_n, D;
};

// carts [k, k+CNT) applied to this lane's window, strictly in cart order with the per-cart reject (c/jda.c:395-399)
// NORM = false: no cart of this launch normalises its score (known on the host) -- the per-cart test of the flag and
// its branch leave the loop.
template <int CNT, bool NORM>
__device__ __forceinline__ void p_apply(const PCtx& c, int k, const int* lf, bool& alive, float& score, unsigned& my_carts) {
  ThNormF p[CNT];
  float lsv[CNT];
#pragma unroll
  for (int u = 0; u < CNT; u++) { p[u] = *(const ThNormF*)&c.t_par[k + u]; lsv[u] = c.t_leaf[(k + u) * c.leaf_n + lf[u]]; }
  float s = score;
  bool dead = false;
  int kd = k;
#pragma unroll
  for (int u = 0; u < CNT; u++) {
    if (!dead) {
      s = s + lsv[u];                                                                                  // c/jda.c:396
      if (NORM && p[u].norm != 0.f) { const CartPar<float> q = c.t_par[k + u]; s = (s - q.mean) / q.std; }   // c/jda.c:397 (rare)
      kd = k + u;
      dead = s < p[u].th;                                                                              // c/jda.c:399
    }
  }
  score = s;
  if (dead) { alive = false; my_carts += kd + 1; }
}

// lane = window: carts [c0, c1) for the window at `base`
// (a branch-free form of the reject -- the four partial sums, a ballot per cart, the carts of rejected windows counted
// per wave in scalar registers -- was measured 9 % slower per walk than this one, whose dead lanes skip the rest)
template <int DEPTH, bool NORM>
__device__ __forceinline__ void p_uni(const PCtx& c, int c0, int c1, int base, bool& alive, float& score, unsigned& my_carts,
                                      bool ilp8) {
  int k = c0;
  if (ilp8) {
    for (; k + 8 <= c1; k += 8) {
      if (__ballot(alive) == 0ull) break;
      if (alive) {
        int lf[8];
        scan_trees<DEPTH, false, 8>(c.t_nodes, k, c.node_n, c.lds, base, c.D, lf, 1, 0x7fffffff, c.bc);
        p_apply<8, NORM>(c, k, lf, alive, score, my_carts);
      }
    }
  }
  for (; k + 4 <= c1; k += 4) {
    if (__ballot(alive) == 0ull) break;
    if (alive) {
      int lf[4];
      scan_trees<DEPTH, false, 4>(c.t_nodes, k, c.node_n, c.lds, base, c.D, lf, 1, 0x7fffffff, c.bc);
      p_apply<4, NORM>(c, k, lf, alive, score, my_carts);
    }
  }
  for (; k < c1; k++) {
    if (__ballot(alive) == 0ull) break;
    if (alive) {
      int lf[1];
      lf[0] = scan_tree<DEPTH, false>(c.t_nodes + k * c.node_n, c.lds, base, c.D, c.bc) - c.node_n;
      p_apply<1, NORM>(c, k, lf, alive, score, my_carts);
    }
  }
}

// Pair task: np = 1 << lg items; lane -> item lane & (np - 1), cart group lane >> lg; a lane walks the 8 consecutive
// carts r0 + 8 * group + u of its item (one batch: three dependent LDS round trips), the leaf indices go to
// lf[item][cart of the round] as ONE 8-byte store, then lanes [0, np) replay the round's scores in cart order.
// Only lanes [0, np) carry alive / score in and out.
template <int DEPTH>
__device__ __forceinline__ void p_pair(const PCtx& c, uint8_t* lf, int lg, int c0, int c1, int lane, bool has_item, int base,
                                       bool& alive, float& score, unsigned& my_carts) {
  const int np = 1 << lg;
  const int rc = 8 * (64 >> lg);                       // carts per round
  const int item = lane & (np - 1);
  const int grp = lane >> lg;
  for (int r0 = c0; r0 < c1; r0 += rc) {
    const unsigned long long live = __ballot(alive);   // (bits [0, np): the replayers)
    if (live == 0ull) break;
    const int r1 = min(c1, r0 + rc);
    const int ka = r0 + 8 * grp;
    if (has_item && ((live >> item) & 1ull) && ka < r1) {
      int lf8[8];
      scan_trees<DEPTH, false, 8>(c.t_nodes, ka, c.node_n, c.lds, base, c.D, lf8, 1, r1 - 1, c.bc);
      uint2 pk;
      pk.x = (unsigned)lf8[0] | ((unsigned)lf8[1] << 8) | ((unsigned)lf8[2] << 16) | ((unsigned)lf8[3] << 24);
      pk.y = (unsigned)lf8[4] | ((unsigned)lf8[5] << 8) | ((unsigned)lf8[6] << 16) | ((unsigned)lf8[7] << 24);
      *(uint2*)(lf + item * rc + 8 * grp) = pk;
    }
    wave_lds_sync();
    {
      int k = r0;
      // 16 carts at a time when none of them normalises: leaf scores and thresholds first, then the recurrence in
      // registers, strictly in cart order (c/jda.c:395-399)
      for (; k + 16 <= r1; k += 16) {
        if (__ballot(alive) == 0ull) break;
        const ThNormF pm = *(const ThNormF*)&c.t_par[k + (lane & 15)];      // lane u (mod 16): cart k+u
        if (__ballot(pm.norm != 0.f) != 0ull) break;                        // rare: the generic loop below takes over
        float thv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) thv[u] = rl(pm.th, u);
        if (alive) {
          int lfi[16];
          float lsv[16];
          const uint4 pk = *(const uint4*)(lf + item * rc + (k - r0));
          const unsigned pw4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
          for (int u = 0; u < 16; u++) lfi[u] = (int)((pw4[u >> 2] >> (8 * (u & 3))) & 0xffu);
#pragma unroll
          for (int u = 0; u < 16; u++) lsv[u] = c.t_leaf[(k + u) * c.leaf_n + lfi[u]];
          float sums[16];
          float sc = score;
          unsigned rej = 0u;
#pragma unroll
          for (int u = 0; u < 16; u++) {
            sc = sc + lsv[u];                                            // c/jda.c:396 (no normalisation here)
            sums[u] = sc;
            rej |= (sc < thv[u]) ? (1u << u) : 0u;                       // c/jda.c:399
          }
          if (rej) {
            const int j = __ffs((int)rej) - 1;
            float sd = sums[0];
#pragma unroll
            for (int u = 1; u < 16; u++) sd = (j >= u) ? sums[u] : sd;
            score = sd;
            alive = false;
            my_carts += k + j + 1;
          } else {
            score = sc;
          }
        }
      }
      for (; k < r1; k++) {
        if (__ballot(alive) == 0ull) break;
        if (alive) {
          int l1[1];
          l1[0] = (int)lf[item * rc + (k - r0)];
          p_apply<1, true>(c, k, l1, alive, score, my_carts);
        }
      }
    }
    wave_lds_sync();                                   // the next round overwrites lf
  }
}

}  // namespace

// Rings: `ring_cap` items each (a power of two).  A producer reserves with a compare-and-swap that checks the room; when
// a ring is full its survivors are not queued -- the task walks them through the next cart range itself (lanes partly
// empty, but no wave ever waits for room: no deadlock, no overflow, whatever the rings' size).
void scan_p_ring_caps(PScanCfg* cfg, int waves) {
  (void)waves;
  int cap = 64;
  while (cap < cfg->ring_cap[0]) cap *= 2;
  for (int b = 0; b < kPScanMaxBuckets; b++) { cfg->ring_cap[b] = cap; cfg->ring_off[b] = b * cap; }
  cfg->ring_items = cfg->nb * cap;
}
size_t scan_p_lds_bytes(const PScanCfg& cfg, int carts, int node_n, int leaf_n, int waves) {
  return (size_t)PLds(carts, node_n, leaf_n, cfg.ring_items, waves, cfg.slots, cfg.slot_bytes).total;
}

// RAGGED: the tiles of a ragged batch (images of different sizes in one pass, kernels.h: RagSeg / RagBlk): a tile is
// named by the launch's block map, its geometry -- the level's tile re-cut for the image, the image's own window grid --
// travels with its slot.  A template parameter like k_scan's (a run-time test in the hot prologue cost that kernel 13 %).
template <int DEPTH, bool RAGGED>
// Registers: launch_bounds(1024) alone lets the compiler take 128 (117 used); a workgroup of 768 threads is three waves
// per SIMD, and what it leaves of the 512 registers decides which of the other batch's kernels can run next to it.
#ifndef JDA_SCAN_P_WAVES_PER_EU
#define JDA_SCAN_P_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(JDA_SCAN_P_WAVES_PER_EU)))
void k_scan_p(const DevPlan* __restrict__ plan, DevModelT<float> m, const S0Node* __restrict__ table, WorkT<float> w,
              int level, PScanCfg cfg, int total_blocks, int blk_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int NW = blockDim.x >> 6;
  const int node_n = m.node_n, leaf_n = m.leaf_n;
  const int K = cfg.bound_last;                        // this kernel stops here and hands survivors to k_finish
  const PLds L(K, node_n, leaf_n, cfg.ring_items, NW, cfg.slots, cfg.slot_bytes);
  PCtl* ctl = (PCtl*)(lds + L.ctl);
  uint2* rings = (uint2*)(lds + L.rings);
  uint8_t* lfw = lds + L.lfbuf + wv * 512;
  const DevLevel lv = plan->lv[level];
  const int W = plan->width;
  const int TH = cfg.th;                               // rows of windows per tile: this kernel's own cut (<= the plan's)
  const int tiles_per_frame = lv.tiles_x * cfg.tiles_y;
  const int G = gridDim.x;
  // Tiles are dealt in sequence v = class + classes * q (class = workgroup index mod 8 = its XCD: the frames of a group
  // of 8 stay on one XCD's L2).  Static dealing gives workgroup b the q = b / 8 + j * G / 8 (its share is fixed: right when
  // the launch has the machine to itself).  Dynamic dealing (cfg.dyn_slot >= 0) takes q from a device counter per class:
  // a workgroup that gets its CU late -- another batch's kernels were on it -- takes fewer tiles instead of keeping the
  // whole launch waiting for its fixed share.  The control block's "next tile" word is then 0 until a claim finds the
  // sequence exhausted (n_my = 1).
  const bool dyn = cfg.dyn_slot >= 0;
  const int classes = G >= 8 ? 8 : 1;
  const int tile_class = (int)blockIdx.x % classes;
  unsigned long long* const dyn_cnt = w.counters + (size_t)tile_class * kCntStride + kCntTotal + (dyn ? cfg.dyn_slot : 0);
  const int n_my = dyn ? 1 : (total_blocks - (int)blockIdx.x + G - 1) / G;
  const int S = cfg.slots;

#ifdef JDA_SCAN_TIMING
  unsigned long long t_cat[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};    // fresh, lane = window bucket, pair bucket, tile load, idle, scheduling (won / lost), snapshot wait, push, item read
  unsigned n_cat[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = t_last;
#define JDA_PCAT(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_cat[i] += t_now - t_last; n_cat[i]++; t_last = t_now; } while (0)
#define JDA_PSUB_BEGIN() const unsigned long long t_sub0 = __builtin_amdgcn_s_memtime()
#define JDA_PSUB_END(i) do { lds_drain(); t_cat[i] += __builtin_amdgcn_s_memtime() - t_sub0; n_cat[i]++; } while (0)
#else
#define JDA_PCAT(i) do { } while (0)
#define JDA_PSUB_BEGIN() do { } while (0)
#define JDA_PSUB_END(i) do { } while (0)
#endif

  // ---- prologue: the cart tables of [0, K) once per workgroup, control block cleared ----
  dma_to_lds_rt(lds + L.nodes, table + lv.s0_table, K * node_n * (int)sizeof(S0Node), tid, (int)blockDim.x);
  dma_to_lds_rt(lds + L.leaf, m.leaf, K * leaf_n * 4, tid, (int)blockDim.x);
  dma_to_lds_rt(lds + L.par, (const CartPar<float>*)m.par0, K * (int)sizeof(CartPar<float>), tid, (int)blockDim.x);
  for (int i = tid; i < (int)(sizeof(PCtl) / 4); i += blockDim.x) ((int*)ctl)[i] = 0;
  __syncthreads();
  if (tid == 0) {
    ctl->cfgw[kCfNb] = cfg.nb;
#pragma unroll
    for (int i = 0; i <= kPScanMaxBuckets; i++) ctl->cfgw[kCfBound + i] = cfg.bound[i];
#pragma unroll
    for (int i = 0; i < kPScanMaxBuckets; i++) { ctl->cfgw[kCfLg + i] = cfg.lg[i]; ctl->cfgw[kCfCap + i] = cfg.ring_cap[i]; ctl->cfgw[kCfOff + i] = cfg.ring_off[i]; }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  const int cfgv = ctl->cfgw[lane];
  auto CF = [&](int i) { return __builtin_amdgcn_readlane(cfgv, i); };
  const int NB = CF(kCfNb);
  // lane 8 + b: items a full task of ring b takes
  // (task form digit: 6 lane = window, 64 items; 3 / 2 lane = window, a full task is 32 / 16 items; 5 / 4 pair task of 32 / 16)
  // (... 7 / 8: pair task of 8 / 4 windows -- 64 / 128 carts per round: the deep cart ranges that few windows reach)
  // (... 9: pair task of up to 4 windows taken as soon as ONE waits: a window that deep must not keep its tile's slot
  //  waiting for company)
  auto take_of = [](int d) { return d == 6 ? 64 : (d == 5 || d == 3) ? 32 : d == 7 ? 8 : (d == 8 || d == 9) ? 4 : 16; };
  auto need_of = [&](int d) { return d == 9 ? 1 : take_of(d); };
  auto is_pair = [](int d) { return d == 4 || d == 5 || d == 7 || d == 8 || d == 9; };
  auto pair_lg = [](int d) { return d == 7 ? 3 : (d == 8 || d == 9) ? 2 : d; };
  const int need_v = need_of(__shfl(cfgv, kCfLg + ((lane - kRecRing) & 7)));
  typedef volatile __attribute__((address_space(3))) v4i* lds_rec_t;
  lds_rec_t rec_l = (lds_rec_t)(ctl->rec);
  int* const recw = (int*)ctl->rec;          // word view for the atomics: record r, field f = recw[4 r + f]

  PCtx c;
  c.lds = lds;
  c.bc = Bc(L.slots, L.slots + (long long)cfg.slots * cfg.slot_bytes);
  c.t_nodes = (const S0Node*)(lds + L.nodes);
  c.t_leaf = (const float*)(lds + L.leaf);
  c.t_par = (const CartPar<float>*)(lds + L.par);
  c.node_n = node_n; c.leaf_n = leaf_n; c.D = m.D;
  const bool ilp8_fresh = (cfg.opts & 1) != 0, ilp8_bucket = (cfg.opts & 2) != 0;
  const bool any_norm = cfg.any_norm != 0;

  unsigned my_carts = 0, handed = 0, win_cov = 0, mids = 0;
  int idle_spins = 0;
  unsigned long long* const scan_err = w.counters + (size_t)kCntScanErrShard * kCntStride + kCntScanErr;

  const int C = CF(kCfCap);                            // items per ring (a power of two)
  // survivors -> ring b, if it has room (reserved by compare-and-swap against the pop counter); false: no room
  auto push_ring = [&](int b, bool alive, uint32_t packed, float score) -> bool {
    const unsigned long long mask = __ballot(alive);
    const int n = __popcll(mask);
    if (n == 0) return true;
    JDA_PSUB_BEGIN();
    int* rw = recw + 4 * (kRecRing + b);
    int start;
    for (;;) {
      const v4i q = rec_l[kRecRing + b];
      const int rsv = uni(q.z), pop = uni(q.y);
      if (rsv + n - pop > C) return false;
      int old = 0;
      if (lane == 0) old = atomicCAS(rw + 2, rsv, rsv + n);
      if (uni(old) == rsv) { start = rsv; break; }
    }
    if (alive) {
      const int rank = __popcll(mask & lanes_below(lane));
      rings[b * C + ((start + rank) & (C - 1))] = make_uint2(packed, __float_as_uint(score));
    }
    lds_drain();
    int spin = 0;
    for (; ld_relaxed(rw) != start && spin < kPSpinMax; spin++) __builtin_amdgcn_s_sleep(1);   // commit in reservation order
    compiler_fence();
    if (spin >= kPSpinMax) {               // (never seen; see kPSpinMax above: reported, not committed)
      if (lane == 0) atomicOr(scan_err, 2ull);
      return true;
    }
    if (lane == 0) st_relaxed(rw, start + n);
    JDA_PSUB_END(8);
    return true;
  };
  // windows alive after cart K - 1 -> hand-off queue (k_finish continues at cart K)
  auto hand_off = [&](bool alive, uint32_t packed, float score) {
    const unsigned long long mask = __ballot(alive);
    const int n = __popcll(mask);
    if (n == 0) return;
    unsigned gbase = 0;
    if (lane == 0) gbase = (unsigned)atomicAdd(&w.counters[cfg.to_mid ? kCntMid : kCntTail], (unsigned long long)n);
    gbase = (unsigned)uni((int)gbase);
    if (alive) {
      const int rank = __popcll(mask & lanes_below(lane));
      const int s = (int)(packed >> (kPBaseBits + kPWidxBits));
      const int widx = (int)((packed >> kPBaseBits) & ((1u << kPWidxBits) - 1u));
      const v4i g = rec_l[kRecGeo + s];
      v4i g2 = g;
      if (RAGGED) g2 = rec_l[kRecGeo2 + s];
      const int tw_s = RAGGED ? g2.z : lv.tw;
      const int wy = widx / tw_s, wx = widx - wy * tw_s;
      const int frame = g.z, wx0 = g.w & 0xffff, wy0 = (int)((unsigned)g.w >> 16);
      const unsigned slot = gbase + (unsigned)rank;
      if (slot < (cfg.to_mid ? w.cap_m : w.cap_q)) {
        JDA_BC(Bc(0, cfg.to_mid ? w.cap_m : w.cap_q), slot, 1, kBcQueue);
        const uint32_t gid = RAGGED ? (uint32_t)(g2.x + (wy0 + wy) * g2.y + wx0 + wx)
                                    : (uint32_t)(frame * plan->windows + lv.base + (wy0 + wy) * lv.nx + wx0 + wx);
        const uint32_t xy = (uint32_t)((wx0 + wx) * lv.step) | ((uint32_t)((wy0 + wy) * lv.step) << 16);
        const uint32_t wf = (uint32_t)lv.win | ((uint32_t)frame << 16);
        if (cfg.to_mid) {                  // stage 0 passed: k_finish(survivors) takes it from the mid queue, as k_filter0 leaves it
          w.m_gid[slot] = gid; w.m_score[slot] = score; w.m_xy[slot] = xy; w.m_wf[slot] = wf;
        } else {
          w.q_gid[slot] = gid; w.q_score[slot] = score; w.q_kstart[slot] = (uint32_t)K; w.q_xy[slot] = xy; w.q_wf[slot] = wf;
        }
      }
      handed += K;
      if (cfg.to_mid) mids += 1;
    }
    lds_drain();                                         // (the slot records have been read before the references go)
  };
  // a set of windows to walk, one per lane where r_ok: level rv (-1 fresh, else they have completed bound[rv] carts)
  int rv = -2;                                         // -2: nothing to walk
  uint32_t r_pk = 0u;
  float r_sc = 0.f;
  bool r_ok = false;

  const unsigned slot_bits = (1u << S) - 1u;
  for (;;) {
    if (rv == -2) {
      // ================= choose what to do next =================
      {
        // ---- pick: one record per lane, candidates as ballots ----
        const v4i r = rec_l[lane & 15];
        compiler_fence();
        const bool is_slot = lane < S, is_ring = (unsigned)(lane - kRecRing) < (unsigned)NB;
        const int bidx = r.y & 0xfff;
        const int avail_v = r.x - r.y;
        const int sx = r.x & 3;            // (slot lanes: the state; r.x >> 2 is the claim generation)
        const unsigned m_dr = (unsigned)__ballot(is_slot && (sx == 0 || (sx == 2 && bidx >= r.z && r.w == 0)));
        const unsigned m_fr = (unsigned)__ballot(is_slot && sx == 2 && bidx < r.z);
        const unsigned m_full = (unsigned)__ballot(is_ring && avail_v >= need_v);
        const int next_j = __builtin_amdgcn_readlane(r.x, kRecMisc);
        int task = -1;                     // 0 fresh, 1 ring, 2 tile load
        int t_b = 0, t_n = 0, t_s = 0, t_j = 0, t_gen = 0, t_xg = 0;
        uint2 t_it = make_uint2(0u, 0u);
        bool lost = false;                 // a compare-and-swap went to another wave: look again
        unsigned m_ring = m_full;
        if (!(next_j < n_my && m_dr != 0u) && m_full == 0u && m_fr == 0u) {
          // nothing fresh, no full ring task.  While a tile is on its way: wait.  Else every slot waits for its last
          // windows (or the launch is ending): whatever the deepest ring holds.
          const unsigned m_ld = (unsigned)__ballot(is_slot && sx == 1);
          if (m_ld == 0u) m_ring = (unsigned)__ballot(is_ring && avail_v > 0);
        }
        if (next_j < n_my && m_dr != 0u) {
          // ---- tiles left and a slot that is free or has drained: bring the next tile in ----
          const int s = __builtin_ctz(m_dr);
          const int stt = __builtin_amdgcn_readlane(r.x, s);
          int got = 0;
          if (lane == 0) got = (atomicCAS(recw + 4 * s, stt, (stt & ~3) | 1) == stt) ? 1 : 0;
          if (uni(got)) { task = 2; t_s = s; t_gen = (__builtin_amdgcn_readlane(r.y, s) >> 12) + 1; t_xg = (int)((((unsigned)stt >> 2) + 1u) & 0x1fffffffu) << 2; }
          else lost = true;
        } else if (m_ring != 0u) {
          // ---- the deepest ring that holds a full task (or, draining, anything).  The items are read BEFORE the
          //      compare-and-swap that takes them (a wave's LDS operations execute in order): while the pop counter
          //      stands, no producer writes into that part of the ring, so a swap that succeeds proves the items read
          //      are the ones taken -- and producers may reuse the room at once ----
          const int rl_ = 31 - __builtin_clz(m_ring);
          const int b = rl_ - kRecRing;
          const int dg = CF(kCfLg + b);
          const int full_n = take_of(dg);
          const int need = m_full != 0u ? need_of(dg) : 1;
          const int cmt = __builtin_amdgcn_readlane(r.x, rl_);
          int expv = __builtin_amdgcn_readlane(r.y, rl_);
          const int item = is_pair(dg) ? (lane & (full_n - 1)) : lane;
          for (;;) {
            const int n = min(cmt - expv, full_n);
            t_it = make_uint2(0u, 0u);
            if (item < n) t_it = rings[b * C + ((expv + item) & (C - 1))];
            int old = 0;
            if (lane == 0) old = atomicCAS(recw + 4 * rl_ + 1, expv, expv + n);
            old = uni(old);
            if (old == expv) { task = 1; t_b = b; t_n = n; break; }
            expv = old;
            if (cmt - expv < need) { lost = true; break; }
          }
        } else if (m_fr != 0u) {
          // ---- fresh windows: y = generation << 12 | next batch, taken with an atomic add (never lost to another
          //      wave).  A wave that looked at the slot's previous tile may get a batch of the NEXT one -- the loader
          //      publishes a new generation only when the tile has landed, so the batch is good ----
          const int s = __builtin_ctz(m_fr);
          int old = 0;
          if (lane == 0) old = atomicAdd(recw + 4 * s + 1, 1);
          old = uni(old);
          int nbt = __builtin_amdgcn_readlane(r.z, s);
          if ((old >> 12) != (__builtin_amdgcn_readlane(r.y, s) >> 12)) nbt = uni(ld_relaxed(recw + 4 * s + 2));
          if ((old & 0xfff) < nbt) { task = 0; t_s = s; t_j = old & 0xfff; }
          else lost = true;
        }
        if (task < 0) {
          if (lost) { __builtin_amdgcn_s_sleep(2); JDA_PCAT(6); continue; }
          // ---- nothing to do: finished when every tile has been brought in and every slot has drained ----
          if (next_j >= n_my && (m_dr & slot_bits) == slot_bits) break;
          if (++idle_spins > kPIdleMax) {                // (watchdog: a scheduling bug must not hang the device -- nor pass unseen)
            if (lane == 0) atomicOr(scan_err, 1ull);
            break;
          }
          __builtin_amdgcn_s_sleep(16);
          JDA_PCAT(4);
          continue;
        }
        JDA_PCAT(5);
        idle_spins = 0;

        if (task == 2) {
          const int s = t_s;
          // skip blocks of the padded frame group that have no frame
          int j = 0, frame = 0, trel = 0;
          bool have = false;
          for (;;) {
            int v;
            if (dyn) {
              unsigned long long q = 0;
              if (lane == 0) q = atomicAdd(dyn_cnt, 1ull);
              const long long vv = (long long)tile_class + (long long)classes * (long long)(unsigned)uni((int)(unsigned)q);
              if (vv >= (long long)total_blocks) { st_relaxed(recw + 4 * kRecMisc, 1); break; }
              v = (int)vv;
            } else {
              if (lane == 0) j = atomicAdd(recw + 4 * kRecMisc, 1);
              j = uni(j);
              if (j >= n_my) break;
              v = (int)blockIdx.x + j * G;
            }
            if (RAGGED) { trel = v; have = true; break; }          // (trel: the block's index in the launch's map)
            const int group = v / (8 * tiles_per_frame);
            const int rr = v - group * (8 * tiles_per_frame);
            frame = group * 8 + (rr & 7);
            trel = rr >> 3;
            if (frame < w.n_frames) { have = true; break; }
          }
          if (!have) {
            if (lane == 0) { recw[4 * s + 2] = 0; recw[4 * s + 3] = 0; }
            lds_drain();
            st_relaxed(recw + 4 * s, t_xg | 0);
          } else {
            // the tile's grid: the level's (uniform batch) or its image's own with the level's tile re-cut for it (ragged:
            // block map -> segment; wave-uniform values that arrive through vector loads, readfirstlane puts them in SGPRs)
            int tiles_x = lv.tiles_x, tw_s = lv.tw, th_s = TH, nx_s = lv.nx, ny_s = lv.ny, gid_b = 0;
            const uint8_t* img;
            if (RAGGED) {
              const RagBlk bs = w.blk[blk_base + trel];
              const RagSeg sg = w.segs[(unsigned)uni((int)bs.seg)];
              trel = uni((int)bs.tile);
              tiles_x = uni((int)sg.tiles_x); tw_s = uni((int)sg.tw); th_s = uni((int)sg.th);
              nx_s = uni((int)sg.nx); ny_s = uni((int)sg.ny); gid_b = uni((int)sg.gid_base); frame = uni((int)sg.image);
              const unsigned long long io = (unsigned long long)(unsigned)uni((int)(unsigned)(sg.img_off & 0xffffffffu)) |
                                            ((unsigned long long)(unsigned)uni((int)(unsigned)(sg.img_off >> 32)) << 32);
              img = w.frames + io;
            } else {
              img = w.frames + (size_t)frame * w.frame_stride;
            }
            const int ty = trel / tiles_x, tx = trel - ty * tiles_x;
            const int wx0 = tx * tw_s, wy0 = ty * th_s;
            const int twe = min(tw_s, nx_s - wx0), the = min(th_s, ny_s - wy0);
            const int x0 = wx0 * lv.step, y0 = wy0 * lv.step;
            const int pw = lv.win + (twe - 1) * lv.step, ph = lv.win + (the - 1) * lv.step;
#ifdef JDA_BOUNDS_CHECK
            const Bc bc_fr((long long)(uintptr_t)w.bc_lo, (long long)(uintptr_t)w.bc_hi);
#else
            const Bc bc_fr;
#endif
            const int xshift = load_tile<64>(lds + L.slots + s * cfg.slot_bytes, w.frames, w.frame_stride, img, W, x0, y0, pw, ph,
                                             lv.pitch, lane, bc_fr, Bc(0, cfg.slot_bytes));
            const int nbatch = (tw_s * the + 63) >> 6;
            if (lane == 0) {
              int* g = recw + 4 * (kRecGeo + s);
              g[0] = twe | (the << 16); g[1] = xshift; g[2] = frame; g[3] = wx0 | (wy0 << 16);
              if (RAGGED) { int* g2 = recw + 4 * (kRecGeo2 + s); g2[0] = gid_b; g2[1] = nx_s; g2[2] = tw_s; g2[3] = 0; }
              recw[4 * s + 2] = nbatch; recw[4 * s + 3] = nbatch;
            }
            win_cov += (lane == 0) ? (unsigned)(twe * the) : 0u;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the tile has landed, the slot record too
            st_relaxed(recw + 4 * s + 1, (t_gen & 0x7ffff) << 12);
            st_relaxed(recw + 4 * s, t_xg | 2);
          }
          JDA_PCAT(3);
          continue;
        } else if (task == 0) {
          // ---- fresh: windows [64 j, 64 j + 64) of slot t_s from cart 0 ----
          const int s = t_s;
          const v4i g = rec_l[kRecGeo + s];
          const int twe = g.x & 0xffff, the = g.x >> 16, xshift = g.y;
          const int i = 64 * t_j + lane;
          int tw_s = lv.tw;
          if (RAGGED) tw_s = rec_l[kRecGeo2 + s].z;
          const int wy = (!RAGGED && cfg.tw_magic) ? (int)(((unsigned)i * (unsigned)cfg.tw_magic) >> 20) : i / tw_s;
          const int wx = i - wy * tw_s;
          r_ok = wx < twe && wy < the;
          const int base = L.slots + s * cfg.slot_bytes + (wy * lv.step) * lv.pitch + wx * lv.step + xshift;
          r_pk = (uint32_t)base | ((uint32_t)i << kPBaseBits) | ((uint32_t)s << (kPBaseBits + kPWidxBits));
          r_sc = 0.f;
          // the batch's reference on the slot becomes one reference per window (ended windows give theirs back below)
          const int nv = __popcll(__ballot(r_ok));
          if (nv != 1 && lane == 0) (void)__hip_atomic_fetch_add(recw + 4 * s + 3, nv - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (nv == 0) { JDA_PCAT(0); continue; }
          rv = -1;
        } else {
          // ---- t_n items of shared ring t_b (already in t_it) ----
          const int b = t_b;
          const int dg = CF(kCfLg + b);
          if (is_pair(dg)) {
            // pair form: walked here, survivors straight on
            const int plg = pair_lg(dg);
            const int np = 1 << plg;
            const int item = lane & (np - 1);
            const bool has_item = item < t_n;
            const bool valid = has_item && lane < np;
            bool alive = valid;
            float score = __uint_as_float(t_it.y);
            p_pair<DEPTH>(c, lfw, plg, CF(kCfBound + b), CF(kCfBound + b + 1), lane, has_item, (int)(t_it.x & ((1u << kPBaseBits) - 1u)),
                          alive, score, my_carts);
            if (valid && !alive) (void)__hip_atomic_fetch_add(recw + 4 * (int)(t_it.x >> (kPBaseBits + kPWidxBits)) + 3, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            JDA_PCAT(2);
            // survivors: next ring, or hand-off, or (ring full) walked on below as a lane = window set
            if (b + 1 == NB) {
              hand_off(alive, t_it.x, score);
              if (alive) (void)__hip_atomic_fetch_add(recw + 4 * (int)(t_it.x >> (kPBaseBits + kPWidxBits)) + 3, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              continue;
            }
            if (push_ring(b + 1, alive, t_it.x, score)) continue;
            rv = b + 1; r_pk = t_it.x; r_sc = score; r_ok = alive;
          } else {
            rv = b; r_pk = t_it.x; r_sc = __uint_as_float(t_it.y); r_ok = lane < t_n;
          }
        }
      }
    }

    // ================= walk the set: carts [bound[rv], bound[rv + 1]), lane = window (the one hot call site) =================
    {
      const int c0 = rv < 0 ? 0 : CF(kCfBound + rv), c1 = CF(kCfBound + rv + 1);
      bool alive = r_ok;
      if (any_norm) p_uni<DEPTH, true>(c, c0, c1, (int)(r_pk & ((1u << kPBaseBits) - 1u)), alive, r_sc, my_carts, rv < 0 ? ilp8_fresh : ilp8_bucket);
      else p_uni<DEPTH, false>(c, c0, c1, (int)(r_pk & ((1u << kPBaseBits) - 1u)), alive, r_sc, my_carts, rv < 0 ? ilp8_fresh : ilp8_bucket);
      const int nx = rv + 1;               // the survivors have completed bound[nx] carts
      if (nx == NB) { hand_off(alive, r_pk, r_sc); alive = false; }
      // windows that have ended give their references back
      if (r_ok && !alive) (void)__hip_atomic_fetch_add(recw + 4 * (int)(r_pk >> (kPBaseBits + kPWidxBits)) + 3, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      JDA_PCAT(rv < 0 ? 0 : 1);
      rv = -2;
      const unsigned long long mask = __ballot(alive);
      const int n = __popcll(mask);
      if (n == 0) continue;
      // the ring of their level; when it is full the same lanes walk on through the next range
      if (!push_ring(nx, alive, r_pk, r_sc)) { rv = nx; r_ok = alive; }
    }
  }

  // ---- counters: rejected windows are final (DetectionStatisic.cart_gothrough_n); handed-off windows are counted
  //      by k_finish when they terminate.  One atomic set per workgroup, on this workgroup's counter shard. ----
  unsigned v = my_carts, hv = handed, cv = win_cov, mv = mids;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o); hv += __shfl_xor(hv, o); cv += __shfl_xor(cv, o); mv += __shfl_xor(mv, o); }
  __syncthreads();
  int* red = (int*)(lds + L.lfbuf);      // (the pair scratch is idle now: 512 B per wave)
  if (lane == 0) { red[wv] = (int)v; red[16 + wv] = (int)hv; red[32 + wv] = (int)cv; red[48 + wv] = (int)mv; }
#ifdef JDA_SCAN_TIMING
  if (lane == 0) {
    for (int i = 0; i < 10; i++) { atomicAdd(&ctl->dbg[i], t_cat[i]); atomicAdd(&ctl->dbg[10 + i], (unsigned long long)n_cat[i]); }
  }
#endif
  __syncthreads();
  if (tid == 0) {
    unsigned long long sv = 0, sh = 0, sc = 0, sm = 0;
    for (int i = 0; i < NW; i++) { sv += (unsigned)red[i]; sh += (unsigned)red[16 + i]; sc += (unsigned)red[32 + i]; sm += (unsigned)red[48 + i]; }
    if (sm) atomicAdd(shard_counter(w.counters, kCntMidScan), sm);
    if (sv) atomicAdd(shard_counter(w.counters, kCntCarts), sv);
    atomicAdd(shard_counter(w.counters, kCntCartsScan), sv + sh);
    atomicAdd(shard_counter(w.counters, kCntWinScan), sc);
#ifdef JDA_SCAN_TIMING
    if (w.dbg && blockIdx.x < 2048) {
      unsigned long long* o = w.dbg + (49152 + (size_t)(level & 7) * 2048 + blockIdx.x) * 32;      // (a row per level: later launches do not overwrite it)
      o[0] = 0x5000ull | ((unsigned long long)level << 32);
      o[1] = __builtin_amdgcn_s_memtime() - t_begin;
      for (int i = 0; i < 20; i++) o[2 + i] = ctl->dbg[i];
      o[22] = (unsigned long long)n_my;
    }
#endif
  }
}

hipError_t launch_scan_persistent(int level, const PScanCfg& cfg, int block, int grid_max, const DevPlan* d_plan,
                                  const DevPlan& h_plan, const DevModelT<float>& m, const S0Node* table,
                                  const WorkT<float>& w, hipStream_t stream, int rag_blk_base, int rag_blk_n) {
  const bool ragged = rag_blk_n >= 0;
  if (!ragged && w.n_frames == 0) return hipSuccess;
  if (ragged && rag_blk_n == 0) return hipSuccess;
  const DevLevel& lv = h_plan.lv[level];
  if (lv.tiled != 1 || cfg.nb < 0 || cfg.nb > kPScanMaxBuckets || cfg.slots < 1 || cfg.slots > kPSlotsMax) return hipErrorInvalidValue;
  if (!ragged && (cfg.th < 1 || cfg.th > lv.th || cfg.tiles_y != (lv.ny + cfg.th - 1) / cfg.th)) return hipErrorInvalidValue;
  if (ragged && (!w.segs || !w.blk)) return hipErrorInvalidValue;
  if (cfg.to_mid && cfg.bound_last != m.K) return hipErrorInvalidValue;
  if (cfg.dyn_slot >= kCntMidScan - kCntTotal) return hipErrorInvalidValue;
  if ((!ragged && lv.tw * cfg.th > (1 << kPWidxBits)) || block < 64 || block > 1024 || (block & 63)) return hipErrorInvalidValue;
  const int K = cfg.bound_last;
  const PLds L(K, m.node_n, m.leaf_n, cfg.ring_items, block / 64, cfg.slots, cfg.slot_bytes);
  if (L.total > 160 * 1024 || L.total > (1 << kPBaseBits)) return hipErrorInvalidValue;
  const int groups = (w.n_frames + 7) / 8;
  const int total_blocks = ragged ? rag_blk_n : groups * 8 * lv.tiles_x * cfg.tiles_y;
  int grid = std::min(total_blocks, grid_max);
  if (grid >= 8) grid &= ~7;             // block b runs on XCD b % 8: a workgroup's tiles b + j * grid stay on its XCD's frames
  auto go = [&](auto kern) {
    if (L.total > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)block), L.total, stream, d_plan, m, table, w, level, cfg,
                       total_blocks, ragged ? rag_blk_base : 0);
  };
  if (ragged) {
    if (m.D == 4) go(k_scan_p<4, true>);
    else if (m.D == 6) go(k_scan_p<6, true>);
    else go(k_scan_p<0, true>);
  } else {
    if (m.D == 4) go(k_scan_p<4, false>);
    else if (m.D == 6) go(k_scan_p<6, false>);
    else go(k_scan_p<0, false>);
  }
  return hipGetLastError();
}

JDA_BC_READER(k_scan_p)

}  // namespace jda
