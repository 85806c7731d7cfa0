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
constexpr int kPSpinMax = 1 << 20;     // watchdog of every wait loop (a wait is microseconds; this is a large fraction of a second)

struct ThNormF { float th, norm; };

struct PCtl {                          // control block in LDS (ints; every access is an LDS atomic or a relaxed load)
  int next_j;                          // next tile (sequence number inside this workgroup's share) to bring in
  int err;
  int pad0[2];
  int r_rsv[kPScanMaxBuckets], r_cmt[kPScanMaxBuckets], r_pop[kPScanMaxBuckets];
  int s_state[kPSlotsMax];             // 0 free, 1 being loaded, 2 published
  int s_pending[kPSlotsMax];           // fresh batches not finished + items alive
  int s_fresh[kPSlotsMax];             // next fresh batch
  int s_nbatch[kPSlotsMax];
  int s_frame[kPSlotsMax], s_wx0[kPSlotsMax], s_wy0[kPSlotsMax], s_twe[kPSlotsMax], s_the[kPSlotsMax], s_xshift[kPSlotsMax];
  unsigned long long dbg[16];
};

struct PLds {
  int nodes, leaf, par, ctl, rings, lfbuf, slots, total;
  __host__ __device__ PLds(int carts, int node_n, int leaf_n, int nb, int ring_cap, int waves, int n_slots, int slot_bytes) {
    int o = 0;
    nodes = o; o += carts * node_n * (int)sizeof(S0Node); o = (o + 15) & ~15;
    leaf = o; o += carts * leaf_n * 4; o = (o + 15) & ~15;
    par = o; o += carts * (int)sizeof(CartPar<float>); o = (o + 15) & ~15;
    ctl = o; o += (int)sizeof(PCtl); o = (o + 15) & ~15;
    rings = o; o += nb * ring_cap * 8; o = (o + 15) & ~15;
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
  const uint8_t* lds;                  // pixels are addressed by absolute LDS offsets
  const S0Node* t_nodes; const float* t_leaf; const CartPar<float>* t_par;
  int node_n, leaf_n, D;
};

// carts [k, k+CNT) applied to this lane's window, strictly in cart order with the per-cart reject (c/jda.c:395-399)
template <int CNT>
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
      if (p[u].norm != 0.f) { const CartPar<float> q = c.t_par[k + u]; s = (s - q.mean) / q.std; }     // c/jda.c:397 (rare)
      kd = k + u;
      dead = s < p[u].th;                                                                              // c/jda.c:399
    }
  }
  score = s;
  if (dead) { alive = false; my_carts += kd + 1; }
}

// lane = window: carts [c0, c1) for the window at `base`
template <int DEPTH>
__device__ __forceinline__ void p_uni(const PCtx& c, int c0, int c1, int base, bool& alive, float& score, unsigned& my_carts,
                                      bool ilp8) {
  int k = c0;
  if (ilp8) {
    for (; k + 8 <= c1; k += 8) {
      if (__ballot(alive) == 0ull) break;
      if (alive) {
        int lf[8];
        scan_trees<DEPTH, false, 8>(c.t_nodes, k, c.node_n, c.lds, base, c.D, lf);
        p_apply<8>(c, k, lf, alive, score, my_carts);
      }
    }
  }
  for (; k + 4 <= c1; k += 4) {
    if (__ballot(alive) == 0ull) break;
    if (alive) {
      int lf[4];
      scan_trees<DEPTH, false, 4>(c.t_nodes, k, c.node_n, c.lds, base, c.D, lf);
      p_apply<4>(c, k, lf, alive, score, my_carts);
    }
  }
  for (; k < c1; k++) {
    if (__ballot(alive) == 0ull) break;
    if (alive) {
      int lf[1];
      lf[0] = scan_tree<DEPTH, false>(c.t_nodes + k * c.node_n, c.lds, base, c.D) - c.node_n;
      p_apply<1>(c, k, lf, alive, score, my_carts);
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
      scan_trees<DEPTH, false, 8>(c.t_nodes, ka, c.node_n, c.lds, base, c.D, lf8, 1, r1 - 1);
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
          p_apply<1>(c, k, l1, alive, score, my_carts);
        }
      }
    }
    wave_lds_sync();                                   // the next round overwrites lf
  }
}

}  // namespace

size_t scan_p_lds_bytes(const PScanCfg& cfg, int carts, int node_n, int leaf_n, int waves) {
  return (size_t)PLds(carts, node_n, leaf_n, cfg.nb, cfg.ring_cap, waves, cfg.slots, cfg.slot_bytes).total;
}

template <int DEPTH>
__global__ __launch_bounds__(1024)
void k_scan_p(const DevPlan* __restrict__ plan, DevModelT<float> m, const S0Node* __restrict__ table, WorkT<float> w,
              int level, PScanCfg cfg, int total_blocks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int NW = blockDim.x >> 6;
  const int node_n = m.node_n, leaf_n = m.leaf_n;
  const int K = cfg.bound[cfg.nb];                     // this kernel stops here and hands survivors to k_finish
  const PLds L(K, node_n, leaf_n, cfg.nb, cfg.ring_cap, NW, cfg.slots, cfg.slot_bytes);
  PCtl* ctl = (PCtl*)(lds + L.ctl);
  uint2* rings = (uint2*)(lds + L.rings);
  uint8_t* lfw = lds + L.lfbuf + wv * 512;
  const DevLevel lv = plan->lv[level];
  const int W = plan->width;
  const int tiles_per_frame = lv.tiles_x * lv.tiles_y;
  const int G = gridDim.x;
  const int n_my = (total_blocks - (int)blockIdx.x + G - 1) / G;
  const int C = cfg.ring_cap;
  const int S = cfg.slots;

#ifdef JDA_SCAN_TIMING
  unsigned long long t_cat[6] = {0, 0, 0, 0, 0, 0};    // fresh, lane = window bucket, pair bucket, tile load, idle, push/pop
  unsigned n_cat[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = t_last;
#define JDA_PCAT(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_cat[i] += t_now - t_last; n_cat[i]++; t_last = t_now; } while (0)
#else
#define JDA_PCAT(i) do { } while (0)
#endif

  // ---- prologue: the cart tables of [0, K) once per workgroup, control block cleared ----
  dma_to_lds_rt(lds + L.nodes, table + lv.s0_table, K * node_n * (int)sizeof(S0Node), tid, (int)blockDim.x);
  dma_to_lds_rt(lds + L.leaf, m.leaf, K * leaf_n * 4, tid, (int)blockDim.x);
  dma_to_lds_rt(lds + L.par, (const CartPar<float>*)m.par0, K * (int)sizeof(CartPar<float>), tid, (int)blockDim.x);
  for (int i = tid; i < (int)(sizeof(PCtl) / 4); i += blockDim.x) ((int*)ctl)[i] = 0;
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  PCtx c;
  c.lds = lds;
  c.t_nodes = (const S0Node*)(lds + L.nodes);
  c.t_leaf = (const float*)(lds + L.leaf);
  c.t_par = (const CartPar<float>*)(lds + L.par);
  c.node_n = node_n; c.leaf_n = leaf_n; c.D = m.D;
  const bool ilp8_fresh = (cfg.opts & 1) != 0, ilp8_bucket = (cfg.opts & 2) != 0;

  unsigned my_carts = 0, handed = 0, win_cov = 0;
  int idle_spins = 0;

  // survivors of a task -> ring `b` (b < nb) or the hand-off queue (b == nb); returns the number pushed
  auto push = [&](int b, bool alive, uint32_t packed, float score) -> int {
    const unsigned long long mask = __ballot(alive);
    const int n = __popcll(mask);
    if (n == 0) return 0;
    const int rank = __popcll(mask & lanes_below(lane));
    if (b < cfg.nb) {
      int start = 0;
      if (lane == 0) start = atomicAdd(&ctl->r_rsv[b], n);
      start = uni(start);
      // (never taken while the scheduling bound holds; guards the unread tail of the ring)
      for (int spin = 0; start + n - ld_relaxed(&ctl->r_pop[b]) > C && spin < kPSpinMax; spin++) __builtin_amdgcn_s_sleep(2);
      if (alive) {
        unsigned pos = (unsigned)(start + rank) % (unsigned)C;
        rings[b * C + pos] = make_uint2(packed, __float_as_uint(score));
      }
      lds_drain();
      for (int spin = 0; ld_relaxed(&ctl->r_cmt[b]) != start && spin < kPSpinMax; spin++) __builtin_amdgcn_s_sleep(1);   // commit in reservation order
      compiler_fence();
      if (lane == 0) st_relaxed(&ctl->r_cmt[b], start + n);
    } else {
      unsigned gbase = 0;
      if (lane == 0) gbase = (unsigned)atomicAdd(&w.counters[kCntTail], (unsigned long long)n);
      gbase = (unsigned)uni((int)gbase);
      if (alive) {
        const int s = (int)(packed >> (kPBaseBits + kPWidxBits));
        const int widx = (int)((packed >> kPBaseBits) & ((1u << kPWidxBits) - 1u));
        const int wy = widx / lv.tw, wx = widx - wy * lv.tw;
        const int frame = ctl->s_frame[s], wx0 = ctl->s_wx0[s], wy0 = ctl->s_wy0[s];
        const unsigned slot = gbase + (unsigned)rank;
        if (slot < w.cap) {
          w.q_gid[slot] = (uint32_t)(frame * plan->windows + lv.base + (wy0 + wy) * lv.nx + wx0 + wx);
          w.q_score[slot] = score;
          w.q_kstart[slot] = (uint32_t)K;
          w.q_xy[slot] = (uint32_t)((wx0 + wx) * lv.step) | ((uint32_t)((wy0 + wy) * lv.step) << 16);
          w.q_wf[slot] = (uint32_t)lv.win | ((uint32_t)frame << 16);
        }
        handed += K;
      }
    }
    return n;
  };
  // an item has ended (died or was handed off): drop its reference; the last one frees the slot
  auto release = [&](bool ended, uint32_t packed) {
    if (ended) {
      const int s = (int)(packed >> (kPBaseBits + kPWidxBits));
      const int old = atomicSub(&ctl->s_pending[s], 1);
      if (old == 1) st_relaxed(&ctl->s_state[s], 0);
    }
  };
  // takes n items off ring b if it holds at least `need`; returns the first position or -1
  auto try_pop = [&](int b, int need, int want, int* n_out) -> int {
    for (;;) {
      const int pop = ld_relaxed(&ctl->r_pop[b]);
      const int cmt = ld_relaxed(&ctl->r_cmt[b]);
      const int avail = uni(cmt - pop);
      if (avail < need || avail <= 0) return -1;
      const int n = min(avail, want);
      int got = 0;
      if (lane == 0) got = (atomicCAS(&ctl->r_pop[b], pop, pop + n) == pop) ? 1 : 0;
      if (uni(got)) { *n_out = n; return uni(pop); }
    }
  };

  for (;;) {
    int task = -1;                       // 0 fresh, 1 bucket
    int t_b = 0, t_n = 0, t_start = 0, t_s = 0, t_j = 0;

    // ---- 1. a free slot and tiles left: bring the next tile in ----
    if (ld_relaxed(&ctl->next_j) < n_my) {
      int claimed = -1;
      for (int s = 0; s < S && claimed < 0; s++) {
        if (ld_relaxed(&ctl->s_state[s]) == 0) {
          int got = 0;
          if (lane == 0) got = (atomicCAS(&ctl->s_state[s], 0, 1) == 0) ? 1 : 0;
          if (uni(got)) claimed = s;
        }
      }
      if (claimed >= 0) {
        const int s = claimed;
        // skip blocks of the padded frame group that have no frame
        int j = 0, frame = 0, trel = 0;
        bool have = false;
        for (;;) {
          if (lane == 0) j = atomicAdd(&ctl->next_j, 1);
          j = uni(j);
          if (j >= n_my) break;
          const int v = (int)blockIdx.x + j * G;
          const int group = v / (8 * tiles_per_frame);
          const int r = v - group * (8 * tiles_per_frame);
          frame = group * 8 + (r & 7);
          trel = r >> 3;
          if (frame < w.n_frames) { have = true; break; }
        }
        if (!have) {
          st_relaxed(&ctl->s_state[s], 0);
        } else {
          const int ty = trel / lv.tiles_x, tx = trel - ty * lv.tiles_x;
          const int wx0 = tx * lv.tw, wy0 = ty * lv.th;
          const int twe = min(lv.tw, lv.nx - wx0), the = min(lv.th, lv.ny - wy0);
          const int x0 = wx0 * lv.step, y0 = wy0 * lv.step;
          const int pw = lv.win + (twe - 1) * lv.step, ph = lv.win + (the - 1) * lv.step;
          const uint8_t* img = w.frames + (size_t)frame * w.frame_stride;
          const int xshift = load_tile<64>(lds + L.slots + s * cfg.slot_bytes, w.frames, w.frame_stride, img, W, x0, y0, pw, ph,
                                           lv.pitch, lane);
          const int nbatch = (lv.tw * the + 63) >> 6;
          const int gen = (ld_relaxed(&ctl->s_fresh[s]) >> 12) + 1;
          if (lane == 0) {
            ctl->s_frame[s] = frame; ctl->s_wx0[s] = wx0; ctl->s_wy0[s] = wy0; ctl->s_twe[s] = twe; ctl->s_the[s] = the;
            ctl->s_xshift[s] = xshift; ctl->s_nbatch[s] = nbatch; ctl->s_pending[s] = nbatch;
          }
          win_cov += (lane == 0) ? (unsigned)(twe * the) : 0u;
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the tile has landed, the slot record too
          st_relaxed(&ctl->s_fresh[s], (gen & 0x7ffff) << 12);
          st_relaxed(&ctl->s_state[s], 2);
        }
        JDA_PCAT(3);
        continue;
      }
    }

    // ---- 2. the deepest ring that holds a full task ----
    for (int b = cfg.nb - 1; b >= 0 && task < 0; b--) {
      const int need = 1 << cfg.lg[b];
      const int st = try_pop(b, need, need, &t_n);
      if (st >= 0) { task = 1; t_b = b; t_start = st; }
    }
    // ---- 3. fresh windows ----
    // (s_fresh = generation << 12 | next batch, taken by compare-and-swap: a wave that looked at the slot's previous
    // tile cannot take a batch of the next one -- the loader publishes a new generation)
    if (task < 0) {
      for (int s = 0; s < S && task < 0; s++) {
        for (;;) {
          const int f = ld_relaxed(&ctl->s_fresh[s]);
          const int nb_s = ld_relaxed(&ctl->s_nbatch[s]);
          const int stt = ld_relaxed(&ctl->s_state[s]);
          if (uni(stt) != 2 || uni(f & 0xfff) >= uni(nb_s)) break;
          int got = 0;
          if (lane == 0) got = (atomicCAS(&ctl->s_fresh[s], f, f + 1) == f) ? 1 : 0;
          if (uni(got)) { task = 0; t_s = s; t_j = uni(f & 0xfff); break; }
        }
      }
    }
    // ---- 4. whatever is left (a tile is late, or the launch is ending) ----
    if (task < 0) {
      for (int b = cfg.nb - 1; b >= 0 && task < 0; b--) {
        const int st = try_pop(b, 1, 1 << cfg.lg[b], &t_n);
        if (st >= 0) { task = 1; t_b = b; t_start = st; }
      }
    }
    if (task < 0) {
      // ---- 5. nothing to do: finished when every tile has been brought in and every slot is free again ----
      bool done = ld_relaxed(&ctl->next_j) >= n_my;
      for (int s = 0; s < S; s++) done = done && ld_relaxed(&ctl->s_state[s]) == 0;
      if (uni(done ? 1 : 0)) break;
      if (++idle_spins > kPSpinMax) break;             // (watchdog: a scheduling bug must not hang the device)
      __builtin_amdgcn_s_sleep(8);
      JDA_PCAT(4);
      continue;
    }
    JDA_PCAT(5);
    idle_spins = 0;

    if (task == 0) {
      // ---- fresh: windows [64 j, 64 j + 64) of slot t_s, carts [0, bound[0]) ----
      const int s = t_s;
      const int twe = uni(ctl->s_twe[s]), the = uni(ctl->s_the[s]), xshift = uni(ctl->s_xshift[s]);
      const int i = 64 * t_j + lane;
      const int wy = i / lv.tw, wx = i - wy * lv.tw;
      bool alive = wx < twe && wy < the;
      const bool valid = alive;
      const int base = L.slots + s * cfg.slot_bytes + (wy * lv.step) * lv.pitch + wx * lv.step + xshift;
      const uint32_t packed = (uint32_t)base | ((uint32_t)i << kPBaseBits) | ((uint32_t)s << (kPBaseBits + kPWidxBits));
      float score = 0.f;
      p_uni<DEPTH>(c, 0, cfg.bound[0], base, alive, score, my_carts, ilp8_fresh);
      (void)valid;
      // the survivors' references are taken BEFORE they become visible in the ring (a consumer may end them at once);
      // then the batch's own reference goes (hand-off: the survivors have ended already)
      const int ns = __popcll(__ballot(alive));
      if (cfg.nb > 0 && ns > 0 && lane == 0) atomicAdd(&ctl->s_pending[s], ns);
      push(0, alive, packed, score);
      if (lane == 0) {
        const int old = atomicSub(&ctl->s_pending[s], 1);
        if (old == 1) st_relaxed(&ctl->s_state[s], 0);
      }
      JDA_PCAT(0);
    } else {
      const int b = t_b;
      const int c0 = cfg.bound[b], c1 = cfg.bound[b + 1];
      const int lg = cfg.lg[b];
      if (lg == 6) {
        // ---- lane = window ----
        bool alive = lane < t_n;
        uint2 it = make_uint2(0u, 0u);
        if (alive) it = rings[b * C + (unsigned)(t_start + lane) % (unsigned)C];
        const bool valid = alive;
        const int base = (int)(it.x & ((1u << kPBaseBits) - 1u));
        float score = __uint_as_float(it.y);
        p_uni<DEPTH>(c, c0, c1, base, alive, score, my_carts, ilp8_bucket);
        push(b + 1, alive, it.x, score);
        release(valid && (!alive || b + 1 == cfg.nb), it.x);
        JDA_PCAT(1);
      } else {
        // ---- pair form ----
        const int np = 1 << lg;
        const int item = lane & (np - 1);
        const bool has_item = item < t_n;
        uint2 it = make_uint2(0u, 0u);
        if (has_item) it = rings[b * C + (unsigned)(t_start + item) % (unsigned)C];
        bool alive = has_item && lane < np;
        const bool valid = alive;
        const int base = (int)(it.x & ((1u << kPBaseBits) - 1u));
        float score = __uint_as_float(it.y);
        p_pair<DEPTH>(c, lfw, lg, c0, c1, lane, has_item, base, alive, score, my_carts);
        push(b + 1, alive, it.x, score);
        release(valid && (!alive || b + 1 == cfg.nb), it.x);
        JDA_PCAT(2);
      }
    }
  }

  // ---- counters: rejected windows are final (DetectionStatisic.cart_gothrough_n); handed-off windows are counted
  //      by k_finish when they terminate.  One atomic set per workgroup, on this workgroup's counter shard. ----
  unsigned v = my_carts, hv = handed, cv = win_cov;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o); hv += __shfl_xor(hv, o); cv += __shfl_xor(cv, o); }
  __syncthreads();
  int* red = (int*)(lds + L.lfbuf);      // (the pair scratch is idle now: 512 B per wave)
  if (lane == 0) { red[wv] = (int)v; red[16 + wv] = (int)hv; red[32 + wv] = (int)cv; }
#ifdef JDA_SCAN_TIMING
  if (lane == 0) {
    for (int i = 0; i < 6; i++) { atomicAdd(&ctl->dbg[i], t_cat[i]); atomicAdd(&ctl->dbg[6 + i], (unsigned long long)n_cat[i]); }
  }
#endif
  __syncthreads();
  if (tid == 0) {
    unsigned long long sv = 0, sh = 0, sc = 0;
    for (int i = 0; i < NW; i++) { sv += (unsigned)red[i]; sh += (unsigned)red[16 + i]; sc += (unsigned)red[32 + i]; }
    if (sv) atomicAdd(shard_counter(w.counters, kCntCarts), sv);
    atomicAdd(shard_counter(w.counters, kCntCartsScan), sv + sh);
    atomicAdd(shard_counter(w.counters, kCntWinScan), sc);
#ifdef JDA_SCAN_TIMING
    if (w.dbg && blockIdx.x < 65536) {
      unsigned long long* o = w.dbg + (size_t)blockIdx.x * 32;
      o[0] = 0x5000ull | ((unsigned long long)level << 32);
      o[1] = __builtin_amdgcn_s_memtime() - t_begin;
      for (int i = 0; i < 12; i++) o[2 + i] = ctl->dbg[i];
      o[14] = (unsigned long long)n_my;
    }
#endif
  }
}

hipError_t launch_scan_persistent(int level, const PScanCfg& cfg, int block, int grid_max, const DevPlan* d_plan,
                                  const DevPlan& h_plan, const DevModelT<float>& m, const S0Node* table,
                                  const WorkT<float>& w, hipStream_t stream) {
  if (w.n_frames == 0) return hipSuccess;
  const DevLevel& lv = h_plan.lv[level];
  if (lv.tiled != 1 || cfg.nb < 0 || cfg.nb > kPScanMaxBuckets || cfg.slots < 1 || cfg.slots > kPSlotsMax) return hipErrorInvalidValue;
  if (lv.tw * lv.th > (1 << kPWidxBits) || block < 64 || block > 1024 || (block & 63)) return hipErrorInvalidValue;
  const int K = cfg.bound[cfg.nb];
  const PLds L(K, m.node_n, m.leaf_n, cfg.nb, cfg.ring_cap, block / 64, cfg.slots, cfg.slot_bytes);
  if (L.total > 160 * 1024 || L.total > (1 << kPBaseBits)) return hipErrorInvalidValue;
  const int groups = (w.n_frames + 7) / 8;
  const int total_blocks = groups * 8 * lv.tiles_x * lv.tiles_y;
  int grid = std::min(total_blocks, grid_max);
  if (grid >= 8) grid &= ~7;             // block b runs on XCD b % 8: a workgroup's tiles b + j * grid stay on its XCD's frames
  auto go = [&](auto kern) {
    if (L.total > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)block), L.total, stream, d_plan, m, table, w, level, cfg,
                       total_blocks);
  };
  if (m.D == 4) go(k_scan_p<4>);
  else if (m.D == 6) go(k_scan_p<6>);
  else go(k_scan_p<0>);
  return hipGetLastError();
}

}  // namespace jda
