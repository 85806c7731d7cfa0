// k_scan: the first `handoff` carts of stage 0 -- lane = window over an LDS pixel tile, survivors
// compacted by ballot/prefix-sum after every phase of carts; phases with few windows left spread
// (window, cart) pairs over all lanes and replay the scores 16 carts at a time in registers.
// Reference loop being replaced: c/jda.c:357-402 (stage 0 only; every window still holds the mean
// shape there, so the feature offsets are resolved per (node, level) by k_prep_stage0).
//
// Three pixel modes (template MODE, DevLevel::tiled):
//   1  LDS tile, 16-bit node offsets (window side x tile pitch below 64 KiB)
//   3  LDS tile, 21-bit node offsets (big windows: a few windows share a tile of up to ~150 KiB)
//   2  no tile: pixels through L1/L2 (windows that do not fit LDS at all)
// The cart tables (resolved nodes, leaf scores, cart parameters) are staged in LDS one CHUNK of
// carts at a time; a workgroup that still has windows alive at the end of a chunk loads the next
// one, up to `handoff` carts.
//
// This file is the kernel and its launchers; three translation units include it and instantiate one third of the
// (Real, TRACE, MODE, BLOCK, DEPTH, RAGGED, NORM) combinations each, so that they compile side by side (one unit took
// 80 s of a 95-s build): k_scan.hip (dialect C, uniform batches), k_scan_d.hip (dialect CPP), k_scan_r.hip (ragged, dialect C),
// k_scan_dr.hip (ragged, dialect CPP).
#pragma once
#include "kernels_common.h"
#include "scan_walk.h"

namespace jda {

#ifndef JDA_SCAN_TU_MAIN
int scan_handoff_cap(int node_n, int leaf_n, int real_bytes);
#else
// Carts per table chunk: nodes, leaf scores and cart parameters of a chunk are staged in LDS
// together, within a 16 KiB table budget.
int scan_handoff_cap(int node_n, int leaf_n, int real_bytes) {
  const int per_cart = node_n * (int)sizeof(S0Node) + leaf_n * real_bytes + 4 * real_bytes;
  int c = (16 * 1024) / per_cart;
  if (c < 8) c = 8;
  return c >= 64 ? (c & ~63) : (c & ~7);     // whole phases per chunk where the budget allows
}
#endif

namespace {

constexpr int kScanMaxWindows = 512;     // windows per tile at most (queue capacity)

template <typename Real>
struct ThNorm { Real th, norm; };        // first half of CartPar: all the common case needs

template <typename Real, bool TRACE>
struct ScanLds {
  // byte offsets inside dynamic LDS
  int pix, nodes, leaf, par, q_widx, q_score, q_hash, lfbuf, misc, total;
  __host__ __device__ ScanLds(int pix_bytes, int carts, int node_n, int leaf_n, int m_max, int lf_bytes) {
    int o = 0;
    pix = o; o += (pix_bytes + 15) & ~15;
    nodes = o; o += carts * node_n * (int)sizeof(S0Node); o = (o + 15) & ~15;
    leaf = o; o += carts * leaf_n * (int)sizeof(Real); o = (o + 15) & ~15;
    par = o; o += carts * (int)sizeof(CartPar<Real>);
    q_score = o; o += 2 * m_max * (int)sizeof(Real);
    q_widx = o; o += 2 * m_max * 2; o = (o + 15) & ~15;
    q_hash = o; if (TRACE) o += 2 * m_max * 4;
    lfbuf = o; o += lf_bytes;           // leaf indices [window][cart of the round], n_pad x (lf_bytes / n_pad) (pair phases)
    misc = o; o += 128;
    total = o;
  }
};

}  // namespace

#ifdef JDA_SCAN_TU_MAIN
size_t scan_lds_bytes(int pix_bytes, int carts, int node_n, int leaf_n, int real_bytes, bool trace, int block) {
  const int lf = block * 8;
  if (real_bytes == 4) return trace ? ScanLds<float, true>(pix_bytes, carts, node_n, leaf_n, kScanMaxWindows, lf).total
                                    : ScanLds<float, false>(pix_bytes, carts, node_n, leaf_n, kScanMaxWindows, lf).total;
  return trace ? ScanLds<double, true>(pix_bytes, carts, node_n, leaf_n, kScanMaxWindows, lf).total
               : ScanLds<double, false>(pix_bytes, carts, node_n, leaf_n, kScanMaxWindows, lf).total;
}
#endif

// Registers: three 512-thread workgroups per CU are 6 waves per SIMD, which 80 VGPRs still allow and 82 do not (a
// two-register creep cost 0.17 ms per step in an r02 experiment) -- the occupancy the LDS footprint permits is pinned.
// RAGGED: the block map of a ragged batch names the tile (kernels.h: RagSeg / RagBlk).  A template parameter, not a
// run-time test: with the two ways of finding the tile in one kernel the uniform batch ran 13 % slower (r03, same
// opcode counts -- the level record no longer stayed where the hot loop wants it).
// NORM = false: no cart of [0, K) normalises its score (known on the host): the per-cart test of the flag and its branch
// leave the walks (k_scan_p measured -23 % per walk from this; a run-time flag instead of the template parameter: no gain).
template <typename Real, int DEPTH, bool TRACE, int MODE, int BLOCK, bool RAGGED = false, bool NORM = true>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BLOCK == 512 ? 6 : 4)))
void k_scan(const DevPlan* __restrict__ plan, DevModelT<Real> m,
                                                const S0Node* __restrict__ table, WorkT<Real> w,
                                                int level, int tiles_total, int pix_bytes, int handoff, int chunk,
                                                int cp_max, int opts, int blk_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr bool GLB = MODE == 2;
  constexpr bool WIDE = MODE != 1;
  constexpr int M_MAX = kScanMaxWindows;
  constexpr int NW = BLOCK / 64;
  constexpr int LF = BLOCK * 8;          // bytes of lfbuf: 8 trees per lane and round in the pair phases
  const int node_n = m.node_n, leaf_n = m.leaf_n;
  const int K = min(m.K, handoff);     // this kernel stops here and hands survivors to k_finish
  // 8 instead of 4 trees in flight per lane (levels with few resident waves); fp64 LDS tiles: never (eight fp64 leaf scores
  // and thresholds in flight are 32 registers the 80-register instantiation does not have: it spilled them, r06)
  const bool ilp8 = GLB || (sizeof(Real) == 4 && (opts & 1));
  const int first_phase = (opts >> 8) & 0xff;   // carts before the first compaction (8 or 16)
  const ScanLds<Real, TRACE> L(pix_bytes, chunk, node_n, leaf_n, M_MAX, LF);
  const uint8_t* pix = lds + L.pix;
  Real* q_score = (Real*)(lds + L.q_score);
  uint16_t* q_widx = (uint16_t*)(lds + L.q_widx);
  unsigned* q_hash = (unsigned*)(lds + L.q_hash);
  uint8_t* lfbuf = lds + L.lfbuf;
  int* misc = (int*)(lds + L.misc);   // [0],[1] queue counts; [2] global base; [4..] counter partials

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
#ifdef JDA_SCAN_TIMING
  unsigned long long stamps[15];
  int n_stamp = 0;
  int items_at[15];
#define JDA_STAMP(v) do { if (n_stamp < 15) { items_at[n_stamp] = (v); stamps[n_stamp++] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define JDA_STAMP(v) do { } while (0)
#endif
  JDA_STAMP(0);

  // XCD-aware block -> (frame, tile): blocks b, b+8, b+16.. land on one XCD
  // (MI355X dispatches block b to XCD b % 8), so the 8 frames of a group each
  // stay inside one XCD's L2.
  // level < 0: one launch covers every level of this pixel mode; tiles_total = their tiles per frame.
  // Ragged batch (w.segs): the host's block map names the (image, level) segment and the tile.
  int frame_, trel_, gid0_, level_ = level;
  unsigned long long img_off_ = 0;
  RagSeg sg{};
  if constexpr (RAGGED) {
    // (wave-uniform values, but loaded through vector memory: readfirstlane puts them where the uniform batch has
    // them, in SGPRs -- the tile geometry derived from them feeds the address arithmetic of the hot loop)
    auto uni = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const RagBlk bs = w.blk[blk_base + blockIdx.x];
    const RagSeg g = w.segs[uni(bs.seg)];
    sg.nx = (uint16_t)uni(g.nx); sg.ny = (uint16_t)uni(g.ny); sg.tiles_x = (uint16_t)uni(g.tiles_x);
    sg.tw = (uint16_t)uni(g.tw); sg.th = (uint16_t)uni(g.th);
    sg.win = (int)uni((unsigned)g.win); sg.step = (int)uni((unsigned)g.step); sg.pitch = (int)uni((unsigned)g.pitch);
    sg.s0_table = (int)uni((unsigned)g.s0_table); sg.tiled = (int)uni((unsigned)g.tiled);
    level_ = (int)uni(g.level);
    frame_ = (int)uni(g.image); trel_ = (int)uni(bs.tile); gid0_ = (int)uni(g.gid_base);
    img_off_ = (unsigned long long)uni((unsigned)(g.img_off & 0xffffffffu)) | ((unsigned long long)uni((unsigned)(g.img_off >> 32)) << 32);
  } else {
    int tiles_per_frame = tiles_total;
    if (level >= 0) tiles_per_frame = plan->lv[level].tiles_x * plan->lv[level].tiles_y;
    const int b = blockIdx.x;
    const int group = b / (8 * tiles_per_frame);
    const int r = b - group * (8 * tiles_per_frame);
    frame_ = group * 8 + (r & 7);
    trel_ = r >> 3;
    if (frame_ >= w.n_frames) return;
    if (level < 0) {
      // (a merged launch may cover a range of levels only: blk_base = first | (last + 1) << 8, 0 = every level)
      const int lv_lo = blk_base & 0xff, lv_hi = blk_base ? (blk_base >> 8) & 0xff : plan->n_levels;
      level_ = lv_lo;
      for (int i = lv_lo; i < lv_hi; i++) {
        const DevLevel* c = &plan->lv[i];
        if (c->tiled != MODE) continue;
        const int cnt = c->tiles_x * c->tiles_y;
        if (trel_ < cnt) { level_ = i; break; }
        trel_ -= cnt;
      }
    }
  }
  level = level_;
  DevLevel lv_;
  if constexpr (RAGGED) {
    lv_.win = sg.win; lv_.step = sg.step; lv_.pitch = sg.pitch; lv_.s0_table = sg.s0_table; lv_.tiled = sg.tiled;
    lv_.nx = sg.nx; lv_.ny = sg.ny; lv_.tiles_x = sg.tiles_x; lv_.tw = sg.tw; lv_.th = sg.th;
    lv_.base = 0; lv_.tiles_y = 0;
  } else {
    lv_ = plan->lv[level];
  }
  const DevLevel lv = lv_;
  const int frame = frame_, trel = trel_;
  const int gid0 = RAGGED ? gid0_ : frame * plan->windows + lv.base;
  const uint8_t* img = RAGGED ? w.frames + img_off_ : w.frames + (size_t)frame * w.frame_stride;
  const int ty = trel / lv.tiles_x, tx = trel - ty * lv.tiles_x;
  const int wx0 = tx * lv.tw, wy0 = ty * lv.th;                 // first window of the tile
  const int twe = min(lv.tw, lv.nx - wx0), the = min(lv.th, lv.ny - wy0);
  const int x0 = wx0 * lv.step, y0 = wy0 * lv.step;             // tile origin in the frame
  const int pw = lv.win + (twe - 1) * lv.step, ph = lv.win + (the - 1) * lv.step;

  // ---- stage the pixel tile (LDS-DMA when the frame is 16-byte aligned) and the first table chunk ----
  const int W = plan->width;
  int xshift = 0;
#ifdef JDA_BOUNDS_CHECK
  const Bc bc_fr((long long)(uintptr_t)w.bc_lo, (long long)(uintptr_t)w.bc_hi);      // the frames' device range
  const Bc bc_pix = GLB ? bc_fr : Bc(0, pix_bytes);                                   // what a pixel gather may touch
#else
  const Bc bc_fr, bc_pix;
#endif
  if (GLB) pix = img + (size_t)y0 * W + x0;       // window origins are offsets from the tile origin in the frame
  else xshift = load_tile<BLOCK>(lds + L.pix, w.frames, w.frame_stride, img, W, x0, y0, pw, ph, lv.pitch, tid, bc_fr, Bc(0, (pix_bytes + 15) & ~15));
  auto load_tables = [&](int kb, int kc) {
    dma_to_lds<BLOCK>(lds + L.nodes, table + lv.s0_table + (size_t)kb * node_n, kc * node_n * (int)sizeof(S0Node), tid);
    dma_to_lds<BLOCK>(lds + L.leaf, m.leaf + (size_t)kb * leaf_n, kc * leaf_n * (int)sizeof(Real), tid);
    dma_to_lds<BLOCK>(lds + L.par, (const CartPar<Real>*)m.par0 + kb, kc * (int)sizeof(CartPar<Real>), tid);
  };
  load_tables(0, min(chunk, K));
  if (tid == 0) { misc[0] = 0; misc[1] = 0; }
  __builtin_amdgcn_s_waitcnt(0);     // vmcnt(0): LDS-DMA loads have landed (a barrier does not drain VMEM)
  __syncthreads();
  JDA_STAMP(lv.tw * lv.th);

  const int n_tile = lv.tw * lv.th;      // phase 0 enumerates the full tile; edge windows are filtered
  int n_items = n_tile;
  int cur = 0;
  bool queued = false;                   // the live windows are in queue `cur` (else: the whole tile, phase 0)
  unsigned my_carts = 0;

  // A tile of few windows (big windows) goes through the pair phases from the start: its valid
  // windows are queued here.
  if (n_tile <= cp_max) {
    for (int i0 = 0; i0 < n_tile; i0 += BLOCK) {
      const int i = i0 + tid;
      const int wy = i / lv.tw, wx = i - wy * lv.tw;
      const bool ok = i < n_tile && wx < twe && wy < the;
      const unsigned long long mask = __ballot(ok);
      if (mask) {
        int wbase = 0;
        if (lane == 0) wbase = atomicAdd(&misc[0], __popcll(mask));
        wbase = __shfl(wbase, 0);
        if (ok) {
          const int pos = wbase + __popcll(mask & lanes_below(lane));
          q_widx[pos] = (uint16_t)i;
          q_score[pos] = (Real)0;
          if (TRACE) q_hash[pos] = kFnvSeed;
        }
      }
    }
    __syncthreads();
    n_items = misc[0];
    queued = true;
  }

  for (int kb = 0; kb < K && n_items > 0; kb += chunk) {
    const int ke = min(K, kb + chunk);
    if (kb > 0) {
      // next chunk of cart tables (every reader of the previous one is past the phase-end barrier)
      load_tables(kb, ke - kb);
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      JDA_STAMP(-200 - kb);
    }
    // tables addressed by absolute cart index
    const S0Node* t_nodes = (const S0Node*)(lds + L.nodes) - (size_t)kb * node_n;
    const Real* t_leaf = (const Real*)(lds + L.leaf) - (size_t)kb * leaf_n;
    const CartPar<Real>* t_par = (const CartPar<Real>*)(lds + L.par) - kb;

    // Phases: carts [0,8) [8,16) [16,32) [32,64) [64,128) [128,256) ... (cut at chunk ends); after each
    // the survivors are compacted so that later phases run on full waves.
    for (int c0 = kb; c0 < ke && n_items > 0;) {
      // No scalar-memory load may be pending when the walk loops start: LDS reads return in order and are waited for
      // one by one (lgkmcnt(n)), but a scalar load shares that counter and returns out of order -- with one possibly
      // in flight the compiler falls back to lgkmcnt(0) after EVERY LDS read of the hot loop (seen in the RAGGED
      // instantiation: 50 x lgkmcnt(0) instead of lgkmcnt(1..4), phases 30-60 % slower).
      __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0), vmcnt / expcnt untouched
      const bool cart_parallel = queued && n_items <= cp_max && leaf_n <= 256;
      int plen = c0 < first_phase ? first_phase - c0 : min(c0, c0 >= 128 ? 128 : 64);
      if (cart_parallel && plen < 16) plen = 16;
      const int c1 = min(ke, c0 + plen);

      // Applies carts [k, k+CNT) to this lane's window: the CNT trees first (they do
      // not depend on the score, so their LDS round trips overlap), then the scores
      // strictly in cart order with the per-cart reject test.
      auto apply = [&](auto cnt_tag, int k, const int* lf, bool& alive, Real& score, unsigned& hash, int gid) {
        constexpr int CNT = decltype(cnt_tag)::value;
        // all table reads first and unconditionally (they depend on the leaves only), so that
        // their LDS round trips overlap; the dependent part below is pure arithmetic
        ThNorm<Real> p[CNT];
        Real lsv[CNT];
#pragma unroll
        for (int u = 0; u < CNT; u++) { p[u] = *(const ThNorm<Real>*)&t_par[k + u]; lsv[u] = t_leaf[(k + u) * leaf_n + lf[u]]; }
        Real s = score;
        bool dead = false;
        int kd = k;
#pragma unroll
        for (int u = 0; u < CNT; u++) {
          if (!dead) {
            s = s + lsv[u];                                                  // c/jda.c:396
            if (NORM && p[u].norm != (Real)0) { const CartPar<Real> q = t_par[k + u]; s = (s - q.mean) / q.std; }   // c/jda.c:397 (rare)
            if (TRACE) hash = fnv_step(hash, lf[u]);
            kd = k + u;
            dead = s < p[u].th;                                              // c/jda.c:399
          }
        }
        score = s;
        if (dead) {
          alive = false;
          my_carts += kd + 1;
          if (TRACE) { w.tr_carts[gid] = kd + 1; w.tr_score[gid] = s; w.tr_hash[gid] = hash; }
        }
      };

      if (!cart_parallel) {
        // ---- lane = window, every wave walks the whole phase for its own windows ----
        for (int i0 = 0; i0 < n_items; i0 += BLOCK) {
          const int i = i0 + tid;
          bool alive = i < n_items;
          int widx = 0;
          Real score = 0;
          unsigned hash = kFnvSeed;
          if (alive) {
            if (!queued) {
              widx = i;
            } else {
              widx = q_widx[cur * M_MAX + i];
              score = q_score[cur * M_MAX + i];
              if (TRACE) hash = q_hash[cur * M_MAX + i];
            }
          }
          const int wy = widx / lv.tw, wx = widx - wy * lv.tw;
          if (!queued) alive = alive && wx < twe && wy < the;
          const int base = (wy * lv.step) * lv.pitch + wx * lv.step + xshift;
          const int gid = gid0 + (wy0 + wy) * lv.nx + wx0 + wx;

          int k = c0;
          if (ilp8) {
            // global pixels (each tree level is a global-load round trip) or few resident waves per CU:
            // twice as many independent trees are kept in flight
            for (; k + 8 <= c1; k += 8) {
              if (__ballot(alive) == 0ull) break;
              if (alive) {
                int lf[8];
                if constexpr (RAGGED) {
                  scan_trees<DEPTH, WIDE, 8>(t_nodes, k, node_n, pix, base, m.D, lf, 1, 0x7fffffff, bc_pix, GLB);
                } else {
#pragma unroll
                  for (int u = 0; u < 8; u++) lf[u] = scan_tree<DEPTH, WIDE>(t_nodes + (k + u) * node_n, pix, base, m.D, bc_pix, GLB) - node_n;
                }
                apply(std::integral_constant<int, 8>{}, k, lf, alive, score, hash, gid);
              }
            }
          }
          for (; k + 4 <= c1; k += 4) {
            if (__ballot(alive) == 0ull) break;
            if (alive) {
              int lf[4];
              if constexpr (RAGGED) {
                scan_trees<DEPTH, WIDE, 4>(t_nodes, k, node_n, pix, base, m.D, lf, 1, 0x7fffffff, bc_pix, GLB);
              } else {
#pragma unroll
                for (int u = 0; u < 4; u++) lf[u] = scan_tree<DEPTH, WIDE>(t_nodes + (k + u) * node_n, pix, base, m.D, bc_pix, GLB) - node_n;
              }
              apply(std::integral_constant<int, 4>{}, k, lf, alive, score, hash, gid);
            }
          }
          for (; k < c1; k++) {
            if (alive) {
              int lf[1];
              lf[0] = scan_tree<DEPTH, WIDE>(t_nodes + k * node_n, pix, base, m.D, bc_pix, GLB) - node_n;
              apply(std::integral_constant<int, 1>{}, k, lf, alive, score, hash, gid);
            }
          }
          // ---- compact survivors into the next queue (ballot + prefix popcount) ----
          const unsigned long long mask = __ballot(alive);
          if (mask) {
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&misc[cur ^ 1], __popcll(mask));
            wbase = __shfl(wbase, 0);
            if (alive) {
              const int pos = wbase + __popcll(mask & lanes_below(lane));
              q_widx[(cur ^ 1) * M_MAX + pos] = (uint16_t)widx;
              q_score[(cur ^ 1) * M_MAX + pos] = score;
              if (TRACE) q_hash[(cur ^ 1) * M_MAX + pos] = hash;
            }
          }
        }
      } else {
        // ---- few windows left: (window, cart) PAIRS are spread over all lanes (trees only), then the
        //      scores of the round are replayed in cart order, lane = window, from the leaf indices in
        //      LDS.  These phases are latency and issue bound (few windows, long cart ranges), so a
        //      lane should walk as few trees as possible: n_pad = windows rounded up to a power of two
        //      (>= 16); lane -> window tid % n_pad, carts tid / n_pad + j * (BLOCK / n_pad); a round
        //      covers as many carts as lfbuf holds (LF / n_pad), i.e. at most 8 trees per lane, walked
        //      as one batch.
        int lg = 4;
        while ((1 << lg) < n_items) lg++;
        const int n_pad = 1 << lg;
        const int rc = LF >> lg;                          // carts per round
        const int cstride = BLOCK >> lg;                  // cart stride of a lane
        const int item = tid & (n_pad - 1);
        const bool has_item = item < n_items;
        const bool replayer = tid < n_pad;                // lanes [0, n_pad) also own the windows' scores
        const int replay_waves = n_pad <= 64 ? 1 : (n_pad >> 6);
        bool alive = replayer && has_item;
        int widx = 0;
        Real score = 0;
        unsigned hash = kFnvSeed;
        if (has_item) {
          widx = q_widx[cur * M_MAX + item];
          if (replayer) { score = q_score[cur * M_MAX + item]; if (TRACE) hash = q_hash[cur * M_MAX + item]; }
        }
        const int wy = widx / lv.tw, wx = widx - wy * lv.tw;
        const int base = (wy * lv.step) * lv.pitch + wx * lv.step + xshift;
        const int gid = gid0 + (wy0 + wy) * lv.nx + wx0 + wx;
        for (int r0 = c0; r0 < c1; r0 += rc) {
          const int r1 = min(c1, r0 + rc);
          const int ka = r0 + (tid >> lg);
          if (has_item) {
            if (ka + 7 * cstride < r1) {
              int lf8[8];
              if constexpr (RAGGED) {
                scan_trees<DEPTH, WIDE, 8>(t_nodes, ka, node_n, pix, base, m.D, lf8, cstride, 0x7fffffff, bc_pix, GLB);
              } else {
#pragma unroll
                for (int u = 0; u < 8; u++)
                  lf8[u] = scan_tree<DEPTH, WIDE>(t_nodes + (ka + u * cstride) * node_n, pix, base, m.D, bc_pix, GLB) - node_n;
              }
#pragma unroll
              for (int u = 0; u < 8; u++) lfbuf[item * rc + (ka + u * cstride - r0)] = (uint8_t)lf8[u];
            } else if (ka + cstride >= r1) {
              if (ka < r1) lfbuf[item * rc + (ka - r0)] = (uint8_t)(scan_tree<DEPTH, WIDE>(t_nodes + ka * node_n, pix, base, m.D, bc_pix, GLB) - node_n);
            } else {
              for (int k = ka; k < r1; k += 4 * cstride) {
                int lf4[4];
                if constexpr (RAGGED) {
                  scan_trees<DEPTH, WIDE, 4>(t_nodes, k, node_n, pix, base, m.D, lf4, cstride, r1 - 1, bc_pix, GLB);
                } else {
#pragma unroll
                  for (int u = 0; u < 4; u++)
                    lf4[u] = scan_tree<DEPTH, WIDE>(t_nodes + min(k + u * cstride, r1 - 1) * node_n, pix, base, m.D, bc_pix, GLB) - node_n;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                  if (k + u * cstride < r1) lfbuf[item * rc + (k + u * cstride - r0)] = (uint8_t)lf4[u];
              }
            }
          }
          __syncthreads();
          JDA_STAMP(-100 - (r0 - c0));          // timing build: trees of this round done
          if (wv < replay_waves) {
            int k = r0;
            // RB carts at a time when none of them is normalised: all leaf scores and thresholds
            // are fetched first (two LDS round trips for the batch instead of two per 4 carts),
            // then the recurrence runs in registers, strictly in cart order (c/jda.c:395-399).
            // RB = 16 for fp32; 8 for fp64, whose sums, leaf scores and thresholds take two registers each (16 at a
            // time were 96 registers of state: the 512-thread instantiation spilled them, r06)
            constexpr int RB = sizeof(Real) == 8 ? 8 : 16;
            for (; k + RB <= r1; k += RB) {
              if (__ballot(alive) == 0ull) break;
              const ThNorm<Real> pm = *(const ThNorm<Real>*)&t_par[k + (lane & (RB - 1))];   // lane u (mod RB): cart k+u
              if (NORM && __ballot(pm.norm != (Real)0) != 0ull) break;       // rare: the generic loop below takes over
              // thresholds: lane u holds cart k+u's; broadcast with readlane HERE, with the whole wave
              // active -- inside the divergent block below the lanes without a live window would not
              // have loaded theirs
              Real thv[RB];
#pragma unroll
              for (int u = 0; u < RB; u++) thv[u] = rl(pm.th, u);
              if (alive) {
                int lf[RB];
                Real lsv[RB];
                // the window's RB leaf indices are RB consecutive bytes of lfbuf[window][cart]
                unsigned pw4[4];
                if constexpr (RB == 16) {
                  const uint4 pk = *(const uint4*)(lfbuf + item * rc + (k - r0));
                  pw4[0] = pk.x; pw4[1] = pk.y; pw4[2] = pk.z; pw4[3] = pk.w;
                } else {
                  const uint2 pk = *(const uint2*)(lfbuf + item * rc + (k - r0));
                  pw4[0] = pk.x; pw4[1] = pk.y; pw4[2] = 0u; pw4[3] = 0u;
                }
#pragma unroll
                for (int u = 0; u < RB; u++) lf[u] = (int)((pw4[u >> 2] >> (8 * (u & 3))) & 0xffu);
#pragma unroll
                for (int u = 0; u < RB; u++) lsv[u] = t_leaf[(k + u) * leaf_n + lf[u]];
                // branch-free: the RB partial sums (the same adds in the same order), a bit per
                // rejecting cart, then the first set bit names the cart the window died at
                Real sums[RB];
                Real sc = score;
                unsigned rej = 0u;
#pragma unroll
                for (int u = 0; u < RB; u++) {
                  sc = sc + lsv[u];                                          // c/jda.c:396 (no normalisation here)
                  sums[u] = sc;
                  rej |= (sc < thv[u]) ? (1u << u) : 0u;                     // c/jda.c:399
                }
                if (rej) {
                  const int j = __ffs((int)rej) - 1;
                  Real sd = sums[0];
#pragma unroll
                  for (int u = 1; u < RB; u++) sd = (j >= u) ? sums[u] : sd;
                  if (TRACE)
                    for (int u = 0; u <= j; u++) hash = fnv_step(hash, lf[u]);
                  score = sd;
                  alive = false;
                  my_carts += k + j + 1;
                  if (TRACE) { w.tr_carts[gid] = k + j + 1; w.tr_score[gid] = sd; w.tr_hash[gid] = hash; }
                } else {
                  if (TRACE) {
#pragma unroll
                    for (int u = 0; u < RB; u++) hash = fnv_step(hash, lf[u]);
                  }
                  score = sc;
                }
              }
            }
            for (; k < r1; k += 4) {
              if (__ballot(alive) == 0ull) break;
              if (alive) {
                int lf[4];
#pragma unroll
                for (int u = 0; u < 4; u++) lf[u] = (k + u < r1) ? (int)lfbuf[item * rc + (k + u - r0)] : 0;
                if (k + 4 <= r1) {
                  apply(std::integral_constant<int, 4>{}, k, lf, alive, score, hash, gid);
                } else {
                  for (int u = 0; k + u < r1 && alive; u++)
                    apply(std::integral_constant<int, 1>{}, k + u, lf + u, alive, score, hash, gid);
                }
              }
            }
          }
          __syncthreads();
        }
        if (wv < replay_waves) {
          const unsigned long long mask = __ballot(alive);
          if (mask) {
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&misc[cur ^ 1], __popcll(mask));
            wbase = __shfl(wbase, 0);
            if (alive) {
              const int pos = wbase + __popcll(mask & lanes_below(lane));
              q_widx[(cur ^ 1) * M_MAX + pos] = (uint16_t)widx;
              q_score[(cur ^ 1) * M_MAX + pos] = score;
              if (TRACE) q_hash[(cur ^ 1) * M_MAX + pos] = hash;
            }
          }
        }
      }
      __syncthreads();
      cur ^= 1;
      queued = true;
      n_items = misc[cur];
      c0 = c1;
      JDA_STAMP(n_items);
      if (tid == 0) misc[cur ^ 1] = 0;     // next phase's output counter (its last readers are past an earlier barrier)
      __syncthreads();
    }
  }

  // ---- windows still alive after cart K-1 -> hand-off queue (k_finish continues at cart K) ----
  unsigned handed = 0;
  if (n_items > 0) {
    if (tid == 0) misc[2] = (int)atomicAdd(&w.counters[kCntTail], (unsigned long long)n_items);
    __syncthreads();
    const unsigned gbase = (unsigned)misc[2];
    for (int i = tid; i < n_items; i += BLOCK) {
      const int widx = q_widx[cur * M_MAX + i];
      const int wy = widx / lv.tw, wx = widx - wy * lv.tw;
      const unsigned slot = gbase + i;
      if (slot < w.cap_q) {
        JDA_BC(Bc(0, w.cap_q), slot, 1, kBcQueue);
        w.q_gid[slot] = (uint32_t)(gid0 + (wy0 + wy) * lv.nx + wx0 + wx);
        w.q_score[slot] = q_score[cur * M_MAX + i];
        w.q_kstart[slot] = (uint32_t)K;
        w.q_xy[slot] = (uint32_t)((wx0 + wx) * lv.step) | ((uint32_t)((wy0 + wy) * lv.step) << 16);
        w.q_wf[slot] = (uint32_t)lv.win | ((uint32_t)frame << 16);
        if (TRACE) w.q_hash[slot] = q_hash[cur * M_MAX + i];
      }
      handed += K;
    }
  }
  // ---- counters: rejected windows are final (DetectionStatisic.cart_gothrough_n);
  //      handed-off windows are counted by k_finish when they terminate.  One
  //      atomic set per workgroup, on this workgroup's counter shard. ----
  unsigned v = my_carts, hv = handed;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o); hv += __shfl_xor(hv, o); }
  __syncthreads();
  if (lane == 0) { misc[4 + wv] = (int)v; misc[4 + NW + wv] = (int)hv; }     // NW <= 8: misc has 32 words
  __syncthreads();
  if (tid == 0) {
    unsigned long long sv = 0, sh = 0;
    for (int i = 0; i < NW; i++) { sv += (unsigned)misc[4 + i]; sh += (unsigned)misc[4 + NW + i]; }
    if (sv) atomicAdd(shard_counter(w.counters, kCntCarts), sv);
    atomicAdd(shard_counter(w.counters, kCntCartsScan), sv + sh);
    if (GLB) atomicAdd(shard_counter(w.counters, kCntCartsScanGlb), sv + sh);
    atomicAdd(shard_counter(w.counters, kCntWinScan), (unsigned long long)(twe * the));
  }
#ifdef JDA_SCAN_TIMING
  JDA_STAMP(-1);
  if (tid == 0 && w.dbg && blockIdx.x < 49152) {      // (rows from 49152: k_scan_p's)
    unsigned long long* o = w.dbg + (size_t)blockIdx.x * 32;
    o[0] = (unsigned long long)n_stamp | ((unsigned long long)level << 32);
    for (int i = 0; i < n_stamp; i++) { o[1 + i] = stamps[i]; o[16 + i] = (unsigned long long)(long long)items_at[i]; }
  }
#endif
}

namespace {

template <typename Real, bool TRACE, int MODE, int BLOCK>
hipError_t launch_scan_mode(const DevPlan* d_plan, const DevPlan& h_plan, const DevModelT<Real>& m,
                            const S0Node* table, const WorkT<Real>& w, int level, int handoff, int cp_max, int opts,
                            hipStream_t stream, int lv_lo, int lv_hi) {
  lv_lo = std::max(0, lv_lo); lv_hi = std::min(lv_hi, h_plan.n_levels);
  // level >= 0: that level; level < 0: every level of pixel mode MODE in one launch, sized for the
  // largest tile (small batches, where one launch per level would only add launch latency)
  int tiles = 0, pix_bytes = 0;
  for (int i = 0; i < h_plan.n_levels; i++) {
    const DevLevel& lv = h_plan.lv[i];
    if (lv.tiled != MODE || (level >= 0 && i != level) || (level < 0 && (i < lv_lo || i >= lv_hi))) continue;
    tiles += lv.tiles_x * lv.tiles_y;
    if (MODE != 2) pix_bytes = std::max(pix_bytes, lv.pitch * (lv.win + (lv.th - 1) * lv.step));
  }
  if (tiles == 0) return hipSuccess;
  const int chunk = std::min(std::min(m.K, handoff), scan_handoff_cap(m.node_n, m.leaf_n, (int)sizeof(Real)));
  const ScanLds<Real, TRACE> L(pix_bytes, chunk, m.node_n, m.leaf_n, kScanMaxWindows, BLOCK * 8);
  if (L.total > 160 * 1024) return hipErrorInvalidValue;
  const int groups = (w.n_frames + 7) / 8;
  dim3 grid((unsigned)(groups * 8 * tiles)), block(BLOCK);
  const int lds_req = L.total;
  auto go = [&](auto kern) {
    if (lds_req > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_req);
    const bool ranged = level < 0 && (lv_lo > 0 || lv_hi < h_plan.n_levels);
    hipLaunchKernelGGL(kern, grid, block, lds_req, stream, d_plan, m, table, w, level < 0 ? -1 : level, tiles,
                       pix_bytes, handoff, chunk, cp_max, opts, ranged ? (lv_lo | (lv_hi << 8)) : 0);
  };
  // (opts bit 1: no cart of [0, handoff) normalises -- the lean instantiation, passes without trace only)
  if constexpr (!TRACE) {
    if (opts & 2) {
      if (m.D == 4) go(k_scan<Real, 4, TRACE, MODE, BLOCK, false, false>);
      else if (m.D == 6) go(k_scan<Real, 6, TRACE, MODE, BLOCK, false, false>);
      else go(k_scan<Real, 0, TRACE, MODE, BLOCK, false, false>);
      return hipGetLastError();
    }
  }
  if (m.D == 4) go(k_scan<Real, 4, TRACE, MODE, BLOCK>);
  else if (m.D == 6) go(k_scan<Real, 6, TRACE, MODE, BLOCK>);
  else go(k_scan<Real, 0, TRACE, MODE, BLOCK>);
  return hipGetLastError();
}

}  // namespace

template <typename Real>
hipError_t launch_scan(int mode, int level, bool trace, int handoff, int cp_max, int opts, const DevPlan* d_plan,
                       const DevPlan& h_plan, const DevModelT<Real>& m, const S0Node* table,
                       const WorkT<Real>& w, hipStream_t stream, int lv_lo, int lv_hi) {
  if (w.n_frames == 0) return hipSuccess;
  if (level >= 0 && h_plan.lv[level].tiled != mode) return hipSuccess;
  // 512-thread workgroups where a level's tiles hold more than 256 windows (one window per lane in
  // phase 0, twice the waves per LDS byte); a merged launch when any of its levels does
  bool big = false;
  for (int i = 0; i < h_plan.n_levels && mode == 1; i++)
    if (h_plan.lv[i].tiled == 1 && (level < 0 ? (i >= lv_lo && i < lv_hi) : i == level)) big = big || h_plan.lv[i].tw * h_plan.lv[i].th > 256;
  auto pick = [&](auto trace_tag) {
    constexpr bool TR = decltype(trace_tag)::value;
    switch (mode) {
      case 1:
        return big ? launch_scan_mode<Real, TR, 1, 512>(d_plan, h_plan, m, table, w, level, handoff, cp_max, opts, stream, lv_lo, lv_hi)
                   : launch_scan_mode<Real, TR, 1, 256>(d_plan, h_plan, m, table, w, level, handoff, cp_max, opts, stream, lv_lo, lv_hi);
      case 2: return launch_scan_mode<Real, TR, 2, 256>(d_plan, h_plan, m, table, w, level, handoff, cp_max, opts, stream, lv_lo, lv_hi);
      case 3: return launch_scan_mode<Real, TR, 3, 256>(d_plan, h_plan, m, table, w, level, handoff, cp_max, opts, stream, lv_lo, lv_hi);
      default: return hipErrorInvalidValue;
    }
  };
  return trace ? pick(std::true_type{}) : pick(std::false_type{});
}

namespace {
template <typename Real, bool TRACE, int MODE, int BLOCK>
hipError_t launch_scan_ragged_mode(const DevPlan* d_plan, const DevModelT<Real>& m, const S0Node* table, const WorkT<Real>& w,
                                   int pix_bytes, int blk_base, int blk_n, int handoff, int cp_max, int opts,
                                   hipStream_t stream) {
  if (MODE == 2) pix_bytes = 0;
  const int chunk = std::min(std::min(m.K, handoff), scan_handoff_cap(m.node_n, m.leaf_n, (int)sizeof(Real)));
  const ScanLds<Real, TRACE> L(pix_bytes, chunk, m.node_n, m.leaf_n, kScanMaxWindows, BLOCK * 8);
  if (L.total > 160 * 1024) return hipErrorInvalidValue;
  auto go = [&](auto kern) {
    if (L.total > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    hipLaunchKernelGGL(kern, dim3((unsigned)blk_n), dim3(BLOCK), L.total, stream, d_plan, m, table, w, -1, 0,
                       pix_bytes, handoff, chunk, cp_max, opts, blk_base);
  };
  if constexpr (!TRACE) {
    if (opts & 2) {
      if (m.D == 4) go(k_scan<Real, 4, TRACE, MODE, BLOCK, true, false>);
      else if (m.D == 6) go(k_scan<Real, 6, TRACE, MODE, BLOCK, true, false>);
      else go(k_scan<Real, 0, TRACE, MODE, BLOCK, true, false>);
      return hipGetLastError();
    }
  }
  if (m.D == 4) go(k_scan<Real, 4, TRACE, MODE, BLOCK, true>);
  else if (m.D == 6) go(k_scan<Real, 6, TRACE, MODE, BLOCK, true>);
  else go(k_scan<Real, 0, TRACE, MODE, BLOCK, true>);
  return hipGetLastError();
}
}  // namespace

#if defined(JDA_SCAN_TU_RAGGED) || defined(JDA_SCAN_TU_RAGGED_DOUBLE)
// (no trace: ragged passes are jdaDetectBatchRagged's / jdaDetectBatchCppRagged's; everything else runs image by image)
#ifdef JDA_SCAN_TU_RAGGED
#define JDA_RAGGED_REAL float
#else
#define JDA_RAGGED_REAL double
#endif
template <>
hipError_t launch_scan_ragged<JDA_RAGGED_REAL>(int mode, int block, bool trace, int handoff, int cp_max, int opts, const DevPlan* d_plan,
                                     const DevModelT<JDA_RAGGED_REAL>& m, const S0Node* table, const WorkT<JDA_RAGGED_REAL>& w, int pix_bytes,
                                     int blk_base, int blk_n, hipStream_t stream) {
  using Real = JDA_RAGGED_REAL;
  if (blk_n <= 0) return hipSuccess;
  if (!w.segs || !w.blk || trace) return hipErrorInvalidValue;
  switch (mode) {
    case 1:
      return block == 512 ? launch_scan_ragged_mode<Real, false, 1, 512>(d_plan, m, table, w, pix_bytes, blk_base, blk_n, handoff, cp_max, opts, stream)
                          : launch_scan_ragged_mode<Real, false, 1, 256>(d_plan, m, table, w, pix_bytes, blk_base, blk_n, handoff, cp_max, opts, stream);
    case 2: return launch_scan_ragged_mode<Real, false, 2, 256>(d_plan, m, table, w, pix_bytes, blk_base, blk_n, handoff, cp_max, opts, stream);
    case 3: return launch_scan_ragged_mode<Real, false, 3, 256>(d_plan, m, table, w, pix_bytes, blk_base, blk_n, handoff, cp_max, opts, stream);
    default: return hipErrorInvalidValue;
  }
}
#endif
#ifdef JDA_SCAN_TU_DOUBLE
template hipError_t launch_scan<double>(int, int, bool, int, int, int, const DevPlan*, const DevPlan&, const DevModelT<double>&,
                                        const S0Node*, const WorkT<double>&, hipStream_t, int, int);
#endif
#ifdef JDA_SCAN_TU_MAIN
template hipError_t launch_scan<float>(int, int, bool, int, int, int, const DevPlan*, const DevPlan&, const DevModelT<float>&,
                                       const S0Node*, const WorkT<float>&, hipStream_t, int, int);
#endif

#ifdef JDA_SCAN_TU_MAIN
JDA_BC_READER(k_scan)
#endif
#ifdef JDA_SCAN_TU_DOUBLE
JDA_BC_READER(k_scan_d)
#endif
#ifdef JDA_SCAN_TU_RAGGED
JDA_BC_READER(k_scan_r)
#endif
#ifdef JDA_SCAN_TU_RAGGED_DOUBLE
JDA_BC_READER(k_scan_dr)
#endif

}  // namespace jda
