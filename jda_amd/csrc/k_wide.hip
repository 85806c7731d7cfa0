// k_finish_wide: WORKGROUP = window -- the finishing path of small jobs (a frame or a few), where the time of a
// call is the latency of ONE window's chain through the cascade and not the machine's throughput.
//
// k_finish gives a window one wave: per stage K/192 rounds of tree walks (each 2 x depth dependent memory round trips)
// and K/32 batches of regression row loads, all in sequence -- about 40 us per stage, 200 us for a window that
// passes all five stages of the shipped dimensions, during which a single frame keeps 0.1 % of the machine busy.
// Here the window gets up to 1,024 threads and most of a CU's LDS:
//   * thread = cart: every tree of a stage is walked at once (c/jda.c:366-394: the shape is fixed during a stage, so
//     the trees are independent of each other and of the score) -- one round of depth-1 levels;
//   * wave 0 replays the score recurrence strictly in cart order (c/jda.c:395-399) from the leaf scores in LDS;
//   * the K weight rows of the stage are fetched by ALL threads into LDS (LDS-DMA, every load in flight at once), then
//     the lanes of the first waves add them to the shape strictly in cart order (c/jda.c:404-411; dialect CPP: the
//     delta is summed from zero and added once, btcart.cpp:407-424).
// Same arithmetic in the same order as k_finish and the reference; the results are bit-identical.
#include "finish_common.h"

namespace jda {

namespace {

struct WideLds {
  int sh, lbf, lsc, lth, lfi, misc, rows, tile, total;
  int row_cap;        // weight rows the row buffer holds
  __host__ __device__ WideLds(int dim, int K, int real_bytes, int tile_bytes, int budget) {
    const int dim_pad = (dim + 1) & ~1;
    int o = 0;
    sh = o; o += dim_pad * real_bytes; o = (o + 15) & ~15;
    lbf = o; o += ((K + 3) & ~3) * 4;
    lsc = o; o += ((K + 3) & ~3) * real_bytes; o = (o + 15) & ~15;
    lth = o; o += ((K + 3) & ~3) * real_bytes; o = (o + 15) & ~15;
    lfi = o; o += (K + 15) & ~15;
    misc = o; o += 64 + kMaxStages * 4;
    tile = o; o += (tile_bytes + 15) & ~15;
    rows = o;
    const int left = budget - o;
    row_cap = left > 0 ? left / (dim * real_bytes) : 0;
    if (row_cap > K) row_cap = K;
    total = o + row_cap * dim * real_bytes;
  }
};

}  // namespace

template <typename DL, bool TRACE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_finish_wide(const DevPlan* __restrict__ plan, DevModelT<typename DL::Real> m,
                                                       WorkT<typename DL::Real> w, int apply_th, typename DL::Real final_th,
                                                       const S0Node* __restrict__ s0_table, int tile_win, int tile_bytes,
                                                       int lds_budget, int conc) {
  using Real = typename DL::Real;
  constexpr bool kCpp = sizeof(Real) == 8;
  constexpr int NW = BLOCK / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int T = m.T, K = m.K, node_n = m.node_n, leaf_n = m.leaf_n, dim = m.dim;
  const WideLds L(dim, K, (int)sizeof(Real), tile_bytes, lds_budget);
  Real* sh = (Real*)(lds + L.sh);
  uint32_t* lbf = (uint32_t*)(lds + L.lbf);        // W row (in elements) chosen by every cart of the stage
  Real* lsc = (Real*)(lds + L.lsc);                // its leaf score
  Real* lth = (Real*)(lds + L.lth);                // the cart's threshold (the replaying wave reads LDS only)
  uint8_t* lfi = lds + L.lfi;                      // its leaf index | 0x80 where the cart normalises the score
  int* misc = (int*)(lds + L.misc);                // [0] reject position of the stage (-1: passed), [1] output slot
  int* stage_cnt = misc + 16;
  Real* rows = (Real*)(lds + L.rows);
  uint8_t* tile = lds + L.tile;
  const int RC = L.row_cap;

#ifdef JDA_SCAN_TIMING
  unsigned long long stamps[15];
  int n_stamp = 0, dbg_win = 0, dbg_stages = 0;
#define JDA_WSTAMP() do { if (n_stamp < 15) stamps[n_stamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define JDA_WSTAMP() do { } while (0)
#endif
  if (tid < kMaxStages) stage_cnt[tid] = 0;
  const unsigned n = (unsigned)min(w.counters[kCntTail], (unsigned long long)w.cap_q);
  unsigned long long carts_acc = 0;

  for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
    JDA_WSTAMP();
    const uint32_t gid = w.q_gid[i];
    Real score = w.q_score[i];                     // (kept by every thread; only wave 0 advances it, then shares it)
    const int kstart = (int)w.q_kstart[i];
    unsigned hash = kFnvSeed;
    if (TRACE) hash = w.q_hash[i];
    int win;
    View v0{}, v1{}, v2{};
    decode_window<Real>(plan, w, w.q_xy[i], w.q_wf[i], 0.f, &win, &v0, &v1, &v2, false);
    const S0Node* s0_tbl = nullptr;
    if (s0_table != nullptr) {
      const bool hit = lane < plan->n_levels && plan->lv[lane].win == win;
      const unsigned long long mh = __ballot(hit);
      if (mh) {
        const DevLevel lv = plan->lv[__ffsll((long long)mh) - 1];
        if (lv.tiled) s0_tbl = s0_table + lv.s0_table;
      }
    }
    const uint8_t* wbase = v0.img + (size_t)v0.oy * v0.w + v0.ox;
    const bool use_tile = win <= tile_win;
    const int tpitch = (win + 3) & ~3;
    __syncthreads();                               // the previous window's readers are done with LDS
    if (use_tile) load_window_tile(wbase, v0.w, win, tile, tpitch, tid, BLOCK, v0.bc);
    for (int d = tid; d < dim; d += BLOCK) sh[d] = m.mean_shape[d];
    __syncthreads();
    JDA_WSTAMP();
#ifdef JDA_SCAN_TIMING
    dbg_win = win;
#endif

    bool alive = true;
    int carts_n = 0;
    Stp<Real> stp;
    stp.scale = 1; stp.r00 = 1; stp.r01 = 0; stp.r10 = 0; stp.r11 = 1;
    for (int t = 0; t < T && alive; t++) {
      const NodeOff<Real>* n_off = (const NodeOff<Real>*)m.lm_off + (size_t)t * K * node_n;
      const uint2* n_meta = m.lm_meta + (size_t)t * K * node_n;
      const int n_split = m.lm_split;
      const typename DL::Node* n_deep = (const typename DL::Node*)m.lm_deep + (size_t)t * K * (node_n - ((1 << min(n_split, m.D - 1)) - 1));
      const Real* leaf_tab = m.leaf + (size_t)t * K * leaf_n;
      const Real* cth = m.cth + (size_t)t * K;
      const Real* cmean = m.cmean + (size_t)t * K;
      const Real* cstd = m.cstd + (size_t)t * K;
      const uint8_t* cnorm = m.cnorm + (size_t)t * K;
      const int kbeg = t == 0 ? min(kstart, K) : 0;    // first cart whose score is still to be applied
      // ---- every tree of the stage at once, thread = cart ----
      for (int k0 = 0; k0 < K; k0 += BLOCK) {
        const int k = k0 + tid;
        int kk[1], lf[1];
        kk[0] = min(k, K - 1);
        if (t == 0 && s0_tbl && use_tile) walk_carts_s0<1, true>(s0_tbl, K, kk, m.D, node_n, tile, tpitch, lf, Bc(0, (long long)tpitch * win));
        else if (t == 0 && s0_tbl) walk_carts_s0<1, false>(s0_tbl, K, kk, m.D, node_n, wbase, v0.w, lf, v0.bc);
        else if (use_tile) walk_carts<DL, 1, false, false, true>(n_off, n_meta, K, kk, m.D, node_n, sh, win, v0, v1, v2, stp, false, lf, tile, tpitch, n_deep, n_split);
        else walk_carts<DL, 1, false, false>(n_off, n_meta, K, kk, m.D, node_n, sh, win, v0, v1, v2, stp, false, lf, nullptr, 0, n_deep, n_split);
        if (k < K) {
          lbf[k] = (uint32_t)(k * leaf_n + lf[0]) * (uint32_t)dim;
          lsc[k] = leaf_tab[(unsigned)(k * leaf_n + lf[0])];
          lth[k] = cth[k];
          lfi[k] = (uint8_t)(lf[0] | (cnorm[k] ? 0x80 : 0));        // leaf_n <= 128 (finish_wide_ok)
        }
      }
      if (tid == 0) misc[0] = -1;
      __syncthreads();
      if (t <= 1) JDA_WSTAMP();
      // ---- the two dependent chains of a stage side by side (conc; r05): wave 0 replays the score recurrence
      //      (c/jda.c:395-399; ~23 k clocks for 540 carts) WHILE waves 1.. add the stage's K weight rows to the shape in
      //      cart order (c/jda.c:404-411; lane = coordinate, rows straight from L2 into registers, one chunk of loads in
      //      flight behind the chunk being added).  Neither needs the other: both follow from the leaves of the walks.
      //      A window the replay rejects just drops the sums.  (Before: replay, then row fetch into LDS by all threads,
      //      then the adds by wave 0 -- 23 k + 11 k + 15 k clocks per stage in sequence, profiles/r03_single_frame_*.) ----
      if (conc) {
        const int n_add = (dim + 63) >> 6;
        const Real* wt = m.w + (size_t)t * K * leaf_n * dim;
        Real a = 0;
        const int d = (wv - 1) * 64 + lane;
        const bool adder = wv >= 1 && wv <= n_add;
        const bool act = adder && d < dim;
        if (wv == 0) {
          int rej = -1;
          for (int kg = kbeg & ~63; kg < K && rej < 0; kg += 64) {
            const int k = kg + lane;
            Real ls = 0, thk = 0, mk = 0, sk = 1;
            int nrm = 0, lf = 0;
            if (k < K) {
              ls = lsc[k]; thk = lth[k]; lf = lfi[k]; nrm = lf >> 7; lf &= 0x7f;
              if (nrm) { mk = cmean[k]; sk = cstd[k]; }             // (rare: every 10*L-th cart, btcart.cpp:173-181)
            }
            const unsigned long long normmask = __ballot(nrm != 0);
            const int jr = replay_scores<Real, TRACE>(score, hash, ls, thk, mk, sk, normmask, lf, max(0, kbeg - kg), min(64, K - kg));
            if (jr >= 0) rej = kg + jr;
          }
          if (lane == 0) { misc[0] = rej; *(Real*)(misc + 2) = score; if (TRACE) misc[4] = (int)hash; }
        } else if (adder) {
          const int dc = act ? d : dim - 1;                      // (idle lanes of the last adding wave repeat a coordinate)
          a = kCpp ? (Real)0 : sh[dc];
          const Real* col = wt + dc;
          constexpr int CH = 24;
          int k = 0;
          if (K >= CH) {
            Real x[CH], y[CH];
#pragma unroll
            for (int q = 0; q < CH; q++) { JDA_BC(Bc(0, (long long)K * leaf_n * dim), (long long)lbf[q] + dc, 1, kBcWRow); x[q] = col[lbf[q]]; }
            for (k = CH; k + CH <= K; k += CH) {
#pragma unroll
              for (int q = 0; q < CH; q++) { JDA_BC(Bc(0, (long long)K * leaf_n * dim), (long long)lbf[k + q] + dc, 1, kBcWRow); y[q] = col[lbf[k + q]]; }
#pragma unroll
              for (int q = 0; q < CH; q++) a = a + x[q];          // c/jda.c:404-411, in cart order
#pragma unroll
              for (int q = 0; q < CH; q++) x[q] = y[q];
            }
#pragma unroll
            for (int q = 0; q < CH; q++) a = a + x[q];
          }
          for (; k < K; k++) a = a + col[lbf[k]];
        }
        __syncthreads();
        const int rej = misc[0];
        score = *(const Real*)(misc + 2);
        if (TRACE) hash = (unsigned)misc[4];
        if (t <= 1) JDA_WSTAMP();
        if (rej >= 0) { alive = false; carts_n = t * K + rej + 1; break; }
        if (adder) {
          if (kCpp) {
            // stp_mc.Apply(delta, delta) with the identity parameter, literally (btcart.cpp:422, data.hpp:42-45), on the
            // (dx, dy) pair held by lanes d, d^1 (same wave: dim is even)
            const Real other = __shfl_xor(a, 1);
            if (act) {
              a = (d & 1) ? stp.scale * (stp.r10 * other + stp.r11 * a) : stp.scale * (stp.r00 * a + stp.r01 * other);
              a = sh[d] + a;
            }
          }
          if (act) sh[d] = a;
        }
        if (tid == 0) stage_cnt[t] += 1;
        __syncthreads();
        JDA_WSTAMP();
#ifdef JDA_SCAN_TIMING
        dbg_stages = t + 1;
#endif
        continue;
      }
      // ---- score recurrence, strictly in cart order, by wave 0 (64 carts per block: per-lane values, v_readlane) ----
      if (wv == 0) {
        int rej = -1;
        for (int kg = kbeg & ~63; kg < K && rej < 0; kg += 64) {
          const int k = kg + lane;
          Real ls = 0, thk = 0, mk = 0, sk = 1;
          int nrm = 0, lf = 0;
          if (k < K) {
            ls = lsc[k]; thk = lth[k]; lf = lfi[k]; nrm = lf >> 7; lf &= 0x7f;
            if (nrm) { mk = cmean[k]; sk = cstd[k]; }             // (rare: every 10*L-th cart, btcart.cpp:173-181)
          }
          const unsigned long long normmask = __ballot(nrm != 0);
          const int jr = replay_scores<Real, TRACE>(score, hash, ls, thk, mk, sk, normmask, lf, max(0, kbeg - kg), min(64, K - kg));
          if (jr >= 0) rej = kg + jr;
        }
        if (lane == 0) { misc[0] = rej; *(Real*)(misc + 2) = score; if (TRACE) misc[4] = (int)hash; }
      }
      __syncthreads();
      const int rej = misc[0];
      score = *(const Real*)(misc + 2);
      if (TRACE) hash = (unsigned)misc[4];
      if (t <= 1) JDA_WSTAMP();
      if (rej >= 0) { alive = false; carts_n = t * K + rej + 1; break; }
      // ---- stage regression: the K weight rows -> LDS (all loads in flight), then added strictly in cart order ----
      const Real* wt = m.w + (size_t)t * K * leaf_n * dim;
      Real acc[4];                                   // coordinates tid, tid + BLOCK, ... (dim <= 4 * BLOCK)
#pragma unroll
      for (int u = 0; u < 4; u++) acc[u] = (tid + u * BLOCK < dim) ? (kCpp ? (Real)0 : sh[tid + u * BLOCK]) : (Real)0;
      for (int r0 = 0; r0 < K; r0 += RC) {
        const int rn = min(RC, K - r0);
        const int total = rn * dim;
        if (sizeof(Real) == 4) {
          // LDS-DMA: a wave instruction brings 64 consecutive elements of the row buffer, 4 bytes per lane, straight
          // from the rows in L2 (no VGPR round trip: every load of the chunk is in flight before the one wait)
          typedef __attribute__((address_space(3))) void* lds_ptr_t;
          typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
          int e = wv * 64 + lane;
          int r = e / dim, d = e - r * dim;
          const int dr = BLOCK / dim, dd = BLOCK - dr * dim;
          for (int base = wv * 64; base < total; base += BLOCK * 8) {
            // the row offsets of 8 steps first (LDS reads), then the 8 loads back to back
            unsigned off[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              off[u] = (e + u * BLOCK < total) ? lbf[r0 + r] + (unsigned)d : 0u;
              r += dr; d += dd;
              if (d >= dim) { d -= dim; r++; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
              if (e + u * BLOCK < total) { JDA_BC(Bc(0, (long long)K * leaf_n * dim), off[u], 1, kBcWRow); JDA_BC(Bc(0, (long long)RC * dim), base + u * BLOCK + (tid & 63), 1, kBcWRow); }
#pragma unroll
            for (int u = 0; u < 8; u++)
              if (e + u * BLOCK < total)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wt + off[u]), (lds_ptr_t)((unsigned char*)rows + (size_t)(base + u * BLOCK) * 4), 4, 0, 0);
            e += 8 * BLOCK;
          }
          __builtin_amdgcn_s_waitcnt(0);
        } else {
          int e = tid;
          int r = e / dim, d = e - r * dim;
          const int dr = BLOCK / dim, dd = BLOCK - dr * dim;
          for (int e0 = 0; e0 < total; e0 += BLOCK * 8) {
            Real v[8];
            int rr[8], ddv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              rr[u] = r; ddv[u] = d;
              if (e0 + u * BLOCK + tid < total) { JDA_BC(Bc(0, (long long)K * leaf_n * dim), (long long)lbf[r0 + r] + d, 1, kBcWRow); v[u] = wt[lbf[r0 + r] + d]; }
              r += dr; d += dd;
              if (d >= dim) { d -= dim; r++; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
              if (e0 + u * BLOCK + tid < total) { JDA_BC(Bc(0, (long long)RC * dim), rr[u] * dim + ddv[u], 1, kBcWRow); rows[rr[u] * dim + ddv[u]] = v[u]; }
          }
        }
        __syncthreads();
        if (t <= 1) JDA_WSTAMP();
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int d = tid + u * BLOCK;
          if (d < dim) {
            Real a = acc[u];
            const Real* col = rows + d;
            int r = 0;
            if (rn >= 12) {
              // 12 LDS reads in flight (the LDS counter holds 15) while the previous 12 are added: the adds are a
              // dependent chain anyway, the reads hide behind it (27 clocks per row measured; forcing the reads ahead
              // of the adds with scheduling barriers: 30)
              Real x[12], y[12];
#pragma unroll
              for (int q = 0; q < 12; q++) x[q] = col[q * dim];
              for (r = 12; r + 12 <= rn; r += 12) {
#pragma unroll
                for (int q = 0; q < 12; q++) y[q] = col[(r + q) * dim];
#pragma unroll
                for (int q = 0; q < 12; q++) a = a + x[q];        // c/jda.c:404-411, in cart order
#pragma unroll
                for (int q = 0; q < 12; q++) x[q] = y[q];
              }
#pragma unroll
              for (int q = 0; q < 12; q++) a = a + x[q];
            }
            for (; r < rn; r++) a = a + col[r * dim];
            acc[u] = a;
          }
        }
        __syncthreads();
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int d = tid + u * BLOCK;
        Real a = acc[u];
        if (kCpp) {
          // stp_mc.Apply(delta, delta) with the identity parameter, literally (btcart.cpp:422, data.hpp:42-45), on the
          // (dx, dy) pair held by lanes d, d^1 (same wave: BLOCK is a multiple of 64 and dim is even)
          const Real other = __shfl_xor(a, 1);
          if (d < dim) {
            a = (d & 1) ? stp.scale * (stp.r10 * other + stp.r11 * a) : stp.scale * (stp.r00 * a + stp.r01 * other);
            a = sh[d] + a;
          }
        }
        if (d < dim) sh[d] = a;
      }
      if (tid == 0) stage_cnt[t] += 1;
      __syncthreads();
      JDA_WSTAMP();
#ifdef JDA_SCAN_TIMING
      dbg_stages = t + 1;
#endif
    }

    // ---- the window's walk is over: account for it (reference counting: Validate's n), final cut, emit ----
    if (alive) carts_n = T * K;
    if (tid == 0) carts_acc += (unsigned long long)carts_n;
    if (TRACE) {
      if (tid == 0) { w.tr_carts[gid] = carts_n; w.tr_score[gid] = score; w.tr_hash[gid] = hash; }
      for (int d = tid; d < dim; d += BLOCK) w.tr_shape[(size_t)gid * dim + d] = sh[d];
    }
    if (alive && !(apply_th && score < final_th)) {              // c/jda.c:414
      if (tid == 0) misc[1] = (int)atomicAdd(&w.counters[kCntOut], 1ull);
      __syncthreads();
      const unsigned o = (unsigned)misc[1];
      if (o < w.cap_m) {
        if (tid == 0) { w.out_gid[o] = gid; w.out_score[o] = score; }
        for (int d = tid; d < dim; d += BLOCK) w.out_shape[(size_t)o * dim + d] = sh[d];
      }
    }
  }
#ifdef JDA_SCAN_TIMING
  JDA_WSTAMP();
  if (tid == 0 && w.dbg && blockIdx.x < 65536 && blockIdx.x < n) {
    unsigned long long* o = w.dbg + (size_t)blockIdx.x * 32;
    o[0] = (unsigned long long)n_stamp | (0x8888ull << 32);
    for (int q = 0; q < n_stamp; q++) o[1 + q] = stamps[q];
    o[16] = (unsigned long long)dbg_win; o[17] = (unsigned long long)dbg_stages;
  }
#endif
  __syncthreads();
  if (tid < T && stage_cnt[tid]) atomicAdd(shard_counter(w.counters, kCntStage0 + tid), (unsigned long long)stage_cnt[tid]);
  if (tid == 0 && carts_acc) atomicAdd(shard_counter(w.counters, kCntCarts), carts_acc);
  (void)NW;
}

namespace {
template <typename DL>
hipError_t launch_finish_wide_impl(bool trace, bool apply_th, typename DL::Real th, const DevPlan* d_plan,
                                   const DevModelT<typename DL::Real>& m, const WorkT<typename DL::Real>& w,
                                   long long n_hint, const S0Node* s0_table, hipStream_t stream, bool conc) {
  using Real = typename DL::Real;
  const int budget = 160 * 1024;
  // window tile: as large as leaves room for at least 64 weight rows
  int tile_win = 0;
  for (int tw = 16; tw <= 255; tw++) {
    const WideLds L(m.dim, m.K, (int)sizeof(Real), tw * ((tw + 3) & ~3) + 16, budget);
    if (L.row_cap >= std::min(m.K, 64) && tw * ((tw + 3) & ~3) + 16 <= 40 * 1024) tile_win = tw;
  }
  const int tile_bytes = tile_win > 0 ? tile_win * ((tile_win + 3) & ~3) + 16 : 0;
  const WideLds L(m.dim, m.K, (int)sizeof(Real), tile_bytes, budget);
  if (L.row_cap < 1 || L.total > budget) return hipErrorInvalidValue;
  const unsigned blocks = (unsigned)std::max<long long>(1, std::min<long long>(n_hint, 1 << 16));
  const int block = m.K > 512 ? 1024 : (m.K > 256 ? 512 : 256);
  // replay and regression side by side: a wave for the replay + one per 64 shape coordinates
  const bool conc_ok = conc && 1 + (m.dim + 63) / 64 <= block / 64;
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(block), L.total, stream, d_plan, m, w, apply_th ? 1 : 0, th, s0_table,
                       tile_win, tile_bytes, budget, conc_ok ? 1 : 0);
  };
  auto pick = [&](auto trace_tag) {
    constexpr bool TR = decltype(trace_tag)::value;
    if (block == 1024) go(k_finish_wide<DL, TR, 1024>);
    else if (block == 512) go(k_finish_wide<DL, TR, 512>);
    else go(k_finish_wide<DL, TR, 256>);
  };
  if (trace) pick(std::true_type{}); else pick(std::false_type{});
  return hipGetLastError();
}
}  // namespace

bool finish_wide_ok(int dim, int K, int leaf_n, int real_bytes, bool multi, bool similarity) {
  if (multi || similarity || dim > 4 * 256 || leaf_n > 128) return false;
  const WideLds L(dim, K, real_bytes, 0, 160 * 1024);
  return L.row_cap >= 1;
}

template <>
hipError_t launch_finish_wide<float>(bool trace, bool apply_final_th, float final_th, const DevPlan* d_plan,
                                     const DevModelT<float>& m, const WorkT<float>& w, long long n_hint,
                                     const S0Node* s0_table, hipStream_t stream, bool conc) {
  return launch_finish_wide_impl<DialectC>(trace, apply_final_th, final_th, d_plan, m, w, n_hint, s0_table, stream, conc);
}
template <>
hipError_t launch_finish_wide<double>(bool trace, bool apply_final_th, double final_th, const DevPlan* d_plan,
                                      const DevModelT<double>& m, const WorkT<double>& w, long long n_hint,
                                      const S0Node* s0_table, hipStream_t stream, bool conc) {
  return launch_finish_wide_impl<DialectCPP>(trace, apply_final_th, final_th, d_plan, m, w, n_hint, s0_table, stream, conc);
}

JDA_BC_READER(k_wide)

}  // namespace jda
