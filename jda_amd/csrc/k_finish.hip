// k_finish: wave = window, for every survivor of the scan: lanes = carts for a stage's tree walks
// (c/jda.c:366-400, cart.cpp:392-404), the score recurrence replayed in cart order (c/jda.c:395-399),
// then lanes = shape coordinates for the regression gather in cart order (c/jda.c:404-411,
// btcart.cpp:407-424), final cut (c/jda.c:414) and emit.
#include "finish_common.h"

#ifndef FIN_ROW_BATCH
#define FIN_ROW_BATCH 0        // (experiment builds: -DFIN_ROW_BATCH=n forces the regression's row batch)
#endif

namespace jda {

// =============================================================================
// k_finish: one wave per surviving window
// =============================================================================

// Stages [t_begin, t_end) for every window of the input queue.  Windows that are
// still alive after stage t_end-1 go to the mid queue (t_end < T) or, after the
// final threshold, to the detection list (t_end == T).
// (One-wave workgroups: the hardware keeps at most 16 workgroups on a CU, so a CU works on 16 windows at a
// time.  Workgroups of 2-4 independent waves lift that to the register limit, measured: no gain -- the launches
// are bound by the CU's texture addresser / L1 (TA_BUSY 70-75 % on average, 95 % on the busiest CU with 16
// windows), not by the number of windows in flight.)
template <typename DL, bool TRACE, int kG, bool MULTI, bool ST, bool STREAM = false>
__global__ __launch_bounds__(64) void k_finish(const DevPlan* __restrict__ plan, DevModelT<typename DL::Real> m,
                                               WorkT<typename DL::Real> w, int multi_i, float inv_sqrt2,
                                               int t_begin, int t_end, int apply_th, typename DL::Real final_th,
                                               const S0Node* __restrict__ s0_table, int tile_win, int survivors) {
  using Real = typename DL::Real;
  constexpr bool kCpp = sizeof(Real) == 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x;
  const int T = m.T, K = m.K, node_n = m.node_n, leaf_n = m.leaf_n, dim = m.dim, w_pitch = m.w_pitch;
  const int dim_pad = (dim + 1) & ~1;
  Real* sh = (Real*)lds;                                     // current shape        [dim_pad] (the regression updates it
                                                             // in place: coordinate d is read and written by one lane only)
  uint32_t* lbf = (uint32_t*)(sh + dim_pad);                 // W row (in elements) chosen by every cart [K] (kept as the
                                                             // finished offset: 16-bit leaf indices and the multiply in the
                                                             // regression loop cost 37 us per step)
  int* stage_cnt = (int*)(lbf + ((K + 3) & ~3));             // per-block stage counters
  Real* st_tmp = (Real*)(stage_cnt + kMaxStages);            // similarity-transform scratch [2*dim_pad + 8] (ST only)
  // the window's pixels (windows up to tile_win pixels, single-scale models), behind the scratch, 16-byte aligned
  uint8_t* tile = lds + (((size_t)((unsigned char*)(st_tmp + (ST ? 2 * dim_pad + 8 : 0)) - lds) + 15) & ~(size_t)15);
  constexpr bool multi = MULTI;   // split nodes read the half/quarter images too
  (void)multi_i;
  if (lane < kMaxStages) stage_cnt[lane] = 0;
  // the input queue: the hand-off queue of k_scan (t_begin == 0), the mid queue of an earlier k_finish launch
  // (t_begin > 0), or -- survivors -- the mid queue as k_filter0 leaves it: windows that passed every cart of stage 0,
  // still holding the mean shape (their stage-0 leaves are walked again here, no score is applied)
  const bool from_scan = t_begin == 0 && !survivors;
  const unsigned n = (unsigned)min(w.counters[from_scan ? kCntTail : kCntMid], (unsigned long long)(from_scan ? w.cap_q : w.cap_m));
  unsigned long long carts_acc = 0;

#ifdef JDA_SCAN_TIMING
  unsigned long long stamps[15];
  int n_stamp = 0;
  int dbg_win = 0;
#define JDA_FSTAMP() do { if (n_stamp < 15) stamps[n_stamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define JDA_FSTAMP() do { } while (0)
#endif
  for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
    JDA_FSTAMP();
    const uint32_t gid = from_scan ? w.q_gid[i] : w.m_gid[i];
    Real score = from_scan ? w.q_score[i] : w.m_score[i];
    const int kstart = from_scan ? (int)w.q_kstart[i] : (survivors ? K : 0);
    unsigned hash = kFnvSeed;
    if (TRACE) hash = from_scan ? w.q_hash[i] : w.m_hash[i];
    int win;
    View v0{}, v1{}, v2{};
    const uint32_t xy = from_scan ? w.q_xy[i] : w.m_xy[i];
    const uint32_t wf = from_scan ? w.q_wf[i] : w.m_wf[i];
    decode_window<Real>(plan, w, xy, wf, inv_sqrt2, &win, &v0, &v1, &v2, multi);
    // stage 0 of a window whose level has resolved tables (every level k_scan covers): walk from them
    const S0Node* s0_tbl = nullptr;
    if (!MULTI && s0_table != nullptr && t_begin == 0) {
      const bool hit = lane < plan->n_levels && plan->lv[lane].win == win;
      const unsigned long long mh = __ballot(hit);
      if (mh) {
        const DevLevel lv = plan->lv[__ffsll((long long)mh) - 1];
        if (lv.tiled) s0_tbl = s0_table + lv.s0_table;
      }
    }
    const uint8_t* wbase = v0.img + (size_t)v0.oy * v0.w + v0.ox;
#ifdef JDA_SCAN_TIMING
    dbg_win = win;
#endif
    const bool use_tile = !MULTI && win <= tile_win;
    const int tpitch = (win + 3) & ~3;
    __syncthreads();                       // previous window's readers are done with sh (and the tile)
    if (!MULTI && use_tile) load_window_tile(wbase, v0.w, win, tile, tpitch, lane, 64, v0.bc);
    {
      const Real* src = t_begin == 0 ? m.mean_shape : w.m_shape + (size_t)i * dim;
      for (int d = lane; d < dim; d += 64) sh[d] = src[d];
    }
    __syncthreads();
    JDA_FSTAMP();

    bool alive = true;
    int carts_n = 0;
    for (int t = t_begin; t < t_end; t++) {
      const NodeOff<Real>* n_off = (const NodeOff<Real>*)m.lm_off + (size_t)t * K * node_n;
      const uint2* n_meta = m.lm_meta + (size_t)t * K * node_n;
      const int n_split = m.lm_split;
      const typename DL::Node* n_deep = (const typename DL::Node*)m.lm_deep + (size_t)t * K * (node_n - ((1 << min(n_split, m.D - 1)) - 1));
      const Real* leaf_tab = m.leaf + (size_t)t * K * leaf_n;
      const Real* cth = m.cth + (size_t)t * K;
      const Real* cmean = m.cmean + (size_t)t * K;
      const Real* cstd = m.cstd + (size_t)t * K;
      const uint8_t* cnorm = m.cnorm + (size_t)t * K;
      const int kbeg = t == 0 ? min(kstart, K) : 0;   // first cart whose score is still to be applied
      const int k_first = kbeg & ~63;
      // similarity transform of this stage (cascador.cpp:180); identity unless enabled
      Stp<Real> stp;
      stp.scale = 1; stp.r00 = 1; stp.r01 = 0; stp.r10 = 0; stp.r11 = 1;
      if constexpr (ST) {
        // m.similarity - 2: the stage in training of a trainer snapshot (negative: a complete model).  Validate does not
        // recompute stp_mc for that stage (cascador.cpp:178-200): it walks with the parameter the PREVIOUS stage computed --
        // which the scratch still holds -- or, being the first stage this window runs (the host keeps such a model's
        // stages in one launch, so that is stage 0), with STParameter's default.
        const bool stale = t == m.similarity - 2;
        if (lane == 0 && (!stale || t == t_begin)) {
          Stp<double> p;
          p.scale = 1.; p.r00 = 1.; p.r01 = 0.; p.r10 = 0.; p.r11 = 1.;
          if (!stale) p = stp_calc((const double*)sh, (const double*)m.mean_shape_raw, m.L, (double*)st_tmp, (double*)st_tmp + dim_pad);
          double* o = (double*)st_tmp + 2 * dim_pad;
          o[0] = p.scale; o[1] = p.r00; o[2] = p.r01; o[3] = p.r10; o[4] = p.r11;
        }
        __syncthreads();
        const Real* o = st_tmp + 2 * dim_pad;
        stp.scale = o[0]; stp.r00 = o[1]; stp.r01 = o[2]; stp.r10 = o[3]; stp.r11 = o[4];
      }
      const bool apply_st = ST && t > 0;              // stage 0's node offsets carry the transform already

      // ---- tree walks, kG groups of 64 carts per round (the shape is fixed during a
      //      stage, so the trees of a stage are independent of each other and of the
      //      score); then the score recurrence replayed in cart order ----
      for (int k0 = k_first; k0 < K && alive; k0 += 64 * kG) {
        int kk[kG], lf[kG], nrm[kG];
        Real ls[kG], thk[kG], mk[kG], sk[kG];
#pragma unroll
        for (int g = 0; g < kG; g++) kk[g] = min(k0 + g * 64 + lane, K - 1);   // clamped lanes repeat cart K-1
        if (t == 0 && s0_tbl && use_tile) walk_carts_s0<kG, true>(s0_tbl, K, kk, m.D, node_n, tile, tpitch, lf, Bc(0, (long long)tpitch * win));
        else if (t == 0 && s0_tbl) walk_carts_s0<kG, false>(s0_tbl, K, kk, m.D, node_n, wbase, v0.w, lf, v0.bc);
        else if (!MULTI && use_tile) walk_carts<DL, kG, MULTI, ST, true>(n_off, n_meta, K, kk, m.D, node_n, sh, win, v0, v1, v2, stp, apply_st, lf, tile, tpitch, n_deep, n_split);
        else walk_carts<DL, kG, MULTI, ST>(n_off, n_meta, K, kk, m.D, node_n, sh, win, v0, v1, v2, stp, apply_st, lf, nullptr, 0, n_deep, n_split);
#pragma unroll
        for (int g = 0; g < kG; g++) {
          const int k = k0 + g * 64 + lane;
          ls[g] = 0; thk[g] = 0; mk[g] = 0; sk[g] = 1; nrm[g] = 0;
          if (k < K) {
            lbf[k] = (uint32_t)(k * leaf_n + lf[g]) * (uint32_t)w_pitch;
            ls[g] = leaf_tab[(unsigned)(k * leaf_n + lf[g])];
            thk[g] = cth[k];
            nrm[g] = cnorm[k];
            if (nrm[g]) { mk[g] = cmean[k]; sk[g] = cstd[k]; }
          }
        }
#pragma unroll
        for (int g = 0; g < kG; g++) {
          const int kg = k0 + g * 64;
          if (kg >= K || !alive) break;
          const unsigned long long normmask = __ballot(nrm[g] != 0);
          const int jr = replay_scores<Real, TRACE>(score, hash, ls[g], thk[g], mk[g], sk[g], normmask, lf[g],
                                                    max(0, kbeg - kg), min(64, K - kg));
          if (jr >= 0) { alive = false; carts_n = t * K + kg + jr + 1; }
        }
        if (t == t_begin) JDA_FSTAMP();
      }
      if (t != t_begin) JDA_FSTAMP();
      if (!alive) break;
      // leaves of the carts k_scan already scored (needed only now that the stage is passed)
      for (int k0 = 0; k0 < k_first; k0 += 128) {
        int kk[2], lf[2];
        kk[0] = min(k0 + lane, k_first - 1); kk[1] = min(k0 + 64 + lane, k_first - 1);
        if (t == 0 && s0_tbl && use_tile) walk_carts_s0<2, true>(s0_tbl, K, kk, m.D, node_n, tile, tpitch, lf, Bc(0, (long long)tpitch * win));
        else if (t == 0 && s0_tbl) walk_carts_s0<2, false>(s0_tbl, K, kk, m.D, node_n, wbase, v0.w, lf, v0.bc);
        else if (!MULTI && use_tile) walk_carts<DL, 2, MULTI, ST, true>(n_off, n_meta, K, kk, m.D, node_n, sh, win, v0, v1, v2, stp, apply_st, lf, tile, tpitch, n_deep, n_split);
        else walk_carts<DL, 2, MULTI, ST>(n_off, n_meta, K, kk, m.D, node_n, sh, win, v0, v1, v2, stp, apply_st, lf, nullptr, 0, n_deep, n_split);
        if (k0 + lane < k_first) lbf[k0 + lane] = (uint32_t)((k0 + lane) * leaf_n + lf[0]) * (uint32_t)w_pitch;
        if (k0 + 64 + lane < k_first) lbf[k0 + 64 + lane] = (uint32_t)((k0 + 64 + lane) * leaf_n + lf[1]) * (uint32_t)w_pitch;
      }
      __syncthreads();
      if (t == t_begin) JDA_FSTAMP();
      // ---- stage regression: K weight rows added strictly in cart order
      //      (c/jda.c:404-411); dialect CPP sums the delta from zero and adds it
      //      once (btcart.cpp:407-424) ----
      const Real* wt = m.w_rows + (size_t)t * K * leaf_n * w_pitch;
      for (int d = lane; d < dim; d += 64) {
        Real acc = kCpp ? (Real)0 : sh[d];
        const Real* col = wt + d;
        int k = 0;
#ifdef JDA_EXP_NOREG
        k = K;      // (experiment build, never the product: how long do the walks alone take?)
#endif
        // RB row loads in flight, then RB ordered adds.  A window's stage is a chain of K / RB such round trips to L2 --
        // the longest part of its latency when few windows are in flight (a rank's shard, a single frame).  fp32: 64 (the
        // registers are there: 16 one-wave workgroups per CU is the LDS limit, 4 waves per SIMD); fp64, whose rows are two
        // registers per element: 32.
        constexpr int RB = FIN_ROW_BATCH > 0 ? FIN_ROW_BATCH : (kCpp ? 32 : 64);
        for (; k + RB <= K; k += RB) {
          Real r[RB];
          // (STREAM is a template parameter: as a run-time branch the optimiser merges the two forms of the load and
          // drops the non-temporal hint)
#pragma unroll
          for (int u = 0; u < RB; u++) { JDA_BC(Bc(0, (long long)K * leaf_n * w_pitch), (long long)lbf[k + u] + d, 1, kBcWRow); r[u] = STREAM ? __builtin_nontemporal_load(col + lbf[k + u]) : col[lbf[k + u]]; }
#pragma unroll
          for (int u = 0; u < RB; u++) acc = acc + r[u];
        }
        for (; k < K; k++) { JDA_BC(Bc(0, (long long)K * leaf_n * w_pitch), (long long)lbf[k] + d, 1, kBcWRow); acc = acc + col[lbf[k]]; }
        if (kCpp) {
          // stp_mc.Apply(delta, delta) (btcart.cpp:422, data.hpp:42-45) on the (dx,dy) pair held by
          // lanes d, d^1; with the identity parameter this is the literal 1*(1*x+0*y) / 1*(0*x+1*y)
          const Real other = __shfl_xor(acc, 1);
          acc = (d & 1) ? stp.scale * (stp.r10 * other + stp.r11 * acc) : stp.scale * (stp.r00 * acc + stp.r01 * other);
          acc = sh[d] + acc;
        }
        sh[d] = acc;
      }
      __syncthreads();
      JDA_FSTAMP();
      if (lane == 0) stage_cnt[t] += 1;
    }

    if (!alive || t_end == T) {
      // the window's walk is over: account for it (reference counting: Validate's n)
      if (alive) carts_n = T * K;
      carts_acc += (unsigned long long)carts_n;
      if (TRACE) {
        if (lane == 0) { w.tr_carts[gid] = carts_n; w.tr_score[gid] = score; w.tr_hash[gid] = hash; }
        for (int d = lane; d < dim; d += 64) w.tr_shape[(size_t)gid * dim + d] = sh[d];
      }
      if (alive && !(apply_th && score < final_th)) {            // c/jda.c:414
        unsigned o = 0;
        if (lane == 0) o = (unsigned)atomicAdd(&w.counters[kCntOut], 1ull);
        o = (unsigned)__shfl((int)o, 0);
        if (o < w.cap_m) {
          if (lane == 0) { w.out_gid[o] = gid; w.out_score[o] = score; }
          for (int d = lane; d < dim; d += 64) w.out_shape[(size_t)o * dim + d] = sh[d];
        }
      }
    } else {
      // alive with stages left: park it in the mid queue for the next launch
      unsigned o = 0;
      if (lane == 0) o = (unsigned)atomicAdd(&w.counters[kCntMid], 1ull);
      o = (unsigned)__shfl((int)o, 0);
      if (o < w.cap_m) {
        if (lane == 0) { w.m_gid[o] = gid; w.m_score[o] = score; w.m_xy[o] = xy; w.m_wf[o] = wf; if (TRACE) w.m_hash[o] = hash; }
        for (int d = lane; d < dim; d += 64) w.m_shape[(size_t)o * dim + d] = sh[d];
      }
    }
  }
#ifdef JDA_SCAN_TIMING
  JDA_FSTAMP();
  if (lane == 0 && w.dbg && blockIdx.x < 65536 && (t_begin > 0 || survivors) && blockIdx.x < n) {
    unsigned long long* o = w.dbg + (size_t)blockIdx.x * 32;
    o[0] = (unsigned long long)n_stamp | (0x7777ull << 32);
    for (int i = 0; i < n_stamp; i++) o[1 + i] = stamps[i];
    o[16] = (unsigned long long)dbg_win;
    o[17] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 32);   // HW_ID, XCC_ID
  }
#endif
  __syncthreads();
  if (lane < T && stage_cnt[lane]) atomicAdd(shard_counter(w.counters, kCntStage0 + lane), (unsigned long long)stage_cnt[lane]);
  if (lane == 0 && carts_acc) atomicAdd(shard_counter(w.counters, kCntCarts), carts_acc);
}

// =============================================================================
// k_filter0: the rest of stage 0 for the hand-off queue, and nothing else
// =============================================================================
// 0.7 % of a batch's windows leave k_scan alive at cart `handoff`, and 93 % of those die before stage 0 ends, most of
// them within a round of 64 carts.  k_finish spends a one-wave workgroup with a stage's worth of state on each: the
// hardware holds 16 such workgroups per CU, and with ~8 us of dependent memory round trips per window (queue entry,
// three tree levels of table + pixel loads, leaf scores, replay) the launch is bound by latency x 16 windows per CU,
// not by any pipe (pass 1: 224 us for 70 k windows).  This kernel does only that filtering: wave = window, four
// independent waves per workgroup (no LDS, no barrier), stage-0 walks from the resolved tables with the pixels read
// from the frame, the systolic replay; few registers, so a CU holds several times as many windows.  A window that is
// rejected is final here (counters, trace); one that passes every cart of stage 0 goes to the mid queue with its
// score, and k_finish(survivors) takes it through the regression of stage 0 (re-walking its trees for the leaves) and
// the later stages.  Same walks, same replay: c/jda.c:366-400.
namespace {
// One depth-4 cart of stage 0 from the level-major table with ALL seven node records fetched up front: the root, the
// pair of its children (16 contiguous bytes) and the four grandchildren (32 contiguous bytes) of cart k sit at
// lm_index(K, k, d, .), so the four loads are independent of the walk and -- lanes = consecutive carts -- coalesced.
// Only the three pixel-pair reads stay dependent: 4 memory round trips per cart instead of 6.
__device__ __forceinline__ int walk_cart_s0_d4(const S0Node* __restrict__ tbl, unsigned K, unsigned k,
                                               const uint8_t* __restrict__ pix, int pitch, const Bc& bc = Bc()) {
  const S0Node r0 = tbl[k];
  const uint4 c1 = *(const uint4*)(tbl + K + 2u * k);                 // children of the root
  const uint4 c2a = *(const uint4*)(tbl + 3u * K + 4u * k);           // grandchildren 0, 1
  const uint4 c2b = *(const uint4*)(tbl + 3u * K + 4u * k + 2u);      // grandchildren 2, 3
  auto left = [&](uint32_t lo, uint32_t hi) {
    const unsigned p1 = lo & 0x3fffffu, p2 = __builtin_amdgcn_alignbit(hi, lo, 22) & 0x3fffffu;
    JDA_BC_ADDR(bc, pix + __umul24(p1 >> 11, (unsigned)pitch) + (p1 & 0x7ffu), 1, kBcFinishPix);
    JDA_BC_ADDR(bc, pix + __umul24(p2 >> 11, (unsigned)pitch) + (p2 & 0x7ffu), 1, kBcFinishPix);
    const int a = pix[__umul24(p1 >> 11, (unsigned)pitch) + (p1 & 0x7ffu)];
    const int b = pix[__umul24(p2 >> 11, (unsigned)pitch) + (p2 & 0x7ffu)];
    return a - b <= (int)((hi >> 12) & 0x3ffu) - 256;                 // c/jda.c:391-393
  };
  const bool l0 = left(r0.lo, r0.hi);                                 // left -> child 1 (index 0 of the pair), right -> child 2
  const uint32_t lo1 = l0 ? c1.x : c1.z, hi1 = l0 ? c1.y : c1.w;
  const bool l1 = left(lo1, hi1);
  // node after level 1: 2*node + (left ? 1 : 2); its record among the four grandchildren: index 2*(l0 ? 0 : 1) + (l1 ? 0 : 1)
  const uint32_t lo2 = l0 ? (l1 ? c2a.x : c2a.z) : (l1 ? c2b.x : c2b.z);
  const uint32_t hi2 = l0 ? (l1 ? c2a.y : c2a.w) : (l1 ? c2b.y : c2b.w);
  const bool l2 = left(lo2, hi2);
  const int n1 = l0 ? 1 : 2, n2 = 2 * n1 + (l1 ? 1 : 2), n3 = 2 * n2 + (l2 ? 1 : 2);
  return n3 - 7;
}
}  // namespace

template <typename DL, bool TRACE>
__global__ __launch_bounds__(256) void k_filter0(const DevPlan* __restrict__ plan, DevModelT<typename DL::Real> m,
                                                 WorkT<typename DL::Real> w, const S0Node* __restrict__ s0_table) {
  using Real = typename DL::Real;
  constexpr int NW = 2;                    // windows a wave walks side by side: their memory round trips overlap
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int K = m.K, node_n = m.node_n, leaf_n = m.leaf_n;
  const unsigned n = (unsigned)min(w.counters[kCntTail], (unsigned long long)w.cap_q);
  const Real* leaf_tab = m.leaf;
  const int W = plan->width;
#ifdef JDA_BOUNDS_CHECK
  const Bc bc_fr((long long)(uintptr_t)w.bc_lo, (long long)(uintptr_t)w.bc_hi);
#else
  const Bc bc_fr;
#endif
  unsigned long long carts_acc = 0;
  const unsigned waves = gridDim.x * 4u;
  for (unsigned i0 = blockIdx.x * 4u + (unsigned)wv; i0 < n; i0 += NW * waves) {
    unsigned idx[NW]; bool has[NW];
    uint32_t gid[NW], xy[NW], wf[NW];
    Real score[NW];
    int kbeg[NW], rej[NW];
    unsigned hash[NW];
    const uint8_t* wbase[NW];
    const S0Node* tbl[NW];
#pragma unroll
    for (int u = 0; u < NW; u++) {
      idx[u] = i0 + (unsigned)u * waves; has[u] = idx[u] < n;
      const unsigned i = has[u] ? idx[u] : i0;
      gid[u] = w.q_gid[i]; score[u] = w.q_score[i]; kbeg[u] = min((int)w.q_kstart[i], K);
      hash[u] = TRACE ? w.q_hash[i] : kFnvSeed;
      xy[u] = w.q_xy[i]; wf[u] = w.q_wf[i];
      rej[u] = has[u] ? -1 : 0;
    }
#pragma unroll
    for (int u = 0; u < NW; u++) {
      const int win = (int)(wf[u] & 0xffffu), frame = (int)(wf[u] >> 16);
      const uint8_t* img = w.img_off != nullptr ? w.frames + w.img_off[frame] : w.frames + (size_t)frame * w.frame_stride;
      wbase[u] = img + (size_t)(xy[u] >> 16) * W + (xy[u] & 0xffffu);
      tbl[u] = s0_table;
      const bool hit = lane < plan->n_levels && plan->lv[lane].win == win;
      const unsigned long long mh = __ballot(hit);
      if (mh) tbl[u] = s0_table + plan->lv[__ffsll((long long)mh) - 1].s0_table;   // (the host only runs this kernel when every level has a table)
    }
    // rounds of 64 carts, both windows in step while both are alive (a window's own cart range starts at its kbeg)
    int k0[NW];
#pragma unroll
    for (int u = 0; u < NW; u++) k0[u] = kbeg[u] & ~63;
    for (;;) {
      bool go[NW], any = false;
#pragma unroll
      for (int u = 0; u < NW; u++) { go[u] = rej[u] < 0 && k0[u] < K; any = any || go[u]; }
      if (!any) break;
      int lf[NW], nrm[NW];
      Real ls[NW], thk[NW], mk[NW], sk[NW];
#pragma unroll
      for (int u = 0; u < NW; u++) {
        const int k = k0[u] + lane;
        thk[u] = 0; nrm[u] = 0; ls[u] = 0; mk[u] = 0; sk[u] = 1; lf[u] = 0;
        if (go[u] && k < K) { thk[u] = m.cth[k]; nrm[u] = m.cnorm[k]; }   // independent of the walk: in flight with it
      }
#pragma unroll
      for (int u = 0; u < NW; u++) {
        if (!go[u]) continue;
        int kk[1], l1[1];
        kk[0] = min(k0[u] + lane, K - 1);
        if (m.D == 4) l1[0] = walk_cart_s0_d4(tbl[u], (unsigned)K, (unsigned)kk[0], wbase[u], W, bc_fr);
        else walk_carts_s0<1, false>(tbl[u], K, kk, m.D, node_n, wbase[u], W, l1, bc_fr);
        lf[u] = l1[0];
      }
#pragma unroll
      for (int u = 0; u < NW; u++) {
        const int k = k0[u] + lane;
        if (go[u] && k < K) {
          ls[u] = leaf_tab[(unsigned)(k * leaf_n + lf[u])];
          if (nrm[u]) { mk[u] = m.cmean[k]; sk[u] = m.cstd[k]; }
        }
      }
#pragma unroll
      for (int u = 0; u < NW; u++) {
        if (!go[u]) continue;
        const unsigned long long normmask = __ballot(nrm[u] != 0);
        const int jr = replay_scores<Real, TRACE>(score[u], hash[u], ls[u], thk[u], mk[u], sk[u], normmask, lf[u],
                                                  max(0, kbeg[u] - k0[u]), min(64, K - k0[u]));
        if (jr >= 0) rej[u] = k0[u] + jr;
        k0[u] += 64;
      }
    }
#pragma unroll
    for (int u = 0; u < NW; u++) {
      if (!has[u]) continue;
      if (rej[u] >= 0) {
        carts_acc += (unsigned long long)(rej[u] + 1);
        if (TRACE && lane == 0) { w.tr_carts[gid[u]] = rej[u] + 1; w.tr_score[gid[u]] = score[u]; w.tr_hash[gid[u]] = hash[u]; }
      } else {
        unsigned o = 0;
        if (lane == 0) o = (unsigned)atomicAdd(&w.counters[kCntMid], 1ull);
        o = (unsigned)__shfl((int)o, 0);
        if (o < w.cap_m && lane == 0) {
          w.m_gid[o] = gid[u]; w.m_score[o] = score[u]; w.m_xy[o] = xy[u]; w.m_wf[o] = wf[u];
          if (TRACE) w.m_hash[o] = hash[u];
        }
      }
    }
  }
  if (lane == 0 && carts_acc) atomicAdd(shard_counter(w.counters, kCntCarts), carts_acc);
}

template <typename Real>
hipError_t launch_filter0(bool trace, const DevPlan* d_plan, const DevModelT<Real>& m, const WorkT<Real>& w, long long n_hint,
                          const S0Node* s0_table, hipStream_t stream) {
  using DL = typename std::conditional<sizeof(Real) == 4, DialectC, DialectCPP>::type;
  const unsigned blocks = (unsigned)std::max<long long>(1, std::min<long long>((n_hint + 7) / 8, 1 << 20));   // 4 waves x 2 windows
  if (trace) hipLaunchKernelGGL((k_filter0<DL, true>), dim3(blocks), dim3(256), 0, stream, d_plan, m, w, s0_table);
  else hipLaunchKernelGGL((k_filter0<DL, false>), dim3(blocks), dim3(256), 0, stream, d_plan, m, w, s0_table);
  return hipGetLastError();
}
template hipError_t launch_filter0<float>(bool, const DevPlan*, const DevModelT<float>&, const WorkT<float>&, long long, const S0Node*, hipStream_t);
template hipError_t launch_filter0<double>(bool, const DevPlan*, const DevModelT<double>&, const WorkT<double>&, long long, const S0Node*, hipStream_t);

namespace {
template <typename DL>
hipError_t launch_finish_impl(bool trace, int t_begin, int t_end, bool apply_th, typename DL::Real th,
                              const DevPlan* d_plan, const DevModelT<typename DL::Real>& m,
                              const WorkT<typename DL::Real>& w, int groups, long long n_hint, const S0Node* s0_table,
                              int tile_win, bool survivors, hipStream_t stream) {
  using Real = typename DL::Real;
  const int dim_pad = (m.dim + 1) & ~1;
  const bool st = sizeof(Real) == 8 && m.similarity != 0;
  const int multi = (w.half != nullptr) ? 1 : 0;
  if (multi) tile_win = 0;
  const size_t base = (size_t)dim_pad * sizeof(Real) + (size_t)((m.K + 3) & ~3) * 4 + kMaxStages * sizeof(int) +
                      (st ? (2 * (size_t)dim_pad + 8) * sizeof(Real) : 0);
  // tile_win < 0: the largest window tile that keeps 16 workgroups on a CU.  Measured on MI355X with one-wave
  // workgroups (profiles/r02_finish_experiments.txt): up to 7,680 bytes of LDS per workgroup the launch time does
  // not depend on the LDS size, 7,712 bytes cost +17 %, 7,856 +23 % (16 x 7.5 KB = 120 KB) -- the launch is bound
  // by the texture addresser but still needs its 16 windows per CU.
  if (tile_win < 0) {
    tile_win = 0;
    for (int tw = 16; tw <= 255; tw++)
      if (base + (size_t)tw * ((tw + 3) & ~3) + 16 <= kFinishLdsPerGroup) tile_win = tw;
  }
  const size_t tile_bytes = tile_win > 0 ? (size_t)tile_win * ((tile_win + 3) & ~3) + 16 : 0;
  const size_t lds = base + tile_bytes;
  const float r = 1.f / sqrtf(2.f);
  // n_hint >= 0: the queue length is known on the host -> one window per workgroup (up to
  // 1M workgroups, grid-stride beyond), so the hardware dispatcher balances the very
  // uneven per-window cost; n_hint < 0: fixed grid, windows dealt round-robin.
  unsigned blocks = t_begin == 0 && !survivors ? w.cap_q : w.cap_m;
  if (blocks > 256u * 64u) blocks = 256u * 64u;
  if (n_hint >= 0) blocks = (unsigned)std::min<long long>(n_hint, 1 << 20);
  if (blocks == 0) return hipSuccess;
  auto go = [&](auto kern) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, stream, d_plan, m, w, multi, r, t_begin, t_end,
                       apply_th ? 1 : 0, th, s0_table, tile_win, survivors ? 1 : 0);
  };
  // groups = 64-cart groups walked speculatively per round: 1 where most windows are
  // rejected within a few carts (throughput), 4 where most pass (latency)
  auto pick = [&](auto trace_tag, auto multi_tag, auto st_tag) {
    constexpr bool TR = decltype(trace_tag)::value, MU = decltype(multi_tag)::value;
    constexpr bool STT = decltype(st_tag)::value && sizeof(Real) == 8;
    if constexpr (!TR && !MU && !STT) {
      // a stage's weight rows far larger than L2 (DevModelT::w_stream): the form that reads them with non-temporal loads
      if (m.w_stream) {
        if (groups >= 4) go(k_finish<DL, false, 4, false, false, true>);
        else if (groups == 3) go(k_finish<DL, false, 3, false, false, true>);
        else if (groups >= 2) go(k_finish<DL, false, 2, false, false, true>);
        else go(k_finish<DL, false, 1, false, false, true>);
        return;
      }
    }
    if (groups >= 4) go(k_finish<DL, TR, 4, MU, STT>);
    else if (groups == 3) go(k_finish<DL, TR, 3, MU, STT>);
    else if (groups >= 2) go(k_finish<DL, TR, 2, MU, STT>);
    else go(k_finish<DL, TR, 1, MU, STT>);
  };
  auto pick_m = [&](auto trace_tag, auto st_tag) {
    if (multi) pick(trace_tag, std::true_type{}, st_tag); else pick(trace_tag, std::false_type{}, st_tag);
  };
  if (st) { if (trace) pick_m(std::true_type{}, std::true_type{}); else pick_m(std::false_type{}, std::true_type{}); }
  else { if (trace) pick_m(std::true_type{}, std::false_type{}); else pick_m(std::false_type{}, std::false_type{}); }
  return hipGetLastError();
}
}  // namespace

template <>
hipError_t launch_finish<float>(bool trace, int t_begin, int t_end, bool apply_final_th, float final_th,
                                const DevPlan* d_plan, const DevModelT<float>& m, const WorkT<float>& w,
                                int groups, long long n_hint, const S0Node* s0_table, int tile_win, hipStream_t stream,
                                bool survivors) {
  return launch_finish_impl<DialectC>(trace, t_begin, t_end, apply_final_th, final_th, d_plan, m, w, groups, n_hint, s0_table, tile_win, survivors, stream);
}
template <>
hipError_t launch_finish<double>(bool trace, int t_begin, int t_end, bool apply_final_th, double final_th,
                                 const DevPlan* d_plan, const DevModelT<double>& m, const WorkT<double>& w,
                                 int groups, long long n_hint, const S0Node* s0_table, int tile_win, hipStream_t stream,
                                 bool survivors) {
  return launch_finish_impl<DialectCPP>(trace, t_begin, t_end, apply_final_th, final_th, d_plan, m, w, groups, n_hint, s0_table, tile_win, survivors, stream);
}


JDA_BC_READER(k_finish)

}  // namespace jda
