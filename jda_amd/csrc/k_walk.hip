// k_walk: the finishing path of the cascade regime for dialect C models whose split nodes read the original
// image only -- every window of the hand-off queue through the rest of stage 0, the later stages, the
// regressions and the final cut, in ONE launch of persistent workgroups (one per CU, 16 waves).
//
// A workgroup takes every gridDim-th window of the hand-off queue and runs the stages one after the other;
// for each stage it holds the stage's split nodes (20 bytes each: four fp32 offsets + landmark indices +
// threshold), leaf scores and cart thresholds in LDS.  Per window, one wave runs
//   lanes = carts     the stage's tree walks, 64 carts per group: node records and the window's shape from LDS,
//                     the two pixels of a node from the frame (c/jda.c:366-394)
//   replay            the score recurrence strictly in cart order with the per-cart reject (c/jda.c:395-399)
//   lanes = coords    the stage's regression: K weight rows added in cart order, streamed through a register
//                     ring (c/jda.c:404-411)
// Survivors of a stage go to the workgroup's own list in global memory (allocated with an LDS counter) and
// are dealt to the workgroup's waves again for the next stage; detections (final threshold c/jda.c:414) are
// collected in LDS and appended to the detection list with one global atomic per workgroup.
//
// Why this shape (measured, r02): same-address device atomics serialise at 13-40 ns each -- a queue counter
// bumped once per surviving window (4.7 k per stage) cost more than the walks themselves, and so did one
// launch per stage; the old wave-per-workgroup k_finish gathered every 32-byte node record through the
// texture addresser (2,700 vector-memory instructions per window).
#include "kernels_common.h"

namespace jda {

namespace {

constexpr int kWalkWaves = 16;
constexpr int kWalkBlock = kWalkWaves * 64;
constexpr int kWalkDets = 48;            // detections a workgroup collects in LDS before it falls back to direct appends

struct WalkLds {
  int off4, meta, leaf, cth, cnorm, det, misc, wave0, per_wave, sh, lbf, stash, det_stride, total;
  __host__ __device__ WalkLds(int K, int node_n, int leaf_n, int dim) {
    int o = 0;
    off4 = o; o += K * node_n * 16;
    meta = o; o += K * node_n * 4; o = (o + 15) & ~15;
    leaf = o; o += K * leaf_n * 4; o = (o + 15) & ~15;
    cth = o; o += K * 4; o = (o + 15) & ~15;
    cnorm = o; o += K; o = (o + 15) & ~15;
    det_stride = (2 + dim + 3) & ~3;                 // gid, score, shape[dim] (floats)
    det = o; o += kWalkDets * det_stride * 4;
    misc = o; o += 256;
    wave0 = o;
    int p = 0;
    sh = p; p += ((dim + 3) & ~3) * 4;
    lbf = p; p += ((K + 7) & ~7) * 2; p = (p + 15) & ~15;
    stash = p; p += 4 * 32;                           // survivors of a lockstep round: 4 x WalkWin
    per_wave = p;
    total = wave0 + kWalkWaves * per_wave;
  }
};

// One window's state between the pieces of the walk.
struct WalkWin { uint32_t gid, xy, wf; float score; unsigned hash; int kstart; };

}  // namespace

size_t walk_lds_bytes(int K, int node_n, int leaf_n, int dim) { return (size_t)WalkLds(K, node_n, leaf_n, dim).total; }

template <bool TRACE, int kG, int NWIN>
__global__ __launch_bounds__(kWalkBlock) void k_walk(const DevPlan* __restrict__ plan, DevModelT<float> m, WorkT<float> w,
                                                     int apply_th, float final_th) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int T = m.T, K = m.K, node_n = m.node_n, leaf_n = m.leaf_n, dim = m.dim, depth = m.D;
  const WalkLds L(K, node_n, leaf_n, dim);
  const float4* t_off = (const float4*)(lds + L.off4);
  const uint32_t* t_meta = (const uint32_t*)(lds + L.meta);
  const float* t_leaf = (const float*)(lds + L.leaf);
  const float* t_cth = (const float*)(lds + L.cth);
  const uint8_t* t_cnorm = lds + L.cnorm;
  float* det_buf = (float*)(lds + L.det);
  int* misc = (int*)(lds + L.misc);     // [0] work cursor, [1] survivors of this stage, [2] detections, [3] their base in the
                                        // detection list, [4..4+T) windows that completed stage t, [24],[25] carts evaluated
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* sh = (float*)(lds + L.wave0 + wv * L.per_wave + L.sh);
  uint16_t* lbf = (uint16_t*)(lds + L.wave0 + wv * L.per_wave + L.lbf);
  uint32_t* stash = (uint32_t*)(lds + L.wave0 + wv * L.per_wave + L.stash);

  const unsigned n0 = (unsigned)min(w.counters[kCntTail], (unsigned long long)w.cap);
  const unsigned G = gridDim.x, b = blockIdx.x;
  if (b >= n0) return;
  const unsigned my_n0 = (n0 - b + G - 1) / G;           // windows b, b+G, b+2G, ... of the hand-off queue are this workgroup's
  const unsigned region = b * (unsigned)w.list_stride;   // its slice of the two survivor lists
  if (tid < 32) misc[tid] = 0;

  const int W = plan->width;
  unsigned long long carts_acc = 0;

  for (unsigned chunk0 = 0; chunk0 < my_n0; chunk0 += (unsigned)w.list_stride) {
    // (a slice of at most list_stride hand-off windows at a time, so that the survivor lists cannot overflow)
    const unsigned chunk_n = min(my_n0 - chunk0, (unsigned)w.list_stride);
    unsigned n_in = chunk_n;
    for (int t = 0; t < T && n_in > 0; t++) {
      const WalkList lin = (t & 1) ? w.la : w.lb;        // stage t reads the list stage t-1 wrote (t >= 1)
      const WalkList lout = (t & 1) ? w.lb : w.la;
      __syncthreads();                                   // everyone is done with the previous stage's tables and lists
      // ---- the stage's tables -> LDS (LDS-DMA) ----
      const size_t s_nodes = (size_t)t * K * node_n;
      dma_to_lds<kWalkBlock>(lds + L.off4, m.off4 + 4 * s_nodes, K * node_n * 16, tid);
      dma_to_lds<kWalkBlock>(lds + L.meta, m.meta + s_nodes, K * node_n * 4, tid);
      dma_to_lds<kWalkBlock>(lds + L.leaf, m.leaf + (size_t)t * K * leaf_n, K * leaf_n * 4, tid);
      dma_to_lds<kWalkBlock>(lds + L.cth, m.cth + (size_t)t * K, K * 4, tid);
      dma_to_lds<kWalkBlock>(lds + L.cnorm, m.cnorm + (size_t)t * K, K, tid);
      if (tid == 0) { misc[0] = 0; misc[1] = 0; }
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();

      const float* cmean = m.cmean + (size_t)t * K;
      const float* cstd = m.cstd + (size_t)t * K;
      const float* wt = m.w + (size_t)t * K * leaf_n * m.wpitch;

      // Feature test of one node for the window (wbase, win) whose shape is sh[]: c/jda.c:370-393 (fp32 add, fp32
      // multiply by the window side, truncate, clamp; two bytes; compare).  -> next node index.
      auto step_node = [&](int node, int k, const uint8_t* wbase, int win) {
        const int ni = k * node_n + node;
        const float4 o = t_off[ni];
        const uint32_t mt = t_meta[ni];
        const int l1 = (int)(mt & 0xffu), l2 = (int)((mt >> 8) & 0xffu);
        const int x1 = clamp_win(DialectC::coord(sh[l1], o.x, win), win), y1 = clamp_win(DialectC::coord(sh[l1 + 1], o.y, win), win);
        const int x2 = clamp_win(DialectC::coord(sh[l2], o.z, win), win), y2 = clamp_win(DialectC::coord(sh[l2 + 1], o.w, win), win);
        const int a = wbase[__umul24((unsigned)y1, (unsigned)W) + (unsigned)x1];
        const int bb = wbase[__umul24((unsigned)y2, (unsigned)W) + (unsigned)x2];
        return 2 * node + ((a - bb <= (int)(mt >> 16) - 256) ? 1 : 2);
      };
      auto window_base = [&](const WalkWin& wn, int* win) {
        const int x0 = (int)(wn.xy & 0xffffu), y0 = (int)(wn.xy >> 16);
        *win = (int)(wn.wf & 0xffffu);
        return w.frames + (size_t)(wn.wf >> 16) * w.frame_stride + (size_t)y0 * W + x0;
      };
      // The window's walk is over (rejected at carts_n, or through the last stage): account, trace, final cut.
      auto retire = [&](const WalkWin& wn, bool alive, int carts_n) {
        carts_acc += (unsigned long long)carts_n;
        if (TRACE) {
          if (lane == 0) { w.tr_carts[wn.gid] = carts_n; w.tr_score[wn.gid] = wn.score; w.tr_hash[wn.gid] = wn.hash; }
          for (int d = lane; d < dim; d += 64) w.tr_shape[(size_t)wn.gid * dim + d] = sh[d];
        }
        if (alive && !(apply_th && wn.score < final_th)) {            // c/jda.c:414
          int p = 0;
          if (lane == 0) p = atomicAdd(&misc[2], 1);
          p = __shfl(p, 0);
          if (p < kWalkDets) {
            float* o = det_buf + p * L.det_stride;
            if (lane == 0) { o[0] = __uint_as_float(wn.gid); o[1] = wn.score; }
            for (int d = lane; d < dim; d += 64) o[2 + d] = sh[d];
          } else {
            // more detections than the LDS buffer holds (not the cascade regime): straight to the list
            unsigned q = 0;
            if (lane == 0) q = (unsigned)atomicAdd(&w.counters[kCntOut], 1ull);
            q = (unsigned)__shfl((int)q, 0);
            if (q < w.cap) {
              if (lane == 0) { w.out_gid[q] = wn.gid; w.out_score[q] = wn.score; }
              for (int d = lane; d < dim; d += 64) w.out_shape[(size_t)q * dim + d] = sh[d];
            }
          }
        }
      };

      // Everything after the entry of ONE window whose stage-start shape is in sh[]: tree walks from cart k_from
      // (a multiple of 64; the scores of earlier carts are already applied), kG groups of 64 carts per round with
      // the score recurrence replayed in cart order after each; then the leaves of the carts below k_from, the
      // stage regression, and the hand-over to the next stage (or the final cut).
      auto finish = [&](WalkWin wn, int k_from) {
        int win;
        const uint8_t* wbase = window_base(wn, &win);
        bool alive = true;
        int carts_n = 0;
        for (int k0 = k_from; k0 < K && alive; k0 += 64 * kG) {
          int node[kG], kk[kG];
#pragma unroll
          for (int g = 0; g < kG; g++) { node[g] = 0; kk[g] = min(k0 + g * 64 + lane, K - 1); }   // clamped lanes repeat cart K-1
          for (int d = 0; d < depth - 1; d++) {
            float4 o[kG];
            uint32_t mt[kG];
#pragma unroll
            for (int g = 0; g < kG; g++) { const int ni = kk[g] * node_n + node[g]; o[g] = t_off[ni]; mt[g] = t_meta[ni]; }
            float s1x[kG], s1y[kG], s2x[kG], s2y[kG];
#pragma unroll
            for (int g = 0; g < kG; g++) {
              const int l1 = (int)(mt[g] & 0xffu), l2 = (int)((mt[g] >> 8) & 0xffu);
              s1x[g] = sh[l1]; s1y[g] = sh[l1 + 1]; s2x[g] = sh[l2]; s2y[g] = sh[l2 + 1];
            }
            int pa[kG], pb[kG];
#pragma unroll
            for (int g = 0; g < kG; g++) {
              const int x1 = clamp_win(DialectC::coord(s1x[g], o[g].x, win), win), y1 = clamp_win(DialectC::coord(s1y[g], o[g].y, win), win);
              const int x2 = clamp_win(DialectC::coord(s2x[g], o[g].z, win), win), y2 = clamp_win(DialectC::coord(s2y[g], o[g].w, win), win);
              pa[g] = wbase[__umul24((unsigned)y1, (unsigned)W) + (unsigned)x1];
              pb[g] = wbase[__umul24((unsigned)y2, (unsigned)W) + (unsigned)x2];
            }
#pragma unroll
            for (int g = 0; g < kG; g++) node[g] = 2 * node[g] + ((pa[g] - pb[g] <= (int)(mt[g] >> 16) - 256) ? 1 : 2);   // c/jda.c:391-393
          }
          int lf[kG], nrm[kG];
          float ls[kG], thk[kG], mk[kG], sk[kG];
#pragma unroll
          for (int g = 0; g < kG; g++) {
            const int k = k0 + g * 64 + lane;
            lf[g] = node[g] - node_n;
            ls[g] = 0; thk[g] = 0; mk[g] = 0; sk[g] = 1; nrm[g] = 0;
            if (k < K) {
              lbf[k] = (uint16_t)(k * leaf_n + lf[g]);
              ls[g] = t_leaf[k * leaf_n + lf[g]];
              thk[g] = t_cth[k];
              nrm[g] = t_cnorm[k];
              if (nrm[g]) { mk[g] = cmean[k]; sk[g] = cstd[k]; }
            }
          }
#pragma unroll
          for (int g = 0; g < kG; g++) {
            const int kg = k0 + g * 64;
            if (kg >= K || !alive) break;
            const unsigned long long normmask = __ballot(nrm[g] != 0);
            const int jr = replay_scores<float, TRACE>(wn.score, wn.hash, ls[g], thk[g], mk[g], sk[g], normmask, lf[g],
                                                       max(0, wn.kstart - kg), min(64, K - kg));
            if (jr >= 0) { alive = false; carts_n = t * K + kg + jr + 1; }
          }
        }
        if (!alive) { retire(wn, false, carts_n); return; }

        // leaves of the carts whose scores were applied earlier (k_scan, or the lockstep round of stage 0):
        // needed only now that the stage is passed
        const int k_lazy = min(k_from, K);
        for (int k0 = 0; k0 < k_lazy; k0 += 64) {
          const int k = min(k0 + lane, k_lazy - 1);
          int node = 0;
          for (int d = 0; d < depth - 1; d++) node = step_node(node, k, wbase, win);
          if (k0 + lane < k_lazy) lbf[k0 + lane] = (uint16_t)((k0 + lane) * leaf_n + node - node_n);
        }
        wave_lds_sync();                          // lbf is complete
        // ---- stage regression: K weight rows added strictly in cart order (c/jda.c:404-411).  The rows stream
        //      through a register ring: every add frees a register for the load kRing rows ahead, so kRing row
        //      loads are in flight all the time ----
        constexpr int kRing = 16;
        for (int d0 = 0; d0 < dim; d0 += 64) {
          const int d = d0 + lane;
          if (d < dim) {
            float acc = sh[d];
            const float* col = wt + d;
            const unsigned wp = (unsigned)m.wpitch;
            float r[kRing];
#pragma unroll
            for (int u = 0; u < kRing; u++) r[u] = col[(unsigned)lbf[min(u, K - 1)] * wp];
            int k = 0;
            for (; k + kRing <= K; k += kRing) {
#pragma unroll
              for (int u = 0; u < kRing; u++) {
                acc = acc + r[u];
                r[u] = col[(unsigned)lbf[min(k + kRing + u, K - 1)] * wp];     // past K: a dummy load, never added
              }
            }
#pragma unroll
            for (int u = 0; u < kRing; u++) if (u < K - k) acc = acc + r[u];
            sh[d] = acc;                            // (this lane's reads of sh[d] are done)
          }
        }
        wave_lds_sync();
        if (lane == 0) atomicAdd(&misc[4 + t], 1);
        if (t + 1 == T) { retire(wn, true, T * K); return; }
        // alive with stages left: the workgroup's list for the next stage
        int p = 0;
        if (lane == 0) p = atomicAdd(&misc[1], 1);
        p = __shfl(p, 0);
        const unsigned o = region + (unsigned)p;
        if (lane == 0) { lout.gid[o] = wn.gid; lout.score[o] = wn.score; lout.xy[o] = wn.xy; lout.wf[o] = wn.wf; if (TRACE) lout.hash[o] = wn.hash; }
        for (int d = lane; d < dim; d += 64) lout.shape[(size_t)o * dim + d] = sh[d];
      };

      if (t == 0) {
        // ---- hand-off queue: every window holds the mean shape and most of them die within their next 64 carts,
        //      so a wave takes NWIN windows at once and walks those 64 carts of all of them in lockstep (one
        //      memory round trip per tree level for NWIN windows); the few that survive continue one by one ----
        const unsigned n_batches = (chunk_n + NWIN - 1) / NWIN;
        bool sh_mean = false;
        for (;;) {
          int q = 0;
          if (lane == 0) q = atomicAdd(&misc[0], 1);
          q = __shfl(q, 0);
          if ((unsigned)q >= n_batches) break;
          if (!sh_mean) {
            for (int d = lane; d < dim; d += 64) sh[d] = m.mean_shape[d];
            wave_lds_sync();
            sh_mean = true;
          }
          WalkWin ws[NWIN];
          const uint8_t* wb[NWIN];
          int wn_[NWIN], kf[NWIN], node[NWIN], kk[NWIN];
          bool valid[NWIN];
#pragma unroll
          for (int g = 0; g < NWIN; g++) {
            const unsigned j = (unsigned)q * NWIN + g;            // index inside this workgroup's slice of the queue
            valid[g] = j < chunk_n;
            const unsigned i = b + G * (chunk0 + (valid[g] ? j : 0u));
            ws[g].gid = w.q_gid[i]; ws[g].score = w.q_score[i]; ws[g].xy = w.q_xy[i]; ws[g].wf = w.q_wf[i];
            ws[g].kstart = min((int)w.q_kstart[i], K);            // first cart whose score is still to be applied
            ws[g].hash = kFnvSeed;
            if (TRACE) ws[g].hash = w.q_hash[i];
            wb[g] = window_base(ws[g], &wn_[g]);
            kf[g] = ws[g].kstart & ~63;
            node[g] = 0; kk[g] = min(kf[g] + lane, K - 1);
          }
          for (int d = 0; d < depth - 1; d++) {
            float4 o[NWIN];
            uint32_t mt[NWIN];
#pragma unroll
            for (int g = 0; g < NWIN; g++) { const int ni = kk[g] * node_n + node[g]; o[g] = t_off[ni]; mt[g] = t_meta[ni]; }
            int pa[NWIN], pb[NWIN];
#pragma unroll
            for (int g = 0; g < NWIN; g++) {
              const int l1 = (int)(mt[g] & 0xffu), l2 = (int)((mt[g] >> 8) & 0xffu);
              const int win = wn_[g];
              const int x1 = clamp_win(DialectC::coord(sh[l1], o[g].x, win), win), y1 = clamp_win(DialectC::coord(sh[l1 + 1], o[g].y, win), win);
              const int x2 = clamp_win(DialectC::coord(sh[l2], o[g].z, win), win), y2 = clamp_win(DialectC::coord(sh[l2 + 1], o[g].w, win), win);
              pa[g] = wb[g][__umul24((unsigned)y1, (unsigned)W) + (unsigned)x1];
              pb[g] = wb[g][__umul24((unsigned)y2, (unsigned)W) + (unsigned)x2];
            }
#pragma unroll
            for (int g = 0; g < NWIN; g++) node[g] = 2 * node[g] + ((pa[g] - pb[g] <= (int)(mt[g] >> 16) - 256) ? 1 : 2);
          }
          float ls[NWIN], thk[NWIN], mk[NWIN], sk[NWIN];
          int lf[NWIN], nrm[NWIN];
#pragma unroll
          for (int g = 0; g < NWIN; g++) {
            const int k = kf[g] + lane;
            lf[g] = node[g] - node_n;
            ls[g] = 0; thk[g] = 0; mk[g] = 0; sk[g] = 1; nrm[g] = 0;
            if (k < K) {
              ls[g] = t_leaf[k * leaf_n + lf[g]];
              thk[g] = t_cth[k];
              nrm[g] = t_cnorm[k];
              if (nrm[g]) { mk[g] = cmean[k]; sk[g] = cstd[k]; }
            }
          }
          int n_surv = 0;
#pragma unroll
          for (int g = 0; g < NWIN; g++) {
            if (!valid[g]) continue;
            int jr = -1;
            if (kf[g] < K) {
              const unsigned long long normmask = __ballot(nrm[g] != 0);
              jr = replay_scores<float, TRACE>(ws[g].score, ws[g].hash, ls[g], thk[g], mk[g], sk[g], normmask, lf[g],
                                               ws[g].kstart - kf[g], min(64, K - kf[g]));
            }
            if (jr >= 0) {
              retire(ws[g], false, kf[g] + jr + 1);          // (sh[] holds the mean shape here)
            } else {
              // survivors go through LDS to the one-window path below, so that the state of the other windows of
              // the round does not stay in registers across it
              if (lane == 0) {
                uint32_t* o = stash + n_surv * 8;
                o[0] = ws[g].gid; o[1] = ws[g].xy; o[2] = ws[g].wf; o[3] = __float_as_uint(ws[g].score); o[4] = ws[g].hash;
                o[5] = (uint32_t)ws[g].kstart;
              }
              n_surv++;
            }
          }
          wave_lds_sync();
          for (int s2 = 0; s2 < n_surv; s2++) {
            if (!sh_mean) { for (int d = lane; d < dim; d += 64) sh[d] = m.mean_shape[d]; wave_lds_sync(); }
            const uint32_t* o = stash + s2 * 8;
            WalkWin wn;
            wn.gid = o[0]; wn.xy = o[1]; wn.wf = o[2]; wn.score = __uint_as_float(o[3]); wn.hash = o[4]; wn.kstart = (int)o[5];
            finish(wn, min((wn.kstart & ~63) + 64, (K + 63) & ~63));
            sh_mean = false;                      // the regression went through sh[]
          }
        }
      } else {
        for (;;) {
          int q = 0;
          if (lane == 0) q = atomicAdd(&misc[0], 1);
          q = __shfl(q, 0);
          if ((unsigned)q >= n_in) break;
          const unsigned i = region + (unsigned)q;
          WalkWin wn;
          wn.gid = lin.gid[i]; wn.score = lin.score[i]; wn.xy = lin.xy[i]; wn.wf = lin.wf[i]; wn.kstart = 0;
          wn.hash = kFnvSeed;
          if (TRACE) wn.hash = lin.hash[i];
          const float* src = lin.shape + (size_t)i * dim;
          for (int d = lane; d < dim; d += 64) sh[d] = src[d];
          wave_lds_sync();
          finish(wn, 0);
        }
      }
      __threadfence_block();                             // this stage's list entries are visible to the workgroup
      __syncthreads();
      n_in = (unsigned)misc[1];
    }
    __syncthreads();
  }

  // ---- detections collected in LDS -> the detection list, one global atomic per workgroup ----
  __syncthreads();
  const int n_det = min(misc[2], kWalkDets);
  if (tid == 0 && n_det > 0) misc[3] = (int)atomicAdd(&w.counters[kCntOut], (unsigned long long)n_det);
  __syncthreads();
  if (n_det > 0) {
    const unsigned base = (unsigned)misc[3];
    for (int p = wv; p < n_det; p += kWalkWaves) {
      const float* o = det_buf + p * L.det_stride;
      const unsigned q = base + (unsigned)p;
      if (q < w.cap) {
        if (lane == 0) { w.out_gid[q] = __float_as_uint(o[0]); w.out_score[q] = o[1]; }
        for (int d = lane; d < dim; d += 64) w.out_shape[(size_t)q * dim + d] = o[2 + d];
      }
    }
  }
  // ---- counters: one atomic set per workgroup ----
  if (lane == 0 && carts_acc) atomicAdd((unsigned long long*)&misc[24], carts_acc);
  __syncthreads();
  if (tid < T && misc[4 + tid]) atomicAdd(shard_counter(w.counters, kCntStage0 + tid), (unsigned long long)misc[4 + tid]);
  if (tid == 0) {
    const unsigned long long ca = *(const unsigned long long*)&misc[24];
    if (ca) atomicAdd(shard_counter(w.counters, kCntCarts), ca);
  }
}

hipError_t launch_walk(bool trace, int groups, bool apply_final_th, float final_th, const DevPlan* d_plan,
                       const DevModelT<float>& m, const WorkT<float>& w, int n_blocks, int nwin, hipStream_t stream) {
  const WalkLds L(m.K, m.node_n, m.leaf_n, m.dim);
  if (L.total > 160 * 1024 || m.off4 == nullptr || m.T > 16 || w.list_stride <= 0 || w.la.gid == nullptr) return hipErrorInvalidValue;
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    hipLaunchKernelGGL(kern, dim3((unsigned)n_blocks), dim3(kWalkBlock), L.total, stream, d_plan, m, w,
                       apply_final_th ? 1 : 0, final_th);
  };
  // groups = 64-cart groups walked per round by a window that has passed the lockstep round of stage 0;
  // nwin = windows of the hand-off queue a wave walks in lockstep (2 or 4)
  const bool g3 = groups >= 2;
  if (trace) {
    if (nwin >= 4) { if (g3) go(k_walk<true, 3, 4>); else go(k_walk<true, 1, 4>); }
    else { if (g3) go(k_walk<true, 3, 2>); else go(k_walk<true, 1, 2>); }
  } else {
    if (nwin >= 4) { if (g3) go(k_walk<false, 3, 4>); else go(k_walk<false, 1, 4>); }
    else { if (g3) go(k_walk<false, 3, 2>); else go(k_walk<false, 1, 2>); }
  }
  return hipGetLastError();
}

}  // namespace jda
