// HIP kernels of the JDA detect path for gfx950 (MI355X, wave64): pyramid images, stage-0 offset tables,
// queue fill, trace defaults.
//   k_resize        bilinear pyramid image        reference c/jda.c:203-230
//   k_resize_cv     cv::resize(INTER_LINEAR) restated (dialect CPP pyramids, cascador.cpp:302,330-331)
//   k_prep_stage0   stage-0 feature offsets per level (hoisted c/jda.c:370-389)
//   k_enqueue       windows of levels k_scan does not cover -> hand-off queue at cart 0
//   k_trace_fill    per-window trace defaults (parity instrumentation)
#include "kernels_common.h"

namespace jda {

// =============================================================================
// pyramid resize
// =============================================================================

__global__ void k_resize(const uint8_t* __restrict__ src, size_t src_stride, int sw, int sh,
                         uint8_t* __restrict__ dst, size_t dst_stride, int dw, int dh,
                         float rx, float ry) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  const int f = blockIdx.z;
  if (j >= dw || i >= dh) return;
  const uint8_t* s = src + (size_t)f * src_stride;
  // c/jda.c:215-226, operation for operation
  const float fx = rx * (float)j;
  const float fy = ry * (float)i;
  const int x = (int)fx;
  const int y = (int)fy;
  const float xd = fx - (float)x;
  const float yd = fy - (float)y;
  const int idx = y * sw + x;
  const float a = (float)(int)s[idx], b = (float)(int)s[idx + 1];
  const float c = (float)(int)s[idx + sw], d = (float)(int)s[idx + sw + 1];
  const float one_x = 1.f - xd, one_y = 1.f - yd;
  float v = a * one_x * one_y;
  v = v + b * xd * one_y;
  v = v + c * one_x * yd;
  v = v + d * xd * yd;
  dst[(size_t)f * dst_stride + (size_t)i * dw + j] = (uint8_t)(int)v;
}

hipError_t launch_resize(const uint8_t* src, size_t src_stride, int n, int sw, int sh,
                         uint8_t* dst, size_t dst_stride, int dw, int dh, float rx, float ry,
                         hipStream_t stream) {
  if (dw <= 0 || dh <= 0 || n <= 0) return hipSuccess;
  dim3 block(256), grid((dw + 255) / 256, dh, n);
  hipLaunchKernelGGL(k_resize, grid, block, 0, stream, src, src_stride, sw, sh, dst, dst_stride, dw, dh, rx, ry);
  return hipGetLastError();
}

// cv::resize(INTER_LINEAR) for 8-bit single-channel images as dialect CPP uses it for the
// half/quarter images (cascador.cpp:329-331) and the method-0 pyramid (cascador.cpp:300-303):
// 11-bit fixed-point bilinear of OpenCV's 2.4/3.x imgwarp.cpp, with its routing of an exact
// 2x2 down-scale to the box average.  PARITY UNPINNED (no OpenCV here to compare with);
// bit-exact against the oracle's restatement of the same algorithm.
// One output pixel (dx, dy) of the resize of an sw x sh image whose rows are `pitch` bytes apart.
__device__ __forceinline__ uint8_t resize_cv_pixel(const uint8_t* __restrict__ s, int pitch, int sw, int sh, int dx, int dy,
                                                   double scale_x, double scale_y, int area_fast) {
  if (area_fast) {
    const uint8_t* p = s + (size_t)(2 * dy) * pitch + 2 * dx;
    return (uint8_t)((p[0] + p[1] + p[pitch] + p[pitch + 1] + 2) >> 2);
  }
  float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  const bool edge = sx + 1 >= sw;            // dx >= xmax in OpenCV's loop
  if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
  float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
  const int sy = (int)floorf(fy);
  fy -= (float)sy;
  auto sat_short = [](float v) { int i = __float2int_rn(v); return i < -32768 ? -32768 : (i > 32767 ? 32767 : i); };
  const int a0 = sat_short((1.f - fx) * 2048.f), a1 = sat_short(fx * 2048.f);
  const int b0 = sat_short((1.f - fy) * 2048.f), b1 = sat_short(fy * 2048.f);
  const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
  const uint8_t* S0 = s + (size_t)y0 * pitch;
  const uint8_t* S1 = s + (size_t)y1 * pitch;
  int r0, r1;
  if (!edge) { r0 = S0[sx] * a0 + S0[sx + 1] * a1; r1 = S1[sx] * a0 + S1[sx + 1] * a1; }
  else { r0 = S0[sx] * 2048; r1 = S1[sx] * 2048; }
  return (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

__global__ void k_resize_cv(const uint8_t* __restrict__ src, size_t src_stride, int sw, int sh,
                            uint8_t* __restrict__ dst, size_t dst_stride, int dw, int dh,
                            double scale_x, double scale_y, int area_fast) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  const int dy = blockIdx.y;
  const int f = blockIdx.z;
  if (dx >= dw || dy >= dh) return;
  dst[(size_t)f * dst_stride + (size_t)dy * dw + dx] =
      resize_cv_pixel(src + (size_t)f * src_stride, sw, sw, sh, dx, dy, scale_x, scale_y, area_fast);
}

// The same resize for the ROI of every window of a level (method 0 on a multi-scale model, cascador.cpp:243-245: the ROI
// is an image of its own to cv::resize): a thread per output pixel of a window's ds x ds patch.
__global__ void k_resize_cv_patches(const uint8_t* __restrict__ src, size_t src_stride, int lw, int nx, int step, int win,
                                    uint8_t* __restrict__ dst, size_t dst_stride, int ds,
                                    double scale, int area_fast) {
  const int pb = (ds * ds + (int)blockDim.x - 1) / (int)blockDim.x;     // blocks per patch
  const int wi = (int)(blockIdx.x / (unsigned)pb);
  const int e = (int)(blockIdx.x - (unsigned)wi * (unsigned)pb) * (int)blockDim.x + (int)threadIdx.x;
  const int f = blockIdx.y;
  if (e >= ds * ds) return;
  const int dy = e / ds, dx = e - dy * ds;
  const int wy = wi / nx, wx = wi - wy * nx;
  const uint8_t* roi = src + (size_t)f * src_stride + (size_t)(wy * step) * lw + wx * step;
  dst[(size_t)f * dst_stride + (size_t)wi * ds * ds + e] = resize_cv_pixel(roi, lw, win, win, dx, dy, scale, scale, area_fast);
}

hipError_t launch_resize_cv(const uint8_t* src, size_t src_stride, int n, int sw, int sh,
                            uint8_t* dst, size_t dst_stride, int dw, int dh, hipStream_t stream) {
  if (dw <= 0 || dh <= 0 || n <= 0) return hipSuccess;
  const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
  const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
  const int area = (fabs(scale_x - 2.) < 2.220446049250313e-16 && fabs(scale_y - 2.) < 2.220446049250313e-16) ? 1 : 0;
  dim3 block(256), grid((dw + 255) / 256, dh, n);
  hipLaunchKernelGGL(k_resize_cv, grid, block, 0, stream, src, src_stride, sw, sh, dst, dst_stride, dw, dh,
                     scale_x, scale_y, area);
  return hipGetLastError();
}

hipError_t launch_resize_cv_patches(const uint8_t* src, size_t src_stride, int n, int lw, int nx, int ny, int step, int win,
                                    uint8_t* dst, size_t dst_stride, int ds, hipStream_t stream) {
  if (ds <= 0 || n <= 0 || nx <= 0 || ny <= 0) return hipSuccess;
  if (n > 65535 || (long long)nx * ny * ((ds * ds + 255) / 256) > 0x7fffffffLL) return hipErrorInvalidValue;
  const double inv = (double)ds / win;
  const double scale = 1. / inv;
  const int area = fabs(scale - 2.) < 2.220446049250313e-16 ? 1 : 0;
  dim3 block(256), grid((unsigned)((long long)nx * ny * ((ds * ds + 255) / 256)), (unsigned)n);
  hipLaunchKernelGGL(k_resize_cv_patches, grid, block, 0, stream, src, src_stride, lw, nx, step, win, dst, dst_stride, ds, scale, area);
  return hipGetLastError();
}

// =============================================================================
// ragged batches: tight images -> one common row pitch
// =============================================================================

// Every image of a ragged batch is staged with the same row pitch (a multiple of 16 bytes): k_scan's LDS-DMA tile
// loads need 16-byte aligned rows and its global-pixel node offsets are resolved for ONE pitch.  The images arrive
// tight (row stride = width, the `data` of reference c/jda.c:445-448); this copies them into place, one
// workgroup per (image, 4 rows), a dword of output per thread and step.
__global__ __launch_bounds__(256) void k_repack(const uint8_t* __restrict__ raw, uint8_t* __restrict__ dst,
                                                const RagImg* __restrict__ imgs, int pitch) {
  const RagImg im = imgs[blockIdx.y];
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (y >= im.h) return;
  const uint8_t* s = raw + im.src_off + (size_t)y * im.w;
  uint32_t* d = (uint32_t*)(dst + im.dst_off + (size_t)y * pitch);
  // 8 output bytes per thread and step: two dword loads at the row's own (arbitrary) alignment -- global memory takes
  // unaligned dwords on this target -- and one aligned 8-byte store; the row's last bytes one by one
  const int nq = im.w >> 3;
  uint2* d2 = (uint2*)d;
  for (int j = threadIdx.x & 63; j < nq; j += 64) {
    uint2 v;
    __builtin_memcpy(&v, s + 8 * j, 8);
    d2[j] = v;
  }
  const int x0 = nq << 3;
  if ((threadIdx.x & 63) == 0 && x0 < im.w) {
    uint32_t v0 = 0, v1 = 0;
    for (int k = 0; k < 4; k++) if (x0 + k < im.w) v0 |= (uint32_t)s[x0 + k] << (8 * k);
    for (int k = 0; k < 4; k++) if (x0 + 4 + k < im.w) v1 |= (uint32_t)s[x0 + 4 + k] << (8 * k);
    d[x0 >> 2] = v0;
    if (x0 + 4 < im.w) d[(x0 >> 2) + 1] = v1;
  }
}

hipError_t launch_repack(const uint8_t* raw, uint8_t* dst, const RagImg* imgs, int n, int max_h, int pitch, hipStream_t stream) {
  if (n <= 0 || max_h <= 0) return hipSuccess;
  // (gridDim.y is limited to 65535: a pass holds at most that many images, the queues pack the index in 16 bits)
  hipLaunchKernelGGL(k_repack, dim3((unsigned)((max_h + 3) / 4), (unsigned)n), dim3(256), 0, stream, raw, dst, imgs, pitch);
  return hipGetLastError();
}

// =============================================================================
// stage-0 offset table
// =============================================================================

template <typename DL>
__global__ void k_prep_stage0(const DevPlan* __restrict__ plan, const typename DL::Node* __restrict__ nodes,
                              const typename DL::Real* __restrict__ mean_shape, int K, int node_n,
                              S0Node* __restrict__ table, S0Node* __restrict__ table_lm) {
  const int l = blockIdx.y;
  const DevLevel lv = plan->lv[l];
  if (!lv.tiled) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * node_n) return;
  const typename DL::Node nd = nodes[i];  // stage 0 occupies the first K*node_n nodes
  const int win = lv.win;
  int x1 = clamp_win(DL::coord(mean_shape[nd.lm1x2], nd.o1x, win), win);
  int y1 = clamp_win(DL::coord(mean_shape[nd.lm1x2 + 1], nd.o1y, win), win);
  int x2 = clamp_win(DL::coord(mean_shape[nd.lm2x2], nd.o2x, win), win);
  int y2 = clamp_win(DL::coord(mean_shape[nd.lm2x2 + 1], nd.o2y, win), win);
  // feature is a difference of two bytes: thresholds beyond [-256,255] behave like the ends
  const int th = nd.th < -256 ? -256 : (nd.th > 255 ? 255 : nd.th);
  // byte offset of the window's pixel (x, y) in k_scan's pixel store: the LDS tile (rows of `pitch` bytes) or the
  // frame itself (tiled == 2: pitch = frame width)
  auto off = [&](int x, int y) -> uint32_t { return (uint32_t)(y * lv.pitch + x); };
  auto wide = [&](uint32_t a, uint32_t b) {
    const unsigned long long v = (unsigned long long)a | ((unsigned long long)b << kS0GlobalOffBits) |
                                 ((unsigned long long)(uint32_t)(th + 256) << (2 * kS0GlobalOffBits));
    S0Node r; r.lo = (uint32_t)v; r.hi = (uint32_t)(v >> 32);
    return r;
  };
  S0Node o;
  if (lv.tiled == 1) {
    o.lo = off(x1, y1) | (off(x2, y2) << 16);
    o.hi = (uint32_t)th;
  } else {
    o = wide(off(x1, y1), off(x2, y2));
  }
  table[lv.s0_table + i] = o;                       // cart-major: k_scan stages chunks of carts into LDS
  // level-major copy for k_finish (kernels.h: lm_index)
  const unsigned k = (unsigned)i / (unsigned)node_n, n = (unsigned)i - k * (unsigned)node_n;
  unsigned d = 0;
  while (n >= (2u << d) - 1u) d++;
  // ... with the two pixels as (x, y) inside the window, 11 bits each (k_finish reads the frame or its own copy of
  // the window, two different pitches; the host only hands it this table when every tiled window is below 2048 px):
  //   x1 : 11 | y1 : 11 | x2 : 11 | y2 : 11 | th + 256 : 10
  const unsigned long long v = (unsigned long long)(uint32_t)(x1 | (y1 << 11)) |
                               ((unsigned long long)(uint32_t)(x2 | (y2 << 11)) << 22) |
                               ((unsigned long long)(uint32_t)(th + 256) << 44);
  S0Node q; q.lo = (uint32_t)v; q.hi = (uint32_t)(v >> 32);
  table_lm[lv.s0_table + lm_index((unsigned)K, k, d, n)] = q;
}

hipError_t launch_prep_stage0(int dialect, const DevPlan* d_plan, const DevPlan& h_plan,
                              const void* nodes, const void* mean_shape, int K, int node_n,
                              S0Node* table, S0Node* table_lm, hipStream_t stream) {
  dim3 block(256), grid((K * node_n + 255) / 256, h_plan.n_levels);
  if (dialect == 0)
    hipLaunchKernelGGL(k_prep_stage0<DialectC>, grid, block, 0, stream, d_plan, (const NodeF*)nodes,
                       (const float*)mean_shape, K, node_n, table, table_lm);
  else
    hipLaunchKernelGGL(k_prep_stage0<DialectCPP>, grid, block, 0, stream, d_plan, (const NodeD*)nodes,
                       (const double*)mean_shape, K, node_n, table, table_lm);
  return hipGetLastError();
}

// =============================================================================
// windows k_scan does not cover -> head of the hand-off queue (k_start = 0)
// =============================================================================

template <typename Real>
__global__ void k_enqueue(const DevPlan* __restrict__ plan, WorkT<Real> w, int per_frame, int all_levels) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)per_frame * w.n_frames;
  if (idx == 0) w.counters[kCntTail] = (unsigned long long)total;   // k_scan appends behind these
  if (idx >= total) return;
  const int frame = (int)(idx / per_frame);
  int r = (int)(idx - (long long)frame * per_frame);
  int wid = -1, lvl = 0, rel = 0;
  for (int i = 0; i < plan->n_levels; i++) {
    const DevLevel* c = &plan->lv[i];
    if (!all_levels && c->tiled) continue;
    const int cnt = c->nx * c->ny;
    if (wid < 0 && r < cnt) { wid = c->base + r; lvl = i; rel = r; }
    r -= cnt;
  }
  {
    const DevLevel* c = &plan->lv[lvl];
    const int iy = rel / c->nx, ix = rel - iy * c->nx;
    w.q_xy[idx] = (uint32_t)(ix * c->step) | ((uint32_t)(iy * c->step) << 16);
    w.q_wf[idx] = (uint32_t)c->win | ((uint32_t)frame << 16);
  }
  w.q_gid[idx] = (uint32_t)(frame * plan->windows + wid);
  w.q_score[idx] = (Real)0;
  w.q_kstart[idx] = 0u;
  if (w.q_hash) w.q_hash[idx] = kFnvSeed;
}

template <typename Real>
hipError_t launch_enqueue(const DevPlan* d_plan, const DevPlan& h_plan, bool all_levels,
                          const WorkT<Real>& w, hipStream_t stream) {
  long long per_frame = 0;
  for (int i = 0; i < h_plan.n_levels; i++)
    if (all_levels || !h_plan.lv[i].tiled) per_frame += (long long)h_plan.lv[i].nx * h_plan.lv[i].ny;
  const long long total = per_frame * w.n_frames;
  if (total == 0) return hipSuccess;
  dim3 block(256), grid((unsigned)((total + 255) / 256));
  hipLaunchKernelGGL(k_enqueue<Real>, grid, block, 0, stream, d_plan, w, (int)per_frame, all_levels ? 1 : 0);
  return hipGetLastError();
}
template hipError_t launch_enqueue<float>(const DevPlan*, const DevPlan&, bool, const WorkT<float>&, hipStream_t);
template hipError_t launch_enqueue<double>(const DevPlan*, const DevPlan&, bool, const WorkT<double>&, hipStream_t);

// =============================================================================
// results -> pinned host memory by a kernel
// =============================================================================
// The counters and the detections of a pass are a few hundred KB.  As hipMemcpyAsync they go to the copy engine, where
// they queue behind the 78-MB frame uploads of the NEXT tickets of a host-frame stream (measured: a ticket's Wait then
// sits out another batch's upload, 2.46 ms per batch on a link that moves a batch in 1.41 ms).  Written by a kernel
// into mapped pinned memory they travel on the lane's own compute queue.
struct CopySeg { const void* src; void* dst; unsigned long long bytes; };
struct CopySegs { CopySeg s[4]; };

__global__ __launch_bounds__(256) void k_copy_out(CopySegs segs) {
  const CopySeg sg = segs.s[blockIdx.y];
  const unsigned long long n16 = sg.bytes >> 4;
  const uint4* s4 = (const uint4*)sg.src;
  uint4* d4 = (uint4*)sg.dst;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * blockDim.x) d4[i] = s4[i];
  if (blockIdx.x == 0 && threadIdx.x < (sg.bytes & 15)) ((unsigned char*)sg.dst)[(n16 << 4) + threadIdx.x] = ((const unsigned char*)sg.src)[(n16 << 4) + threadIdx.x];
}

// up to 4 segments (src device / dst mapped host, both 16-byte aligned) in one launch
hipError_t launch_copy_out(const void* const* src, void* const* dst, const size_t* bytes, int n, hipStream_t stream) {
  CopySegs segs{};
  size_t most = 0;
  int m = 0;
  for (int i = 0; i < n && m < 4; i++)
    if (bytes[i]) { segs.s[m].src = src[i]; segs.s[m].dst = dst[i]; segs.s[m].bytes = bytes[i]; most = std::max(most, bytes[i]); m++; }
  if (m == 0) return hipSuccess;
  const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>(256, (most / 16 + 255) / 256));
  hipLaunchKernelGGL(k_copy_out, dim3(blocks, (unsigned)m), dim3(256), 0, stream, segs);
  return hipGetLastError();
}

// =============================================================================
// trace defaults: every window starts as "0 carts, mean shape"
// =============================================================================

template <typename Real>
__global__ void k_trace_fill(DevModelT<Real> m, WorkT<Real> w, unsigned n_windows) {
  const unsigned long long total = (unsigned long long)n_windows * m.dim;
  for (unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (unsigned long long)gridDim.x * blockDim.x) {
    const int d = (int)(idx % m.dim);
    w.tr_shape[idx] = m.mean_shape[d];
  }
}

template <typename Real>
hipError_t launch_trace_fill(const DevModelT<Real>& m, const WorkT<Real>& w, unsigned n_windows, hipStream_t stream) {
  hipLaunchKernelGGL(k_trace_fill<Real>, dim3(1024), dim3(256), 0, stream, m, w, n_windows);
  return hipGetLastError();
}
template hipError_t launch_trace_fill<float>(const DevModelT<float>&, const WorkT<float>&, unsigned, hipStream_t);
template hipError_t launch_trace_fill<double>(const DevModelT<double>&, const WorkT<double>&, unsigned, hipStream_t);

// =============================================================================
// Hardware-queue probe (lanes.cpp: StreamPool).  The runtime deals a process's streams to FOUR hardware queues and the
// packets of one queue run one after the other, so two lanes whose streams share a queue do not overlap at all
// (profiles/r06_hwq.txt).  A one-wave spin of `ticks` of the 100-MHz wall clock on the streams already placed, a time
// stamp on the new stream right behind them: the stamp lands before a spin's end unless it sits in that spin's queue.
__global__ void k_hwq_spin(long long ticks, unsigned long long* out) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  __hip_atomic_store(out, (unsigned long long)wall_clock64(), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_hwq_stamp(unsigned long long* out) {
  __hip_atomic_store(out, (unsigned long long)wall_clock64(), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_hwq_spin(long long ticks, unsigned long long* out, hipStream_t stream) {
  hipLaunchKernelGGL(k_hwq_spin, dim3(1), dim3(1), 0, stream, ticks, out);
  return hipGetLastError();
}
hipError_t launch_hwq_stamp(unsigned long long* out, hipStream_t stream) {
  hipLaunchKernelGGL(k_hwq_stamp, dim3(1), dim3(1), 0, stream, out);
  return hipGetLastError();
}


}  // namespace jda
