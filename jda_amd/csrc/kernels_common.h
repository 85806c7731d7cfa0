// Device-side helpers shared by the kernel translation units (k_*.hip): numeric dialects, wave
// primitives, global -> LDS staging (LDS-DMA), per-cart parameter record.
//
// Built with -ffp-contract=off -fno-gpu-flush-denormals-to-zero: every fp operation must round
// exactly like the reference's scalar C/C++ (no FMA, IEEE division, denormals kept).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <type_traits>

#include "kernels.h"

namespace jda {

namespace {

constexpr unsigned kFnvSeed = 2166136261u;
__device__ __forceinline__ unsigned fnv_step(unsigned h, int v) { return (h ^ (unsigned)v) * 16777619u; }

// float/double -> int the way the reference build does it (x86 cvttss2si /
// cvttsd2si: truncation, and INT_MIN for NaN or out-of-range values; the GPU
// conversion saturates instead).
__device__ __forceinline__ int to_int_x86(float v) {
  const int r = (int)v;
  return (fabsf(v) < 2147483648.f) ? r : INT_MIN;
}
__device__ __forceinline__ int to_int_x86(double v) {
  const int r = (int)v;
  return (v > -2147483649.0 && v < 2147483648.0) ? r : INT_MIN;
}

__device__ __forceinline__ int clamp_win(int v, int win) { return v < 0 ? 0 : (v >= win ? win - 1 : v); }

// ---- numeric dialects ---------------------------------------------------------

struct DialectC {           // reference c/jda.c
  using Real = float;
  using Node = NodeF;
  // c/jda.c:373-381: fp32 add, fp32 multiply by the window side, truncate
  static __device__ __forceinline__ int coord(float s, float o, int win) {
    const float v = (s + o) * (float)win;
    return to_int_x86(v);
  }
  // clamp_win(coord(s, o, win), win) in 4 instructions instead of 8 (the walks are VALU bound).
  // v_cvt_i32_f32 saturates: NaN -> 0, v >= 2^31 -> INT_MAX, v <= -2^31 -> INT_MIN, where x86
  // gives INT_MIN for all three, which the clamp then turns into 0.  No float below 2^31
  // converts to INT_MAX (the largest is 2^31 - 128), so INT_MAX marks exactly the positive
  // overflow; r + 1 wraps it to INT_MIN, and median(r + 1, 1, win) - 1 is the clamped pixel:
  // NaN -> 0, either overflow -> 0, r < 0 -> 0, r >= win -> win - 1.
  static __device__ __forceinline__ int pixel(float s, float o, int win) {
    const float v = (s + o) * (float)win;
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    const int r1 = (int)((unsigned)r + 1u);
    int m;
    asm("v_med3_i32 %0, %1, 1, %2" : "=v"(m) : "v"(r1), "v"(win));
    return m - 1;
  }
  // The same for an (x, y) pair, returned 1-BASED (kBias): the caller folds the -1s into its
  // tile base address.  The add and the multiply are packed (two independent IEEE operations).
  static constexpr int kBias = 1;
  static __device__ __forceinline__ void pixel_pair(float sx, float sy, float ox, float oy, int win, int* x, int* y) {
    typedef float V2 __attribute__((ext_vector_type(2)));
    const float fw = (float)win;
    const V2 v = (V2{sx, sy} + V2{ox, oy}) * V2{fw, fw};
    int rx, ry;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(rx) : "v"(v.x));
    asm("v_cvt_i32_f32 %0, %1" : "=v"(ry) : "v"(v.y));
    const int rx1 = (int)((unsigned)rx + 1u), ry1 = (int)((unsigned)ry + 1u);
    asm("v_med3_i32 %0, %1, 1, %2" : "=v"(*x) : "v"(rx1), "v"(win));
    asm("v_med3_i32 %0, %1, 1, %2" : "=v"(*y) : "v"(ry1), "v"(win));
  }
};

struct DialectCPP {         // reference src/jda (Validate / CalcFeatureValue)
  using Real = double;
  using Node = NodeD;
  // data.cpp:40-47: fp64, round half away from zero
  static __device__ __forceinline__ int coord(double s, double o, int win) {
    const double v = (s + o) * (double)win;
    return to_int_x86(round(v));
  }
  static __device__ __forceinline__ int pixel(double s, double o, int win) { return clamp_win(coord(s, o, win), win); }
  static constexpr int kBias = 0;
  static __device__ __forceinline__ void pixel_pair(double sx, double sy, double ox, double oy, int win, int* x, int* y) {
    *x = clamp_win(coord(sx, ox, win), win); *y = clamp_win(coord(sy, oy, win), win);
  }
};

__device__ __forceinline__ int wave_lane() { return threadIdx.x & 63; }

// value of lane j (wave-uniform j), for any lane mask
__device__ __forceinline__ int rl(int v, int j) { return __builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ float rl(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ double rl(double v, int j) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}

// ---- bounds-check build (`python -m jda_amd.build --bounds` -> libjda_bounds.so; never the product) ----------------------
// The pool refuses GPU AddressSanitizer runs, and output equality cannot see a read that is in bounds by luck.  With
// -DJDA_BOUNDS_CHECK every accessor below that computes an address from model or plan data -- tile loads, pixel gathers
// of the scans and the finishing kernels, the stage-0 tables, the weight-row gather -- checks it against the extent of
// what it reads (Bc: LDS tile bytes, the frames' device range, table entries) and records the first violation (site,
// source line) and their number in a per-translation-unit device word that jdaDebugBoundsReport collects.  In the
// product build Bc is empty and every check compiles to nothing.
struct Bc {
#ifdef JDA_BOUNDS_CHECK
  long long lo = 0, hi = (1ll << 62);      // valid indices (or addresses): [lo, hi)
  __host__ __device__ Bc() {}
  __host__ __device__ Bc(long long lo_, long long hi_) : lo(lo_), hi(hi_) {}
#else
  __host__ __device__ Bc() {}
  __host__ __device__ Bc(long long, long long) {}
#endif
};
enum BcSite : int { kBcScanPixLds = 1, kBcScanPixGlb = 2, kBcTileLoad = 3, kBcFinishPix = 4, kBcFinishTile = 5, kBcWRow = 6,
                    kBcNodeTable = 7, kBcWindowTileLoad = 8, kBcS0Table = 9, kBcStagePix = 10, kBcQueue = 11 };
#ifdef JDA_BOUNDS_CHECK
static __device__ unsigned long long jda_bc_word[2];      // [0] first violation: site << 32 | line; [1] violations
__device__ __forceinline__ void jda_bc_fail(int site, int line) {
  atomicCAS(&jda_bc_word[0], 0ull, ((unsigned long long)(unsigned)site << 32) | (unsigned)line);
  atomicAdd(&jda_bc_word[1], 1ull);
}
// index (or address) i, n elements from it, inside bc?
#define JDA_BC(bc, i, n, site) do { const long long i_ = (long long)(i); if (i_ < (bc).lo || i_ + (long long)(n) > (bc).hi) jda_bc_fail(site, __LINE__); } while (0)
#define JDA_BC_ADDR(bc, p, n, site) JDA_BC(bc, (long long)(uintptr_t)(p), n, site)
// one per translation unit that has device code: hands its word to abi.cpp and clears it
#define JDA_BC_READER(tu) \
  extern "C" void jda_bc_read_##tu(unsigned long long* out) { \
    unsigned long long v[2] = {0, 0}, z[2] = {0, 0}; \
    (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(jda_bc_word), sizeof v); (void)hipMemcpyToSymbol(HIP_SYMBOL(jda_bc_word), z, sizeof z); \
    out[0] = v[0]; out[1] = v[1]; }
#else
#define JDA_BC(bc, i, n, site) do { } while (0)
#define JDA_BC_ADDR(bc, p, n, site) do { } while (0)
#define JDA_BC_READER(tu)
#endif

// Orders this wave's LDS writes before its later LDS reads by OTHER lanes of the same wave (a wave's DS
// operations execute in order, so no instruction is needed -- only the compiler must not move them).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned long long lanes_below(int lane) {
  return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

__device__ __forceinline__ unsigned long long* shard_counter(unsigned long long* counters, int idx) {
  return counters + (size_t)(blockIdx.x % kCntShards) * kCntStride + idx;
}


template <typename Real>
struct CartPar {          // per-cart parameters as k_scan / k_stage read them from LDS
  Real th;
  Real norm;              // != 0 where (mean,std) != (0,1); next to th: the common case reads only this half
  Real mean, std;
};

// global -> LDS copy of n elements by the whole workgroup with all loads of a
// thread in flight before its first store (one memory latency, not one per element)
template <typename T, int BLOCK, int UNROLL>
__device__ __forceinline__ void stage_to_lds(T* __restrict__ dst, const T* __restrict__ src, int n, int tid) {
  for (int i0 = 0; i0 < n; i0 += BLOCK * UNROLL) {
    T v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i = i0 + u * BLOCK + tid;
      if (i < n) v[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i = i0 + u * BLOCK + tid;
      if (i < n) dst[i] = v[u];
    }
  }
}

// Stages the pw x ph pixel tile whose origin in the frame is (x0, y0) into LDS at row pitch
// `pitch` (a multiple of 16, at least (x0 & 15) + pw).  Returns the x offset of the tile origin
// inside the LDS rows.
//  * frame 16-byte aligned (base, stride, width, x0): LDS-DMA -- `global_load_lds_dwordx4` moves
//    16 bytes per lane straight from the frame into LDS (no VGPR round trip); a wave instruction
//    fills 1 KiB of consecutive LDS = 64 consecutive chunks of the row-major tile; all of a
//    wave's loads are in flight together.  The CALLER waits vmcnt(0) before its barrier.
//  * 4-byte aligned: coalesced dword rows, 8 rows in flight per wave;  * else: byte rows.
template <int BLOCK>
__device__ __forceinline__ int load_tile(unsigned char* lds_pix, const uint8_t* frames, size_t frame_stride,
                                         const uint8_t* img, int W, int x0, int y0, int pw, int ph, int pitch,
                                         int tid, const Bc& bc_frames = Bc(), const Bc& bc_lds = Bc()) {
  constexpr int NW = BLOCK / 64;
  const int lane = tid & 63, wv = tid >> 6;
  if (((W & 15) | (int)(frame_stride & 15) | (int)(((uintptr_t)frames) & 15)) == 0) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    // a tile origin that is not a multiple of 16 pixels starts at the 16-byte chunk below it; the
    // caller's pitch covers the lead-in (x0 & 15) + pw
    const int xa = x0 & ~15;
    const int cpr = pitch >> 4;                          // chunks per tile row
    const int nchunks = ph * cpr;
    const int maxcol = ((W - xa) >> 4) - 1;              // last chunk that ends inside the frame row
    int i = wv * 64 + lane;
    int row = i / cpr, col = i - row * cpr;
    const int dr = BLOCK / cpr, dc = BLOCK - dr * cpr;
    const uint8_t* g0 = img + (size_t)y0 * W + xa;
    for (int base = wv * 64; base < nchunks; base += BLOCK) {
      if (i < nchunks) {
        const uint8_t* g = g0 + (size_t)row * W + (min(col, maxcol) << 4);
        JDA_BC_ADDR(bc_frames, g, 16, kBcTileLoad); JDA_BC(bc_lds, ((long long)(base + lane)) << 4, 16, kBcTileLoad);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(lds_pix + (base << 4)), 16, 0, 0);
      }
      i += BLOCK; row += dr; col += dc;
      if (col >= cpr) { col -= cpr; row++; }
    }
    return x0 - xa;
  }
  const bool al4 = ((W & 3) == 0) && ((frame_stride & 3) == 0) && ((((uintptr_t)frames) & 3) == 0);
  if (al4) {
    const int x0a = x0 & ~3, xshift = x0 - x0a;
    const int ndw = (xshift + pw + 3) >> 2;
    const int w4 = W >> 2, p4 = pitch >> 2;
    const uint32_t* g = (const uint32_t*)(img + (size_t)y0 * W + x0a);
    uint32_t* d = (uint32_t*)lds_pix;
    for (int c0 = 0; c0 < ndw; c0 += 64) {
      const int c = c0 + lane;
      const bool cok = c < ndw;
      for (int r0 = wv * 8; r0 < ph; r0 += NW * 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) if (cok && r0 + u < ph) { JDA_BC_ADDR(bc_frames, &g[(size_t)(r0 + u) * w4 + c], 4, kBcTileLoad); v[u] = g[(size_t)(r0 + u) * w4 + c]; }
#pragma unroll
        for (int u = 0; u < 8; u++) if (cok && r0 + u < ph) { JDA_BC(bc_lds, ((long long)(r0 + u) * p4 + c) * 4, 4, kBcTileLoad); d[(r0 + u) * p4 + c] = v[u]; }
      }
    }
    return xshift;
  }
  const uint8_t* g = img + (size_t)y0 * W + x0;
  for (int c0 = 0; c0 < pw; c0 += 64) {
    const int c = c0 + lane;
    const bool cok = c < pw;
    for (int r0 = wv * 8; r0 < ph; r0 += NW * 8) {
      uint8_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) if (cok && r0 + u < ph) { JDA_BC_ADDR(bc_frames, &g[(size_t)(r0 + u) * W + c], 1, kBcTileLoad); v[u] = g[(size_t)(r0 + u) * W + c]; }
#pragma unroll
      for (int u = 0; u < 8; u++) if (cok && r0 + u < ph) { JDA_BC(bc_lds, (long long)(r0 + u) * pitch + c, 1, kBcTileLoad); lds_pix[(r0 + u) * pitch + c] = v[u]; }
    }
  }
  return 0;
}

// global -> LDS by LDS-DMA (16 bytes per lane, no VGPR round trip); falls back to stage_to_lds
// for sources that are not 16-byte aligned and for the tail.  The caller waits vmcnt(0) + barrier.
template <int BLOCK>
__device__ __forceinline__ void dma_to_lds(unsigned char* lds_dst, const void* __restrict__ src, int nbytes, int tid) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  const unsigned char* g = (const unsigned char*)src;
  int done = 0;
  if ((((uintptr_t)g) & 15) == 0) {
    const int chunks = nbytes >> 4;
    const int lane = tid & 63, wv = tid >> 6;
    for (int base = wv * 64; base < chunks; base += BLOCK) {
      if (base + lane < chunks)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + ((size_t)(base + lane) << 4)), (lds_ptr_t)(lds_dst + (base << 4)), 16, 0, 0);
    }
    done = chunks << 4;
  }
  stage_to_lds<unsigned char, BLOCK, 4>(lds_dst + done, g + done, nbytes - done, tid);
}

// lane-masked move: r = (bit `lane` of m) ? s : mine, one v_cndmask per dword with the mask in an SGPR pair (no compare,
// no VCC round trip)
__device__ __forceinline__ float capture_lane(float mine, float s, unsigned long long m) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(mine), "v"(s), "s"(m));
  return r;
}
__device__ __forceinline__ double capture_lane(double mine, double s, unsigned long long m) {
  int lo, hi;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(mine)), "v"(__double2loint(s)), "s"(m));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(mine)), "v"(__double2hiint(s)), "s"(m));
  return __hiloint2double(hi, lo);
}

// value of the lane below (lane l gets lane l-1's; lane 0 keeps `old`): one DPP move per dword, wave_shr:1
__device__ __forceinline__ float lane_below(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double lane_below(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Score recurrence of c/jda.c:395-399 over the carts held by lanes [jbeg, jend)
// of one 64-cart group, replayed strictly in cart order.  ls/th_k/mean_k/std_k/lf
// are per-lane values of cart (group base + lane).  Returns the lane of the
// rejecting cart or -1; score/hash are left as they stood at that cart.
//
// A systolic chain over the lanes: p[j] = f_j(p[j-1]) with f_j = "add cart j's leaf score (and normalise, where cart j
// does)", every lane applying its own f to the value of the lane below (DPP wave_shr:1), lane jbeg pinned to
// f(score).  After i rounds the lanes up to jbeg + i hold exactly the scores of the scalar loop -- each is computed from
// its predecessor's final value by the same operations in the same order; the earlier, unfinished values are
// overwritten.  jend - jbeg - 1 rounds of one DPP add (+ one masked move) instead of a read-lane, an add, a compare
// and a mask update per cart; then ALL lanes compare at once and the first set bit of the ballot is the rejecting cart.
// (r02's form cost 84 clocks per cart -- 48 k clocks per 540-cart stage; this one 40, measured with shader-clock
// stamps: tools/wide_timing.py.  Also measured, slower or equal: a wave-uniform running score fed by v_readlane with a
// masked move per cart (52-60 clocks), row_shr/row_bcast steps instead of wave_shr (equal).)
template <typename Real, bool TRACE>
__device__ __forceinline__ int replay_scores(Real& score, unsigned& hash, Real ls, Real th_k, Real mean_k, Real std_k,
                                             unsigned long long normmask, int lf, int jbeg, int jend) {
  jbeg = __builtin_amdgcn_readfirstlane(jbeg);                        // (wave-uniform by contract; values that come out of vector
  jend = __builtin_amdgcn_readfirstlane(jend);                        // loads are not known to be: the lane mask below lives in SGPRs)
  if (jbeg >= jend) return -1;                                        // no cart of this group is still to be scored
  const int lane = wave_lane();
  const bool norm = (normmask >> lane) & 1ull;
  auto f_plain = [&](Real below) { return below + ls; };              // c/jda.c:396
  auto f_norm = [&](Real below) {                                     // ... and c/jda.c:397 where this lane's cart normalises
    const Real v = below + ls;
    const Real n = (v - mean_k) / std_k;
    return norm ? n : v;
  };
  const unsigned long long pin = 1ull << jbeg;
  const int rounds = jend - jbeg - 1;
  Real p;
  if (normmask == 0ull && jbeg == 0 && sizeof(Real) == 4) {
    // the common case in ONE dependent instruction per cart: v_add_f32_dpp adds the lane's leaf score to the value of
    // the lane below; lane 0 has no lane below, so the instruction leaves it alone (bound_ctrl off) and it keeps
    // score + ls.  (s_nop 1: a DPP operand must have been written two wait states earlier.)
    float pf = (float)f_plain(score);
    const float lsf = (float)ls;
    if (rounds == 63) {
#pragma unroll
      for (int i = 0; i < 63; i++)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pf) : "v"(lsf));
    } else {
      for (int i = 0; i < rounds; i++)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pf) : "v"(lsf));
    }
    p = (Real)pf;
  } else if (normmask == 0ull) {
    const Real first = f_plain(score);                                // what lane jbeg holds
    p = first;
    for (int i = 0; i < rounds; i++) p = capture_lane(f_plain(lane_below(first, p)), first, pin);
  } else {                                                            // rare: a group with a normalising cart
    const Real first = f_norm(score);
    p = first;
    for (int i = 0; i < rounds; i++) p = capture_lane(f_norm(lane_below(first, p)), first, pin);
  }
  const unsigned long long rej = __ballot(lane >= jbeg && lane < jend && p < th_k);   // c/jda.c:399
  if (rej == 0ull) {
    if (TRACE) {
      for (int j = jbeg; j < jend; j++) hash = fnv_step(hash, rl(lf, j));
    }
    score = rl(p, jend - 1);
    return -1;
  }
  const int jr = __ffsll((long long)rej) - 1;
  score = rl(p, jr);                                                  // the score as it stood at the rejecting cart
  if (TRACE) {
    for (int j = jbeg; j <= jr; j++) hash = fnv_step(hash, rl(lf, j));
  }
  return jr;
}

}  // namespace

}  // namespace jda
