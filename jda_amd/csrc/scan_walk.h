// Stage-0 tree walks over resolved nodes (S0Node): shared by k_scan.hip and k_scan_p.hip.
// Reference loop being replaced: c/jda.c:357-402 (stage 0: every window still holds the mean shape, so the
// feature offsets are resolved per (node, level) by k_prep_stage0).
#pragma once
#include "kernels_common.h"

namespace jda {

namespace {

// Feature test of one resolved stage-0 node for the window at `base`: true = go left
// (feature <= threshold, c/jda.c:391-393).  WIDE: the 21-bit packing (S0Node).
// bc: what `pix + index` may touch -- indices inside the LDS tile, or (global-pixel mode) addresses inside the frames
template <bool WIDE>
__device__ __forceinline__ bool s0_left(const S0Node r, const uint8_t* __restrict__ pix, int base, const Bc& bc = Bc(), bool bc_addr = false) {
  if (!WIDE) {
    JDA_BC(bc, base + (int)(r.lo & 0xffffu), 1, kBcScanPixLds); JDA_BC(bc, base + (int)(r.lo >> 16), 1, kBcScanPixLds);
    const int a = pix[base + (int)(r.lo & 0xffffu)];
    const int b = pix[base + (int)(r.lo >> 16)];
    return a - b <= (int)r.hi;
  }
  const uint32_t o1 = r.lo & 0x1fffffu;
  const uint32_t o2 = __builtin_amdgcn_alignbit(r.hi, r.lo, 21) & 0x1fffffu;
#ifdef JDA_BOUNDS_CHECK
  if (bc_addr) { JDA_BC_ADDR(bc, pix + base + (int)o1, 1, kBcScanPixGlb); JDA_BC_ADDR(bc, pix + base + (int)o2, 1, kBcScanPixGlb); }
  else { JDA_BC(bc, base + (int)o1, 1, kBcScanPixLds); JDA_BC(bc, base + (int)o2, 1, kBcScanPixLds); }
#endif
  const int a = pix[base + (int)o1];
  const int b = pix[base + (int)o2];
  return a - b + 256 <= (int)(r.hi >> 10);
}

// One cart of stage 0 for one window -> node index reached below the last split level:
// D-1 dependent (node record, 2 pixels) reads.  The node table is in LDS; pixels come
// from the LDS tile or from the frame through L1/L2.
// (Testing the root and both children at once -- 2 dependent round trips instead of 3,
// 8 pixel reads instead of 6 -- was measured 15 % SLOWER: the kernel is sensitive to LDS
// instruction count, see DESIGN.md.)
template <int DEPTH, bool WIDE>
__device__ __forceinline__ int scan_tree(const S0Node* __restrict__ tbl, const uint8_t* __restrict__ pix,
                                         int base, int depth_rt, const Bc& bc = Bc(), bool bc_addr = false) {
  int node = 0;
  const int levels = DEPTH > 0 ? DEPTH - 1 : depth_rt - 1;
#pragma unroll
  for (int d = 0; d < levels; d++) {
    const S0Node r = tbl[node];
    node = 2 * node + (s0_left<WIDE>(r, pix, base, bc, bc_addr) ? 1 : 2);
  }
  return node;
}

// N trees in lockstep, written level-major: the N node records of a level first, then the 2N pixels, then the N
// compares.  scan_tree called N times leaves the interleaving of the N independent chains to the instruction
// scheduler, which does it for the uniform-batch instantiation and -- two VGPRs of pressure later -- walks the trees
// one after the other in the RAGGED one (every LDS read followed by lgkmcnt(0): phases 30-60 % slower, stamps and ISA
// in profiles/r03_ragged_scan.txt).  Same reads, same compares.
template <int DEPTH, bool WIDE, int N>
__device__ __forceinline__ void scan_trees(const S0Node* __restrict__ t_nodes, int k, int node_n,
                                           const uint8_t* __restrict__ pix, int base, int depth_rt, int* lf,
                                           int kstride = 1, int kmax = 0x7fffffff, const Bc& bc = Bc(), bool bc_addr = false) {
  int node[N];
#pragma unroll
  for (int u = 0; u < N; u++) node[u] = 0;
  const int levels = DEPTH > 0 ? DEPTH - 1 : depth_rt - 1;
#pragma unroll
  for (int d = 0; d < levels; d++) {
    S0Node r[N];
#pragma unroll
    for (int u = 0; u < N; u++) r[u] = t_nodes[min(k + u * kstride, kmax) * node_n + node[u]];
    int a[N], b[N];
#pragma unroll
    for (int u = 0; u < N; u++) {
      if (!WIDE) {
        JDA_BC(bc, base + (int)(r[u].lo & 0xffffu), 1, kBcScanPixLds); JDA_BC(bc, base + (int)(r[u].lo >> 16), 1, kBcScanPixLds);
        a[u] = pix[base + (int)(r[u].lo & 0xffffu)];
        b[u] = pix[base + (int)(r[u].lo >> 16)];
      } else {
        const uint32_t o1 = r[u].lo & 0x1fffffu;
        const uint32_t o2 = __builtin_amdgcn_alignbit(r[u].hi, r[u].lo, 21) & 0x1fffffu;
#ifdef JDA_BOUNDS_CHECK
        if (bc_addr) { JDA_BC_ADDR(bc, pix + base + (int)o1, 1, kBcScanPixGlb); JDA_BC_ADDR(bc, pix + base + (int)o2, 1, kBcScanPixGlb); }
        else { JDA_BC(bc, base + (int)o1, 1, kBcScanPixLds); JDA_BC(bc, base + (int)o2, 1, kBcScanPixLds); }
#endif
        a[u] = pix[base + (int)o1];
        b[u] = pix[base + (int)o2];
      }
    }
#pragma unroll
    for (int u = 0; u < N; u++) {
      const bool left = WIDE ? (a[u] - b[u] + 256 <= (int)(r[u].hi >> 10)) : (a[u] - b[u] <= (int)r[u].hi);
      node[u] = 2 * node[u] + (left ? 1 : 2);
    }
  }
#pragma unroll
  for (int u = 0; u < N; u++) lf[u] = node[u] - node_n;
}

}  // namespace

}  // namespace jda
