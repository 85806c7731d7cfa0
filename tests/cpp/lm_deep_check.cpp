// Host check of the device table layouts of jda_amd/csrc/kernels.h (no GPU): lm_index (level-major, every level) and
// lm_deep_index (the levels from `split` on, grouped per ancestor) must each be a bijection onto their arrays, and a
// path's deep records must lie inside one group of lm_deep_group() consecutive records.
#include <cstdio>
#include <vector>
#include "kernels.h"

int main() {
  int bad = 0;
  for (unsigned levels = 1; levels <= 8; levels++) {
    const unsigned node_n = (1u << levels) - 1u;
    for (unsigned K : {1u, 7u, 64u}) {
      std::vector<int> hit((size_t)K * node_n, 0);
      for (unsigned k = 0; k < K; k++)
        for (unsigned d = 0; d < levels; d++)
          for (unsigned n = (1u << d) - 1u; n < (2u << d) - 1u; n++) {
            const unsigned o = jda::lm_index(K, k, d, n);
            if (o >= hit.size() || hit[o]++) bad++;
          }
      for (unsigned split = 1; split <= levels; split++) {
        const unsigned per_cart = node_n - ((1u << split) - 1u), grp = split < levels ? jda::lm_deep_group(levels, split) : 0u;
        if (per_cart != (1u << (split - 1u)) * grp) bad++;
        std::vector<int> seen((size_t)K * per_cart, 0);
        for (unsigned k = 0; k < K; k++)
          for (unsigned d = split; d < levels; d++)
            for (unsigned n = (1u << d) - 1u; n < (2u << d) - 1u; n++) {
              const unsigned o = jda::lm_deep_index(k, d, n, levels, split);
              if (o >= seen.size() || seen[o]++) { bad++; continue; }
              // the ancestor on level split - 1 names the group
              unsigned a = n;
              for (unsigned u = d; u > split - 1u; u--) a = (a - 1u) / 2u;
              const unsigned anc = a - ((1u << (split - 1u)) - 1u);
              if (o / grp != (k << (split - 1u)) + anc) bad++;
            }
        for (int v : seen) if (v != 1) bad++;
      }
    }
  }
  std::printf("%d\n", bad);
  return bad ? 1 : 0;
}
