// Host-side cascade model: parsing and writing of the JDA binary model stream.
//
// Stream layout (little-endian, packed; reference README.md:84-111,
// src/jda/cascador.cpp:79-164, src/jda/cart.cpp:406-450, c/jda.c:486-716):
//   i32 mask | i32 T,K,landmark_n,tree_depth,stage_idx,cart_idx |
//   real mean_shape[2L] |
//   T x { K x { NODE x {i32 scale,i32 lm1,i32 lm2, real o1x,o1y,o2x,o2y, i32 th},
//               real score[LEAF], real th, real mean, real std },
//         real w[K*LEAF][2L] } |
//   i32 mask
// `real` is f64 in trainer files and f32 in files written by SerializeTo.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace jda {

struct SplitNode {       // one internal node of a cart
  int32_t scale;         // 0 origin, 1 half, 2 quarter
  int32_t lm1, lm2;      // landmark ids (NOT pre-doubled)
  double off[4];         // o1x, o1y, o2x, o2y as stored (f64 keeps f32 files exact too)
  int32_t th;            // feature threshold
};

struct HostModel {
  int T = 0, K = 0, L = 0, D = 0;
  int hdr_stage = 0, hdr_cart = 0;   // header ints 5,6 as found in the file
  int real_bytes = 0;                // 8 or 4: layout the stream was read from
  int node_n() const { return (1 << (D - 1)) - 1; }
  int leaf_n() const { return 1 << (D - 1); }
  int dim() const { return 2 * L; }
  long long carts() const { return (long long)T * K; }

  // Everything is kept in f64 exactly as read (an f32 file widens exactly);
  // the fp32 dialect narrows with a plain cast like reference c/jda.c:509-552.
  std::vector<double> mean_shape;    // [2L]
  std::vector<SplitNode> nodes;      // [T*K*NODE]
  std::vector<double> leaf_score;    // [T*K*LEAF]
  std::vector<double> cart_th, cart_mean, cart_std;  // [T*K]
  std::vector<double> w;             // [T][K*LEAF][2L]

  bool multi_scale() const;          // any split node reads the half/quarter image (cached after the first call)
  mutable int multi_cache = -1;
};

// Bytes of a well-formed stream with these dimensions.
long long model_stream_bytes(int T, int K, int L, int D, int real_bytes);

// real_bytes: 8, 4, or 0 = infer from the file size. Returns false and fills
// err on any failure (missing file, bad header, size mismatch, short read).
bool load_model(const char* path, int real_bytes, HostModel* out, std::string* err);

// Float layout with the header convention of reference c/jda.c:652-665.
bool save_model_f32(const HostModel& m, const char* path);

}  // namespace jda
