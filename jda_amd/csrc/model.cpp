// Model stream reader/writer. See model.h for the layout and its reference citations.
#include "model.h"

#include <cstdio>
#include <cstring>
#include <memory>

namespace jda {

bool HostModel::multi_scale() const {
  if (multi_cache < 0) {              // asked on every detect call: scan the nodes once
    int any = 0;
    for (const SplitNode& n : nodes)
      if (n.scale != 0) { any = 1; break; }
    multi_cache = any;
  }
  return multi_cache != 0;
}

long long model_stream_bytes(int T, int K, int L, int D, int rb) {
  const long long node_n = (1LL << (D - 1)) - 1, leaf_n = 1LL << (D - 1);
  const long long cart = node_n * (4 * 4 + 4 * rb) + leaf_n * rb + 3 * rb;
  const long long stage = K * cart + (long long)K * leaf_n * 2 * L * rb;
  return 7 * 4 + 2LL * L * rb + T * stage + 4;
}

namespace {

struct Cursor {
  const unsigned char* p;
  const unsigned char* end;
  bool ok = true;
  int32_t i32() {
    int32_t v = 0;
    if (end - p < 4) { ok = false; return 0; }
    std::memcpy(&v, p, 4); p += 4; return v;
  }
  double real(int rb) {
    if (end - p < rb) { ok = false; return 0; }
    double v;
    if (rb == 8) { std::memcpy(&v, p, 8); }
    else { float f; std::memcpy(&f, p, 4); v = f; }
    p += rb; return v;
  }
};

bool sane_dims(int T, int K, int L, int D) {
  return T >= 1 && T <= 16 && K >= 1 && K <= (1 << 20) && L >= 1 && L <= 4096 && D >= 2 && D <= 12;
}

}  // namespace

bool load_model(const char* path, int real_bytes, HostModel* out, std::string* err) {
  FILE* f = path ? std::fopen(path, "rb") : nullptr;
  if (!f) { if (err) *err = std::string("cannot open model file: ") + (path ? path : "(null)"); return false; }
  std::fseek(f, 0, SEEK_END);
  const long long size = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (size < 32) { std::fclose(f); if (err) *err = "model file too short for a header"; return false; }
  std::unique_ptr<unsigned char[]> buf(new unsigned char[size]);
  const size_t got = std::fread(buf.get(), 1, (size_t)size, f);
  std::fclose(f);
  if ((long long)got != size) { if (err) *err = "short read on model file"; return false; }

  Cursor c{buf.get(), buf.get() + size};
  (void)c.i32();  // mask
  HostModel m;
  m.T = c.i32(); m.K = c.i32(); m.L = c.i32(); m.D = c.i32();
  m.hdr_stage = c.i32(); m.hdr_cart = c.i32();
  if (!sane_dims(m.T, m.K, m.L, m.D)) {
    if (err) *err = "model header holds implausible dimensions (T,K,landmark_n,tree_depth)";
    return false;
  }
  const long long s8 = model_stream_bytes(m.T, m.K, m.L, m.D, 8);
  const long long s4 = model_stream_bytes(m.T, m.K, m.L, m.D, 4);
  int rb = real_bytes;
  if (rb == 0) rb = (size == s8) ? 8 : (size == s4 ? 4 : 0);
  if (rb != 4 && rb != 8) { if (err) *err = "file size matches neither the f64 nor the f32 layout of its header"; return false; }
  if (size != (rb == 8 ? s8 : s4)) {
    if (err) *err = "file size does not match the header for the requested real type";
    return false;
  }
  m.real_bytes = rb;

  const int node_n = m.node_n(), leaf_n = m.leaf_n(), dim = m.dim();
  const long long carts = m.carts();
  m.mean_shape.resize(dim);
  for (int i = 0; i < dim; i++) m.mean_shape[i] = c.real(rb);
  m.nodes.resize(carts * node_n);
  m.leaf_score.resize(carts * leaf_n);
  m.cart_th.resize(carts); m.cart_mean.resize(carts); m.cart_std.resize(carts);
  m.w.resize((size_t)m.T * m.K * leaf_n * dim);
  for (int t = 0; t < m.T; t++) {
    for (int k = 0; k < m.K; k++) {
      const long long ci = (long long)t * m.K + k;
      for (int i = 0; i < node_n; i++) {
        SplitNode& n = m.nodes[ci * node_n + i];
        n.scale = c.i32(); n.lm1 = c.i32(); n.lm2 = c.i32();
        for (int j = 0; j < 4; j++) n.off[j] = c.real(rb);
        n.th = c.i32();
      }
      for (int i = 0; i < leaf_n; i++) m.leaf_score[ci * leaf_n + i] = c.real(rb);
      m.cart_th[ci] = c.real(rb); m.cart_mean[ci] = c.real(rb); m.cart_std[ci] = c.real(rb);
    }
    double* w = &m.w[(size_t)t * m.K * leaf_n * dim];
    const size_t cnt = (size_t)m.K * leaf_n * dim;
    if (rb == 8) {
      if ((size_t)(c.end - c.p) < cnt * 8) { c.ok = false; break; }
      std::memcpy(w, c.p, cnt * 8); c.p += cnt * 8;
    } else {
      for (size_t i = 0; i < cnt; i++) w[i] = c.real(4);
    }
  }
  (void)c.i32();  // trailing mask
  if (!c.ok) { if (err) *err = "model stream ended early"; return false; }
  for (const SplitNode& n : m.nodes) {
    if (n.scale < 0 || n.scale > 2 || n.lm1 < 0 || n.lm1 >= m.L || n.lm2 < 0 || n.lm2 >= m.L) {
      if (err) *err = "model holds a split node with scale or landmark id out of range";
      return false;
    }
  }
  *out = std::move(m);
  return true;
}

bool save_model_f32(const HostModel& m, const char* path) {
  FILE* f = path ? std::fopen(path, "wb") : nullptr;
  if (!f) return false;
  std::vector<unsigned char> buf;
  buf.reserve((size_t)model_stream_bytes(m.T, m.K, m.L, m.D, 4));
  auto put_i = [&](int32_t v) { unsigned char b[4]; std::memcpy(b, &v, 4); buf.insert(buf.end(), b, b + 4); };
  auto put_f = [&](double d) { float v = (float)d; unsigned char b[4]; std::memcpy(b, &v, 4); buf.insert(buf.end(), b, b + 4); };
  const int node_n = m.node_n(), leaf_n = m.leaf_n(), dim = m.dim();
  put_i(0); put_i(m.T); put_i(m.K); put_i(m.L); put_i(m.D);
  put_i(m.T + 1); put_i(-1);  // header convention of reference c/jda.c:662-665
  for (int i = 0; i < dim; i++) put_f(m.mean_shape[i]);
  for (int t = 0; t < m.T; t++) {
    for (int k = 0; k < m.K; k++) {
      const long long ci = (long long)t * m.K + k;
      for (int i = 0; i < node_n; i++) {
        const SplitNode& n = m.nodes[ci * node_n + i];
        put_i(n.scale); put_i(n.lm1); put_i(n.lm2);
        for (int j = 0; j < 4; j++) put_f(n.off[j]);
        put_i(n.th);
      }
      for (int i = 0; i < leaf_n; i++) put_f(m.leaf_score[ci * leaf_n + i]);
      put_f(m.cart_th[ci]); put_f(m.cart_mean[ci]); put_f(m.cart_std[ci]);
    }
    const double* w = &m.w[(size_t)t * m.K * leaf_n * dim];
    const size_t cnt = (size_t)m.K * leaf_n * dim;
    for (size_t i = 0; i < cnt; i++) put_f(w[i]);
  }
  put_i(0);
  const bool ok = std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
  std::fclose(f);
  return ok;
}

}  // namespace jda
