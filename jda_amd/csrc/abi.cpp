// libjda.so: the extern "C" entry points of include/jda.h (reference interface: c/jda.h:18-68) and of the batch,
// trace and dialect-CPP extensions, on top of detect.cpp / tickets.cpp / ragged.cpp.
#include "detect.h"

// =============================================================================
// C ABI
// =============================================================================

using namespace jda;

// ---- exception barrier -------------------------------------------------------------------------------------------
// No C++ exception may cross `extern "C"`: the host side allocates (std::vector growth in the post-processing, the
// 512 MB of a big model's tables, std::string in the error channel, std::thread) and a std::bad_alloc that left
// jdaDetect* would end the caller's process in std::terminate.  The reference answers an allocation failure with NULL
// (c/jda.c:487-493); every entry below is a function-try-block whose handler reports through jdaGetLastError() and
// returns the entry's error value (NULL / -1 / an empty jdaResult).  Stack unwinding has given back what the call held
// (PlanPin; LaneSet, which waits for its lanes' streams BEFORE it returns them to the pool when it is destroyed by an
// exception); the detect entries also wait for the device as a whole, because the caller is free to release its frames
// as soon as the entry returns.  Helper threads (the ticket issuer, the ragged uploader, the post-processing workers)
// catch inside their bodies and hand the failure to the thread that joins them.
namespace {
void abi_exception(const char* fn, bool sync_device) noexcept {
  const char* what = "unknown C++ exception";
  char buf[200];
  try { throw; }
  catch (const std::bad_alloc&) { what = "out of host memory (std::bad_alloc)"; }
  catch (const std::exception& e) { std::snprintf(buf, sizeof buf, "%s", e.what()); what = buf; }
  catch (...) {}
  try { fail(std::string(fn) + ": " + what); }
  catch (...) { std::fprintf(stderr, "libjda: %s: %s\n", fn, what); }       // (not even the message could be allocated)
  if (sync_device) { (void)hipDeviceSynchronize(); (void)hipGetLastError(); }
}
// test hook (option test_throw; tests/test_abi.py): 1 = std::bad_alloc, 2 = std::runtime_error inside the entry
void maybe_throw(const Cascador* c) {
  if (c->kn.test_throw == 1) throw std::bad_alloc();
  if (c->kn.test_throw == 2) throw std::runtime_error("injected by test_throw");
}
}  // namespace
#define JDA_ABI_CATCH(ret) catch (...) { abi_exception(__func__, false); return ret; }
#define JDA_ABI_CATCH_SYNC(ret) catch (...) { abi_exception(__func__, true); return ret; }
#define JDA_ABI_CATCH_VOID catch (...) { abi_exception(__func__, false); }

extern "C" {

const char* jdaGetLastError(void) { return g_err.c_str(); }

static void* create_impl(const char* path, int real_bytes) {
  Cascador* c = nullptr;
  try {
    g_err.clear();
    c = new Cascador();
    c->kn.load();
    std::string err;
    if (!load_model(path, real_bytes, &c->hm, &err)) {
      g_err = err;   // reference returns NULL silently (c/jda.c:487-488); keep the reason retrievable
      delete c;
      return nullptr;
    }
    return c;
  } catch (...) {      // malloc failure: NULL like the reference (c/jda.c:489-493)
    abi_exception("jdaCascadorCreate", false);
    delete c;
    return nullptr;
  }
}

void* jdaCascadorCreateDouble(const char* model) { return create_impl(model, 8); }
void* jdaCascadorCreateFloat(const char* model) { return create_impl(model, 4); }
void* jdaCascadorCreate(const char* model) { return create_impl(model, 0); }

void jdaCascadorSerializeTo(void* cascador, const char* model) try {
  if (!cascador) return;
  (void)save_model_f32(((Cascador*)cascador)->hm, model);
} JDA_ABI_CATCH_VOID

void jdaCascadorRelease(void* cascador) try {
  Cascador* c = (Cascador*)cascador;
  if (!c) return;
  // Submitted batches nobody waited for are drained here (their helper threads joined, their streams synchronised by
  // Lane::destroy).  A call still running on another thread is the caller's error, as with the reference, whose
  // release frees what jdaDetect reads (c/jda.c:718-720); such a call is given ten seconds to return before the
  // lanes go -- a race at shutdown then ends in a late but orderly release instead of a use-after-free.
  for (int i = 0; c->pending && i < kTickets; i++) c->pending[i].join_issuer();
  {
    std::unique_lock<std::mutex> lk(c->mu);
    for (int i = 0; c->pending && i < kTickets; i++)
      if (c->pending[i].active && c->pending[i].lane) { c->pending[i].lane->busy = false; c->pending[i].active = false; }
    c->lane_cv.wait_for(lk, std::chrono::seconds(10), [&]() {
      for (auto& l : c->lanes) if (l->busy) return false;
      return true;
    });
  }
  if (c->dev_init) {
    (void)hipSetDevice(c->device);
    for (auto& l : c->lanes) l->destroy();
    c->streams.destroy();        // (aux, the upload stream and the lanes' streams are the pool's)
    c->aux = c->h2d = nullptr;
    for (auto& kv : c->plans) { if (kv.second.dp) (void)hipFree(kv.second.dp); if (kv.second.table) (void)hipFree(kv.second.table); }
    for (auto& b : c->plan_pool) { if (b.dp) (void)hipFree(b.dp); if (b.table) (void)hipFree(b.table); }
    c->mf.buf.release(); c->md.buf.release();
  }
  delete[] c->pending;
  delete c;
} JDA_ABI_CATCH_VOID

int jdaCascadorInfo(void* cascador, jdaModelInfo* info) try {
  if (!cascador || !info) return -1;
  const HostModel& h = ((Cascador*)cascador)->hm;
  info->T = h.T; info->K = h.K; info->landmark_n = h.L; info->tree_depth = h.D;
  info->multi_scale = h.multi_scale() ? 1 : 0; info->source_real_bytes = h.real_bytes;
  return 0;
} JDA_ABI_CATCH(-1)

int jdaSetSimilarityTransform(void* cascador, int on) try {
  Cascador* c = (Cascador*)cascador;
  if (!c) return -1;
  std::lock_guard<std::mutex> lock(c->mu);
  for (auto& l : c->lanes)
    if (l->busy) { fail("jdaSetSimilarityTransform while a call is running on this cascador"); return -1; }
  on = on ? 1 : 0;
  if (c->similarity != on) {
    c->similarity = on;
    c->md.ready = false;          // the fp64 node table depends on it (stage-0 offsets carry the transform)
  }
  return 0;
} JDA_ABI_CATCH(-1)

int jdaSetDevice(void* cascador, int device) try {
  Cascador* c = (Cascador*)cascador;
  if (!c) return -1;
  std::lock_guard<std::mutex> lock(c->mu);
  if (c->dev_init && c->device != device) { fail("jdaSetDevice after first use"); return -1; }
  c->device = device;
  return 0;
} JDA_ABI_CATCH(-1)

int jdaSetOption(void* cascador, const char* key, long long value) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !key) { fail("jdaSetOption: null cascador or key"); return -1; }
  std::lock_guard<std::mutex> lock(c->mu);
  for (auto& l : c->lanes)
    if (l->busy) { fail("jdaSetOption while a call is running or a submitted batch is pending on this cascador"); return -1; }
  for (auto& kv : c->plans)
    if (kv.second.pins) { fail("jdaSetOption while a call is running on this cascador"); return -1; }
  if (!c->kn.set(key, value)) { fail(std::string("jdaSetOption: unknown option or value out of range: '") + key + "'"); return -1; }
  // scan plans (tile shapes, table chunking) depend on the knobs: rebuild them on next use (no lane is busy, so
  // nothing runs on the old ones)
  for (auto& kv : c->plans) c->plan_pool.push_back({kv.second.dp, kv.second.table, kv.second.table_cap});
  c->plans.clear();
  return 0;
} JDA_ABI_CATCH(-1)

long long jdaGetOption(void* cascador, const char* key) try {
  Cascador* c = (Cascador*)cascador;
  long long v = 0;
  if (c && key && std::strncmp(key, "hwq_", 4) == 0 && std::strcmp(key, "hwq_place") != 0) {
    // read-only: what the stream pool found (host.h: StreamPool) -- hardware queues known, streams created, probes run,
    // and the most main streams of busy-or-free lanes that share one queue
    std::lock_guard<std::mutex> lk(c->streams.mu);
    if (std::strcmp(key, "hwq_queues") == 0) return (long long)c->streams.rep.size();
    if (std::strcmp(key, "hwq_streams") == 0) return c->streams.created;
    if (std::strcmp(key, "hwq_probes") == 0) return c->streams.probes;
    if (std::strcmp(key, "hwq_max_mains") == 0) { int m = 0; for (int x : c->streams.mains) m = std::max(m, x); return m; }
  }
  if (!c || !key || !c->kn.get(key, &v)) { fail("jdaGetOption: unknown option"); return -1; }
  return v;
} JDA_ABI_CATCH(-1)

int jdaCountWindows(int width, int height, float scale, int min_size, int max_size,
                    long long* n_windows, int* n_levels) try {
  ScanPlan sp; std::string err;
  if (!plan_dialect_c(width, height, scale, min_size, max_size, &sp, &err)) { g_err = err; return -1; }
  if (n_windows) *n_windows = sp.windows;
  if (n_levels) *n_levels = (int)sp.levels.size();
  return 0;
} JDA_ABI_CATCH(-1)

void jdaDetectOptionsInit(jdaDetectOptions* opt) try {
  if (!opt) return;
  std::memset(opt, 0, sizeof(*opt));
  opt->dialect = JDA_DIALECT_C; opt->nms = 1; opt->nms_overlap = 0.3f; opt->cpp_step = 5;
} JDA_ABI_CATCH_VOID

int jdaDetectBatchDevice(void* cascador, const unsigned char* d_frames, size_t frame_stride, int n,
                         int width, int height, float scale, float step, int min_size, int max_size,
                         float th, const jdaDetectOptions* opt, jdaResult* out) try {
  (void)step;  // ignored like the reference (c/jda.c:333)
  g_err.clear();
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchDevice runs dialect C; use jdaDetectBatchCpp"); return -1; }
  if (!cascador) { fail("null cascador"); return -1; }
  return detect_c_device((Cascador*)cascador, d_frames, frame_stride, n, width, height, scale, min_size, max_size, th, opt, out);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchSubmit(void* cascador, const unsigned char* d_frames, size_t frame_stride, int n,
                         int width, int height, float scale, float step, int min_size, int max_size,
                         float th, const jdaDetectOptions* opt) try {
  (void)step;
  g_err.clear();
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchSubmit runs dialect C"); return -1; }
  if (!cascador) { fail("null cascador"); return -1; }
  Cascador* c = (Cascador*)cascador;
  return submit_c_device(c, d_frames, frame_stride, n, width, height, scale, min_size, max_size, th, opt);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchSubmitHost(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                             float scale, float step, int min_size, int max_size, float th,
                             const jdaDetectOptions* opt) try {
  (void)step;
  g_err.clear();
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchSubmitHost runs dialect C"); return -1; }
  if (!cascador || !frames) { fail("null cascador or frames"); return -1; }
  if (width <= 0 || height <= 0) { fail("frame has no pixels"); return -1; }
  Cascador* c = (Cascador*)cascador;
  return submit_c_device(c, nullptr, 0, n, width, height, scale, min_size, max_size, th, opt, frames);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchWait(void* cascador, int ticket, jdaStats* stats, jdaResult* out) try {
  g_err.clear();
  if (!cascador) { fail("null cascador"); return -1; }
  Cascador* c = (Cascador*)cascador;
  return wait_c_device(c, ticket, stats, out);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatch(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                   float scale, float step, int min_size, int max_size, float th,
                   const jdaDetectOptions* opt, jdaResult* out) try {
  (void)step;
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !frames || !out || n < 0) { fail("bad arguments"); return -1; }
  if (width <= 0 || height <= 0) { fail("frame has no pixels"); return -1; }
  for (int i = 0; i < n; i++) if (!frames[i]) { fail("null frame pointer"); return -1; }
  maybe_throw(c);
  return detect_c_device(c, nullptr, 0, n, width, height, scale, min_size, max_size, th, opt, out, frames);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchRagged(void* cascador, const unsigned char* const* images, const int* widths, const int* heights, int n,
                         float scale, float step, int min_size, int max_size, float th,
                         const jdaDetectOptions* opt, jdaResult* out) try {
  (void)step;
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !images || !widths || !heights || !out || n < 0) { fail("bad arguments"); return -1; }
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchRagged runs dialect C"); return -1; }
  return detect_ragged(c, images, nullptr, nullptr, widths, heights, n, scale, min_size, max_size, th, opt, out);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchRaggedDevice(void* cascador, const unsigned char* d_base, const size_t* offsets, const int* widths,
                               const int* heights, int n, float scale, float step, int min_size, int max_size, float th,
                               const jdaDetectOptions* opt, jdaResult* out) try {
  (void)step;
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !d_base || !offsets || !widths || !heights || !out || n < 0) { fail("bad arguments"); return -1; }
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchRaggedDevice runs dialect C"); return -1; }
  return detect_ragged(c, nullptr, d_base, offsets, widths, heights, n, scale, min_size, max_size, th, opt, out);
} JDA_ABI_CATCH_SYNC(-1)

namespace {
// rows of a job -> the caller: the buffer itself, not a copy
int ragged_rows_out_f(Cascador* c, int rc, RowsOut<float>& v, float** rows, int* n_rows) {
  *rows = nullptr; *n_rows = 0;
  if (rc != 0) return rc;
  *n_rows = (int)(v.n / ((size_t)5 + c->hm.dim()));
  *rows = v.release();
  if (!*rows) { *n_rows = 0; fail("out of memory for the detection rows"); return -1; }
  return 0;
}
int ragged_rows_out_d(Cascador* c, int rc, RowsOut<double>& v, double** rows, int* n_rows) {
  *rows = nullptr; *n_rows = 0;
  if (rc != 0) return rc;
  *n_rows = (int)(v.n / ((size_t)6 + c->hm.dim()));
  *rows = v.release();
  if (!*rows) { *n_rows = 0; fail("out of memory for the detection rows"); return -1; }
  return 0;
}
}  // namespace

int jdaDetectBatchRaggedRows(void* cascador, const unsigned char* const* images, const int* widths, const int* heights, int n,
                             float scale, float step, int min_size, int max_size, float th, const jdaDetectOptions* opt,
                             int frame_offset, float** rows, int* n_rows) try {
  (void)step;
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !images || !widths || !heights || !rows || !n_rows || n < 0) { fail("bad arguments"); return -1; }
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchRaggedRows runs dialect C"); return -1; }
  RowsOut<float> v;
  const int rc = detect_ragged_rows(c, images, nullptr, nullptr, widths, heights, n, scale, min_size, max_size, th, opt, frame_offset, &v);
  return ragged_rows_out_f(c, rc, v, rows, n_rows);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchRaggedDeviceRows(void* cascador, const unsigned char* d_base, const size_t* offsets, const int* widths,
                                   const int* heights, int n, float scale, float step, int min_size, int max_size, float th,
                                   const jdaDetectOptions* opt, int frame_offset, float** rows, int* n_rows) try {
  (void)step;
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !d_base || !offsets || !widths || !heights || !rows || !n_rows || n < 0) { fail("bad arguments"); return -1; }
  if (opt && opt->dialect != JDA_DIALECT_C) { fail("jdaDetectBatchRaggedDeviceRows runs dialect C"); return -1; }
  RowsOut<float> v;
  const int rc = detect_ragged_rows(c, nullptr, d_base, offsets, widths, heights, n, scale, min_size, max_size, th, opt, frame_offset, &v);
  return ragged_rows_out_f(c, rc, v, rows, n_rows);
} JDA_ABI_CATCH_SYNC(-1)

void jdaRowsRelease(float* rows) { std::free(rows); }

jdaResult jdaDetect(void* cascador, unsigned char* data, int width, int height,
                    float scale, float step, int min_size, int max_size, float th) {
  Cascador* c = (Cascador*)cascador;
  jdaResult r;
  r.n = 0; r.landmark_n = c ? c->hm.L : 0; r.bboxes = nullptr; r.shapes = nullptr; r.scores = nullptr;
  try {
    if (!c || !data) { fail("jdaDetect: null cascador or image"); return empty_result(r.landmark_n); }
    const unsigned char* frames[1] = {data};
    if (jdaDetectBatch(cascador, frames, 1, width, height, scale, step, min_size, max_size, th, nullptr, &r) != 0) {
      jdaResultRelease(r);
      return empty_result(c->hm.L);
    }
    return r;
  } catch (...) {      // (jdaDetectBatch has its own barrier; what is left is the error string of the first branch)
    abi_exception("jdaDetect", false);
    r.n = 0; r.bboxes = nullptr; r.shapes = nullptr; r.scores = nullptr;
    return r;
  }
}

void jdaResultRelease(jdaResult result) {
  std::free(result.bboxes);
  std::free(result.shapes);
  std::free(result.scores);
}

int jdaTraceBatch(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                  float scale, int min_size, int max_size, int* carts_n, float* score,
                  unsigned int* path_hash, float* shapes) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !frames || n < 0) { fail("bad arguments"); return -1; }
  ScanPlan sp; std::string err;
  if (!plan_dialect_c(width, height, scale, min_size, max_size, &sp, &err)) { fail(err); return -1; }
  unsigned sb; std::memcpy(&sb, &scale, 4);
  PlanKey key{width, height, JDA_DIALECT_C, (int)sb, std::max(min_size, 24), max_size <= 0 ? -1 : max_size, 0ull};
  PlanEntry* pe = nullptr;
  if (!begin_call<float>(c, key, sp, JDA_DIALECT_C, &pe)) return -1;
  PlanPin pin{c, pe};
  LaneSet lanes(c);
  size_t stride = 0;
  if (!lanes.take(1) || !stage_frames(lanes.v[0], frames, n, (size_t)width * height, &stride, true)) return -1;
  TraceOut<float> tr{carts_n, score, path_hash, shapes};
  RunStats rs;
  if (!run_device<float>(c, lanes, pe, (const uint8_t*)lanes.v[0]->frames.p, stride, n, false, 0.f, nullptr, nullptr, &tr, &rs,
                         HostFrames{frames, (size_t)width * height})) return -1;
  return 0;
} JDA_ABI_CATCH_SYNC(-1)

int jdaBuildPyramid(void* cascador, const unsigned char* data, int width, int height,
                    unsigned char* half, int* hw, int* hh, unsigned char* quarter, int* qw, int* qh) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !data || width <= 0 || height <= 0) { fail("bad arguments"); return -1; }
  const float r = 1.f / sqrtf(2.f);
  const int w1 = (int)((float)width * r), h1 = (int)((float)height * r), w2 = width / 2, h2 = height / 2;
  if (hw) *hw = w1; if (hh) *hh = h1; if (qw) *qw = w2; if (qh) *qh = h2;
  if (!half && !quarter) return 0;
  if (!begin_device(c)) return -1;
  LaneSet lanes(c);
  if (!lanes.take(1)) return -1;
  Lane* ln = lanes.v[0];
  const unsigned char* frames[1] = {data};
  size_t stride = 0;
  if (!stage_frames(ln, frames, 1, (size_t)width * height, &stride)) return -1;
  auto one = [&](unsigned char* dst, int dw, int dh) -> bool {
    if (!dst || dw < 1 || dh < 1) return true;
    if (!ln->pyr.reserve((size_t)dw * dh + 256)) return false;
    JDA_HIP(launch_resize((const uint8_t*)ln->frames.p, stride, 1, width, height, (uint8_t*)ln->pyr.p,
                          (size_t)dw * dh, dw, dh, (float)(width - 1) / dw, (float)(height - 1) / dh, ln->stream));
    JDA_HIP(hipMemcpyAsync(dst, ln->pyr.p, (size_t)dw * dh, hipMemcpyDeviceToHost, ln->stream));
    JDA_HIP(hipStreamSynchronize(ln->stream));
    return true;
  };
  if (!one(half, w1, h1) || !one(quarter, w2, h2)) return -1;
  return 0;
} JDA_ABI_CATCH_SYNC(-1)

int jdaTraceBatchCpp(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                     int minimum_size, int step, double factor, int* carts_n, double* score,
                     unsigned int* path_hash, double* shapes) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !frames || n < 0) { fail("bad arguments"); return -1; }
  if (!cpp_model_complete(c)) return -1;
  ScanPlan sp; std::string err;
  if (!plan_dialect_cpp(width, height, minimum_size, step, factor, &sp, &err)) { fail(err); return -1; }
  unsigned long long fb; std::memcpy(&fb, &factor, 8);
  PlanKey key{width, height, JDA_DIALECT_CPP, minimum_size, step, c->similarity, fb};
  PlanEntry* pe = nullptr;
  if (!begin_call<double>(c, key, sp, JDA_DIALECT_CPP, &pe)) return -1;
  PlanPin pin{c, pe};
  LaneSet lanes(c);
  size_t stride = 0;
  if (!lanes.take(1) || !stage_frames(lanes.v[0], frames, n, (size_t)width * height, &stride, true)) return -1;
  TraceOut<double> tr{carts_n, score, path_hash, shapes};
  RunStats rs;
  if (!run_device<double>(c, lanes, pe, (const uint8_t*)lanes.v[0]->frames.p, stride, n, false, 0.0, nullptr, nullptr, &tr, &rs,
                          HostFrames{frames, (size_t)width * height})) return -1;
  return 0;
} JDA_ABI_CATCH_SYNC(-1)

int jdaResizeCv(void* cascador, const unsigned char* data, int width, int height, unsigned char* out, int ow, int oh) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !data || !out || width <= 0 || height <= 0 || ow <= 0 || oh <= 0) { fail("bad arguments"); return -1; }
  if (!begin_device(c)) return -1;
  LaneSet lanes(c);
  if (!lanes.take(1)) return -1;
  Lane* ln = lanes.v[0];
  const unsigned char* frames[1] = {data};
  size_t stride = 0;
  if (!stage_frames(ln, frames, 1, (size_t)width * height, &stride)) return -1;
  auto run = [&]() -> bool {
    if (!ln->pyr.reserve((size_t)ow * oh + 256)) return false;
    JDA_HIP(launch_resize_cv((const uint8_t*)ln->frames.p, stride, 1, width, height, (uint8_t*)ln->pyr.p,
                             (size_t)ow * oh, ow, oh, ln->stream));
    JDA_HIP(hipMemcpyAsync(out, ln->pyr.p, (size_t)ow * oh, hipMemcpyDeviceToHost, ln->stream));
    JDA_HIP(hipStreamSynchronize(ln->stream));
    return true;
  };
  return run() ? 0 : -1;
} JDA_ABI_CATCH_SYNC(-1)

static int detect_cpp_pyramid_impl(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                                   int origin_size, int half_size, int quarter_size, int step, double factor, double overlap, int nms,
                                   jdaStats* stats, jdaResultD* out) {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !frames || !out || n < 0) { fail("bad arguments"); return -1; }
  const int L = c->hm.L, dim = c->hm.dim();
  for (int i = 0; i < n; i++) { out[i].n = 0; out[i].landmark_n = L; out[i].rects = nullptr; out[i].shapes = nullptr; out[i].scores = nullptr; }
  if (origin_size < 1 || step < 1 || !(factor > 1.0)) { fail("origin_size/step must be positive and factor > 1"); return -1; }
  const bool multi = c->hm.multi_scale();
  if (multi && (half_size < 1 || quarter_size < 1 || half_size > 4096 || quarter_size > 4096)) {
    fail("method 0 on a model with scale != 0 split nodes needs the config's half_size and quarter_size: jdaDetectBatchCppPyramidMS "
         "(jdaDetectBatchCppPyramid serves scale==0 models)");
    return -1;
  }
  if (!cpp_model_complete(c)) return -1;
  if (!begin_device(c)) return -1;
  LaneSet lanes(c);
  if (!lanes.take(1)) return -1;
  Lane* ln = lanes.v[0];
  size_t stride0 = 0;
  if (!stage_frames(ln, frames, n, (size_t)width * height, &stride0)) return -1;

  // per level: rects (already scaled back), scores, normalised shapes, per frame, in scan order
  struct Cand { int rect[4]; double score; size_t shape_at; };
  std::vector<std::vector<Cand>> per_frame(n);
  std::vector<double> shape_pool;
  RunStats rs_total;
  long long patch_total = 0;
  // level images ping-pong inside one buffer; level 0 is the staged input
  struct LevelBuf : DevBuf { ~LevelBuf() { release(); } } levels;      // (freed on every way out, also an exception's)
  const size_t lvl_stride = ((size_t)width * height + 255) & ~(size_t)255;
  auto body = [&]() -> bool {
    if (!levels.reserve(2 * lvl_stride * (size_t)std::max(n, 1))) return false;
    const uint8_t* cur = (const uint8_t*)ln->frames.p;
    size_t cur_stride = stride0;
    int w = width, h = height, li = 0;
    double scale = 1.;
    while (w >= origin_size && h >= origin_size) {               // cascador.cpp:283
      ScanPlan sp; std::string err;
      if (!plan_single_level(w, h, origin_size, step, &sp, &err)) { fail(err); return false; }
      PlanKey key{w, h, 2 /* method 0 level */, origin_size, step, c->similarity, 0ull};
      PlanEntry* pe = nullptr;
      if (!begin_call<double>(c, key, sp, JDA_DIALECT_CPP, &pe)) return false;
      PlanPin pin{c, pe};
      RawDets<double> dets;
      RunStats rs;
      HostFrames hf;
      if (multi) { hf.patch_hs = half_size; hf.patch_qs = quarter_size; }
      if (!run_device<double>(c, lanes, pe, cur, cur_stride, n, false, 0.0, nullptr, &dets, nullptr, &rs, hf)) return false;
      rs_total.carts += rs.carts; rs_total.out += rs.out; rs_total.gpu_ms += rs.gpu_ms; rs_total.scan_ms += rs.scan_ms;
      rs_total.carts_scan += rs.carts_scan; rs_total.win_scan += rs.win_scan; rs_total.scan_launches += rs.scan_launches;
      rs_total.tail += rs.tail;
      for (int t = 0; t < c->hm.T; t++) rs_total.stage_done[t] += rs.stage_done[t];
      patch_total += sp.windows * n;
      for (size_t i = 0; i < dets.gid.size(); i++) {
        const WinRef wr = locate(sp, dets.gid[i]);
        Cand cd;
        int rx = wr.x, ry = wr.y, rw = wr.win, rh = wr.win;
        rx = (int)(rx * scale); ry = (int)(ry * scale); rw = (int)(rw * scale); rh = (int)(rh * scale);   // cascador.cpp:292-294
        cd.rect[0] = rx; cd.rect[1] = ry; cd.rect[2] = rw; cd.rect[3] = rh;
        cd.score = dets.score[i];
        cd.shape_at = shape_pool.size();
        shape_pool.insert(shape_pool.end(), dets.shape.begin() + i * dim, dets.shape.begin() + (i + 1) * dim);
        per_frame[wr.frame].push_back(cd);
      }
      scale *= factor;                                            // cascador.cpp:299
      const int nw = (int)(w / factor), nh = (int)(h / factor);   // cascador.cpp:300-301
      if (nw < 1 || nh < 1) break;
      uint8_t* nxt = (uint8_t*)levels.p + (size_t)(li & 1) * lvl_stride * (size_t)n;
      JDA_HIP(launch_resize_cv(cur, cur_stride, n, w, h, nxt, lvl_stride, nw, nh, ln->stream));   // cascador.cpp:302
      // (no host wait: the next level's pass is queued behind the resize on this lane's stream; a second lane the first
      // level took -- the first level has the most windows, so no later level takes one more -- waits for it on the device)
      if (lanes.v.size() > 1) {
        JDA_HIP(hipEventRecord(ln->ev_user, ln->stream));
        for (size_t l = 1; l < lanes.v.size(); l++) JDA_HIP(hipStreamWaitEvent(lanes.v[l]->stream, ln->ev_user, 0));
      }
      cur = nxt; cur_stride = lvl_stride; w = nw; h = nh; li++;
    }
    return true;
  };
  const bool ok = body();
  levels.release();
  if (!ok) return -1;

  const double t0 = now_ms();
  size_t total = 0;
  for (auto& v : per_frame) total += v.size();
  parallel_for(n, [&](int f) {
    const std::vector<Cand>& cs = per_frame[f];
    const size_t cnt = cs.size();
    std::vector<int> rc(cnt * 4);
    std::vector<double> sc(cnt);
    for (size_t i = 0; i < cnt; i++) { std::memcpy(&rc[4 * i], cs[i].rect, 16); sc[i] = cs[i].score; }
    std::vector<int> pick;
    if (nms) pick = nms_dialect_cpp(rc.data(), sc.data(), (int)cnt, overlap);
    else { pick.resize(cnt); std::iota(pick.begin(), pick.end(), 0); }
    jdaResultD& r = out[f];
    r.n = (int)pick.size(); r.landmark_n = L;
    r.rects = (int*)std::malloc(std::max<size_t>(1, pick.size() * 4) * sizeof(int));
    r.scores = (double*)std::malloc(std::max<size_t>(1, pick.size()) * sizeof(double));
    r.shapes = (double*)std::malloc(std::max<size_t>(1, pick.size() * dim) * sizeof(double));
    for (size_t i = 0; i < pick.size(); i++) {
      const int k = pick[i];
      std::memcpy(r.rects + 4 * i, &rc[4 * k], 4 * sizeof(int));
      r.scores[i] = sc[k];
      double* sh = r.shapes + i * dim;
      std::memcpy(sh, &shape_pool[cs[k].shape_at], dim * sizeof(double));
      relocate_dialect_cpp(sh, L, rc[4 * k], rc[4 * k + 1], rc[4 * k + 2], rc[4 * k + 3]);
    }
  }, total < 6000);
  fill_stats(stats, rs_total, patch_total, c->hm.T, c->hm.K, now_ms() - t0);
  return 0;
}

int jdaDetectBatchCppPyramid(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                             int origin_size, int step, double factor, double overlap, int nms,
                             jdaStats* stats, jdaResultD* out) try {
  return detect_cpp_pyramid_impl(cascador, frames, n, width, height, origin_size, 0, 0, step, factor, overlap, nms, stats, out);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchCppPyramidMS(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                               int origin_size, int half_size, int quarter_size, int step, double factor, double overlap, int nms,
                               jdaStats* stats, jdaResultD* out) try {
  return detect_cpp_pyramid_impl(cascador, frames, n, width, height, origin_size, half_size, quarter_size, step, factor, overlap, nms, stats, out);
} JDA_ABI_CATCH_SYNC(-1)

int jdaNmsC(const int* bboxes, const float* scores, int n, float overlap, unsigned char* keep) try {
  if (n < 0 || (n > 0 && (!bboxes || !scores || !keep))) return -1;
  std::vector<int> k = nms_dialect_c(bboxes, scores, n, overlap);
  std::memset(keep, 0, (size_t)n);
  for (int i : k) keep[i] = 1;
  return (int)k.size();
} JDA_ABI_CATCH(-1)

int jdaNmsCpp(const int* rects, const double* scores, int n, double overlap, int* picked) try {
  if (n < 0 || (n > 0 && (!rects || !scores || !picked))) return -1;
  std::vector<int> k = nms_dialect_cpp(rects, scores, n, overlap);
  std::copy(k.begin(), k.end(), picked);
  return (int)k.size();
} JDA_ABI_CATCH(-1)

int jdaResultsPack(const jdaResult* results, int n, int frame_offset, float* rows, int capacity_rows) try {
  if (!results || n < 0) return -1;
  long long total = 0;
  for (int i = 0; i < n; i++) total += results[i].n;
  if (!rows) return (int)total;
  if (total > capacity_rows) return -1;
  float* o = rows;
  for (int i = 0; i < n; i++) {
    const jdaResult& r = results[i];
    const int dim = 2 * r.landmark_n;
    for (int j = 0; j < r.n; j++) {
      o[0] = (float)(frame_offset + i);
      o[1] = (float)r.bboxes[3 * j]; o[2] = (float)r.bboxes[3 * j + 1]; o[3] = (float)r.bboxes[3 * j + 2];
      o[4] = r.scores[j];
      std::memcpy(o + 5, r.shapes + (size_t)j * dim, dim * sizeof(float));
      o += 5 + dim;
    }
  }
  return (int)total;
} JDA_ABI_CATCH(-1)

void jdaResultsRelease(jdaResult* results, int n) {
  if (!results) return;
  for (int i = 0; i < n; i++) {
    std::free(results[i].bboxes); std::free(results[i].shapes); std::free(results[i].scores);
    results[i].bboxes = nullptr; results[i].shapes = nullptr; results[i].scores = nullptr; results[i].n = 0;
  }
}

// Tile plan of a dialect-C call without touching a device (tests, tools): per level 10 ints
// {win, step, nx, ny, mode, tw, th, pitch, tiles_x, tiles_y}.  Returns the number of levels.
int jdaDebugPlanTiles(void* cascador, int width, int height, float scale, int min_size, int max_size, int* out, int cap_levels) try {
  Cascador* c = (Cascador*)cascador;
  if (!c) return -1;
  ScanPlan sp; std::string err;
  if (!plan_dialect_c(width, height, scale, min_size, max_size, &sp, &err)) { fail(err); return -1; }
  if ((int)sp.levels.size() > kMaxLevels) return -1;
  PlanEntry pe;
  bool s0_plain = true;
  const size_t n0 = (size_t)c->hm.K * c->hm.node_n();
  for (size_t i = 0; i < n0; i++) s0_plain = s0_plain && c->hm.nodes[i].scale == 0;
  assign_tiles(sp, c->hm, c->kn, s0_plain, 4, &pe);
  for (int i = 0; i < pe.hp.n_levels && i < cap_levels && out; i++) {
    const DevLevel& d = pe.hp.lv[i];
    const int v[10] = {d.win, d.step, d.nx, d.ny, d.tiled, d.tw, d.th, d.pitch, d.tiles_x, d.tiles_y};
    std::memcpy(out + 10 * i, v, sizeof v);
  }
  return pe.hp.n_levels;
} JDA_ABI_CATCH(-1)

long long jdaModelStreamBytes(int T, int K, int landmark_n, int tree_depth, int real_bytes) {
  return model_stream_bytes(T, K, landmark_n, tree_depth, real_bytes);
}

#ifdef JDA_BOUNDS_CHECK
namespace jda {
void jda_bc_read_k_scan(unsigned long long*); void jda_bc_read_k_scan_d(unsigned long long*); void jda_bc_read_k_scan_r(unsigned long long*);
void jda_bc_read_k_scan_dr(unsigned long long*); void jda_bc_read_k_scan_p(unsigned long long*); void jda_bc_read_k_finish(unsigned long long*);
void jda_bc_read_k_wide(unsigned long long*); void jda_bc_read_k_stage(unsigned long long*);
}
// bounds-check build only (libjda_bounds.so): per translation unit {first violation: site << 32 | source line, violations}
// since the last call -- out[16]; returns the total number of violations (kernels_common.h: Bc)
__attribute__((visibility("default"))) long long jdaDebugBoundsReport(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  void (*rd[8])(unsigned long long*) = {jda_bc_read_k_scan, jda_bc_read_k_scan_d, jda_bc_read_k_scan_r, jda_bc_read_k_scan_dr,
                                        jda_bc_read_k_scan_p, jda_bc_read_k_finish, jda_bc_read_k_wide, jda_bc_read_k_stage};
  long long total = 0;
  for (int i = 0; i < 8; i++) { unsigned long long v[2] = {0, 0}; rd[i](v); if (out) { out[2 * i] = v[0]; out[2 * i + 1] = v[1]; } total += (long long)v[1]; }
  return total;
}
#endif

#ifdef JDA_SCAN_TIMING
// timing build only: shader-clock stamps of the k_scan workgroups of the last float pass
__attribute__((visibility("default"))) int jdaDebugScanTiming(void* cascador, unsigned long long* out) {
  Cascador* c = (Cascador*)cascador;
  if (!c || c->lanes.empty()) return -1;
  const unsigned long long* dbg = c->lanes[0]->real_bytes == 8 ? c->lanes[0]->wd.dbg : c->lanes[0]->wf.dbg;
  if (!dbg) return -1;
  return hipMemcpy(out, dbg, sizeof(unsigned long long) * 65536 * 32, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

void jdaResultDRelease(jdaResultD result) {
  std::free(result.rects);
  std::free(result.shapes);
  std::free(result.scores);
}

int jdaDetectBatchCpp(void* cascador, const unsigned char* const* frames, int n, int width, int height,
                      int minimum_size, int step, double factor, double overlap, int nms,
                      jdaStats* stats, jdaResultD* out) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !frames || !out || n < 0) { fail("bad arguments"); return -1; }
  if (width <= 0 || height <= 0) { fail("frame has no pixels"); return -1; }
  for (int i = 0; i < n; i++) if (!frames[i]) { fail("null frame pointer"); return -1; }
  return detect_cpp_device(c, nullptr, 0, n, width, height, CppCall{minimum_size, step, factor, overlap, nms}, stats, out, frames);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchCppDevice(void* cascador, const unsigned char* d_frames, size_t frame_stride, int n, int width, int height,
                            int minimum_size, int step, double factor, double overlap, int nms,
                            jdaStats* stats, jdaResultD* out) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !d_frames || !out || n < 0) { fail("bad arguments"); return -1; }
  return detect_cpp_device(c, d_frames, frame_stride, n, width, height, CppCall{minimum_size, step, factor, overlap, nms}, stats, out);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchCppRagged(void* cascador, const unsigned char* const* images, const int* widths, const int* heights, int n,
                            int minimum_size, int step, double factor, double overlap, int nms,
                            jdaStats* stats, jdaResultD* out) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !images || !widths || !heights || !out || n < 0) { fail("bad arguments"); return -1; }
  return detect_ragged_cpp(c, images, nullptr, nullptr, widths, heights, n, CppCall{minimum_size, step, factor, overlap, nms}, stats, out);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchCppRaggedDevice(void* cascador, const unsigned char* d_base, const size_t* offsets, const int* widths,
                                  const int* heights, int n, int minimum_size, int step, double factor, double overlap, int nms,
                                  jdaStats* stats, jdaResultD* out) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !d_base || !offsets || !widths || !heights || !out || n < 0) { fail("bad arguments"); return -1; }
  return detect_ragged_cpp(c, nullptr, d_base, offsets, widths, heights, n, CppCall{minimum_size, step, factor, overlap, nms}, stats, out);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchCppRaggedRows(void* cascador, const unsigned char* const* images, const int* widths, const int* heights, int n,
                                int minimum_size, int step, double factor, double overlap, int nms, jdaStats* stats,
                                int frame_offset, double** rows, int* n_rows) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !images || !widths || !heights || !rows || !n_rows || n < 0) { fail("bad arguments"); return -1; }
  RowsOut<double> v;
  const int rc = detect_ragged_cpp_rows(c, images, nullptr, nullptr, widths, heights, n, CppCall{minimum_size, step, factor, overlap, nms}, stats, frame_offset, &v);
  return ragged_rows_out_d(c, rc, v, rows, n_rows);
} JDA_ABI_CATCH_SYNC(-1)

int jdaDetectBatchCppRaggedDeviceRows(void* cascador, const unsigned char* d_base, const size_t* offsets, const int* widths,
                                      const int* heights, int n, int minimum_size, int step, double factor, double overlap, int nms,
                                      jdaStats* stats, int frame_offset, double** rows, int* n_rows) try {
  g_err.clear();
  Cascador* c = (Cascador*)cascador;
  if (!c || !d_base || !offsets || !widths || !heights || !rows || !n_rows || n < 0) { fail("bad arguments"); return -1; }
  RowsOut<double> v;
  const int rc = detect_ragged_cpp_rows(c, nullptr, d_base, offsets, widths, heights, n, CppCall{minimum_size, step, factor, overlap, nms}, stats, frame_offset, &v);
  return ragged_rows_out_d(c, rc, v, rows, n_rows);
} JDA_ABI_CATCH_SYNC(-1)

void jdaRowsDRelease(double* rows) { std::free(rows); }

int jdaResultsDPack(const jdaResultD* results, int n, int frame_offset, double* rows, int capacity_rows) try {
  if (!results || n < 0) return -1;
  long long total = 0;
  for (int i = 0; i < n; i++) total += results[i].n;
  if (!rows) return (int)total;
  if (total > capacity_rows) return -1;
  double* o = rows;
  for (int i = 0; i < n; i++) {
    const jdaResultD& r = results[i];
    const int dim = 2 * r.landmark_n;
    for (int j = 0; j < r.n; j++) {
      o[0] = (double)(frame_offset + i);
      for (int k = 0; k < 4; k++) o[1 + k] = (double)r.rects[4 * j + k];
      o[5] = r.scores[j];
      std::memcpy(o + 6, r.shapes + (size_t)j * dim, dim * sizeof(double));
      o += 6 + dim;
    }
  }
  return (int)total;
} JDA_ABI_CATCH(-1)

void jdaResultsDRelease(jdaResultD* results, int n) {
  if (!results) return;
  for (int i = 0; i < n; i++) {
    std::free(results[i].rects); std::free(results[i].shapes); std::free(results[i].scores);
    results[i].rects = nullptr; results[i].shapes = nullptr; results[i].scores = nullptr; results[i].n = 0;
  }
}

}  // extern "C"
