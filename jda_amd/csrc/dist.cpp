// libjda_dist.so: gather of detection rows on rank 0 over RCCL (include/jda_dist.h).
// One process per GPU; the communicator, its stream and its buffers live behind the opaque handle.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/jda_dist.h"

namespace {

thread_local std::string g_err;

void fail(const std::string& m) {
  g_err = m;
  std::fprintf(stderr, "libjda_dist: %s\n", m.c_str());
}

#define D_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fail(std::string(#expr) + ": " + hipGetErrorString(e_)); return -1; } } while (0)
#define D_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { fail(std::string(#expr) + ": " + ncclGetErrorString(r_)); return -1; } } while (0)

struct Slot {                      // one pipelined gather in flight
  float* d_block = nullptr;        // [1 + block_rows][row_floats]  this rank's block
  float* d_all = nullptr;          // [world][1 + block_rows][row_floats]
  float* h_block = nullptr;        // pinned staging of d_block
  float* h_all = nullptr;          // pinned copy of d_all's first rows (counts) and, on rank 0, the rows
  hipEvent_t done = nullptr;
  bool busy = false;
  std::vector<float> rows;         // this rank's rows, kept for the fallback
};

struct Dist {
  int rank = 0, world = 1, device = 0, row_floats = 0, block_rows = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  Slot slot[2];
  int head = 0, tail = 0;          // ring of started gathers: collect at tail, start at head
  int* d_counts = nullptr; int* h_counts = nullptr;     // exact path
  float* d_send = nullptr; float* d_recv = nullptr;     // ... its grow-only row buffers
  size_t send_cap = 0, recv_cap = 0;
};

// grow-only device buffer of the exact path (no hipMalloc / hipFree -- device-wide synchronisations -- per gather)
int reserve_dev(float** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap) return 0;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  const size_t want = bytes + bytes / 2;
  D_HIP(hipMalloc((void**)p, want));
  *cap = want;
  return 0;
}

int gather_exact(Dist* d, const float* rows, int n_rows, float** all_rows, int* n_all) {
  *all_rows = nullptr; *n_all = 0;
  D_HIP(hipSetDevice(d->device));
  const size_t rowb = (size_t)d->row_floats * sizeof(float);
  // counts of every rank (ncclAllGather of 1 x i32 per rank)
  int* h = d->h_counts;
  h[d->world] = n_rows;
  D_HIP(hipMemcpyAsync(d->d_counts + d->world, h + d->world, sizeof(int), hipMemcpyHostToDevice, d->stream));
  D_NCCL(ncclAllGather(d->d_counts + d->world, d->d_counts, 1, ncclInt32, d->comm, d->stream));
  D_HIP(hipMemcpyAsync(h, d->d_counts, sizeof(int) * d->world, hipMemcpyDeviceToHost, d->stream));
  D_HIP(hipStreamSynchronize(d->stream));
  long long total = 0;
  for (int r = 0; r < d->world; r++) total += h[r];
  // rows: grouped ncclSend -> 0 / ncclRecv x (world - 1), through the handle's grow-only buffers.  Everything that can
  // fail locally (allocations, the copy of this rank's rows) happens BEFORE the group is opened; inside it every
  // call is attempted and the group is always closed, so a failing rank does not leave an open group behind (the
  // next collective would hang on it).
  if (n_rows > 0 && d->rank != 0) {
    if (reserve_dev(&d->d_send, &d->send_cap, (size_t)n_rows * rowb) != 0) return -1;
    D_HIP(hipMemcpyAsync(d->d_send, rows, (size_t)n_rows * rowb, hipMemcpyHostToDevice, d->stream));
  }
  if (d->rank == 0 && total > h[0] && reserve_dev(&d->d_recv, &d->recv_cap, (size_t)total * rowb) != 0) return -1;
  float* out = nullptr;
  if (d->rank == 0) {
    out = (float*)std::malloc(std::max<size_t>(1, (size_t)total * rowb));
    if (!out) { fail("out of memory"); return -1; }
  }
  ncclResult_t bad = ncclSuccess;
  auto keep = [&](ncclResult_t r) { if (r != ncclSuccess && bad == ncclSuccess) bad = r; };
  keep(ncclGroupStart());
  if (d->rank == 0) {
    long long off = h[0];
    for (int r = 1; r < d->world; r++) {
      if (h[r] > 0) keep(ncclRecv(d->d_recv + off * d->row_floats, (size_t)h[r] * d->row_floats, ncclFloat, r, d->comm, d->stream));
      off += h[r];
    }
  } else if (n_rows > 0) {
    keep(ncclSend(d->d_send, (size_t)n_rows * d->row_floats, ncclFloat, 0, d->comm, d->stream));
  }
  keep(ncclGroupEnd());
  if (bad != ncclSuccess) { std::free(out); fail(std::string("RCCL exact gather: ") + ncclGetErrorString(bad)); return -1; }
  hipError_t e = hipSuccess;
  if (d->rank == 0) {
    if (h[0] > 0) std::memcpy(out, rows, (size_t)h[0] * rowb);                       // rank 0's own rows never leave the host
    if (total > h[0])
      e = hipMemcpyAsync(out + (size_t)h[0] * d->row_floats, d->d_recv + (size_t)h[0] * d->row_floats,
                         (size_t)(total - h[0]) * rowb, hipMemcpyDeviceToHost, d->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
  if (e != hipSuccess) { std::free(out); fail(std::string("exact gather: ") + hipGetErrorString(e)); return -1; }
  if (d->rank == 0) { *all_rows = out; *n_all = (int)total; }
  return 0;
}

}  // namespace

extern "C" {

const char* jdaDistLastError(void) { return g_err.c_str(); }

int jdaDistUniqueId(unsigned char id[JDA_DIST_ID_BYTES]) {
  static_assert(JDA_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId u;
  D_NCCL(ncclGetUniqueId(&u));
  std::memcpy(id, u.internal, JDA_DIST_ID_BYTES);
  return 0;
}

void* jdaDistCreate(int rank, int world, const unsigned char id[JDA_DIST_ID_BYTES], int device, int row_floats, int block_rows) {
  g_err.clear();
  if (world < 1 || rank < 0 || rank >= world || !id || row_floats < 1 || block_rows < 1) { fail("bad arguments"); return nullptr; }
  Dist* d = new Dist();
  d->rank = rank; d->world = world; d->device = device; d->row_floats = row_floats; d->block_rows = block_rows;
  auto init = [&]() -> int {
    D_HIP(hipSetDevice(device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, JDA_DIST_ID_BYTES);
    D_NCCL(ncclCommInitRank(&d->comm, world, u, rank));
    D_HIP(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    const size_t blk = (size_t)(1 + block_rows) * row_floats * sizeof(float);
    for (Slot& s : d->slot) {
      D_HIP(hipMalloc((void**)&s.d_block, blk));
      D_HIP(hipMalloc((void**)&s.d_all, blk * world));
      D_HIP(hipHostMalloc((void**)&s.h_block, blk, hipHostMallocDefault));
      D_HIP(hipHostMalloc((void**)&s.h_all, blk * world, hipHostMallocDefault));
      D_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    D_HIP(hipMalloc((void**)&d->d_counts, sizeof(int) * (world + 1)));
    D_HIP(hipHostMalloc((void**)&d->h_counts, sizeof(int) * (world + 1), hipHostMallocDefault));
    return 0;
  };
  if (init() != 0) { jdaDistDestroy(d); return nullptr; }
  return d;
}

void jdaDistDestroy(void* dist) {
  Dist* d = (Dist*)dist;
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  for (Slot& s : d->slot) {
    if (s.d_block) (void)hipFree(s.d_block);
    if (s.d_all) (void)hipFree(s.d_all);
    if (s.h_block) (void)hipHostFree(s.h_block);
    if (s.h_all) (void)hipHostFree(s.h_all);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  if (d->d_counts) (void)hipFree(d->d_counts);
  if (d->d_send) (void)hipFree(d->d_send);
  if (d->d_recv) (void)hipFree(d->d_recv);
  if (d->h_counts) (void)hipHostFree(d->h_counts);
  if (d->comm) (void)ncclCommDestroy(d->comm);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

int jdaDistGatherRows(void* dist, const float* rows, int n_rows, float** all_rows, int* n_all) {
  g_err.clear();
  Dist* d = (Dist*)dist;
  if (!d || !all_rows || !n_all || n_rows < 0 || (n_rows > 0 && !rows)) { fail("bad arguments"); return -1; }
  if (d->head != d->tail) { fail("pipelined gathers are pending: collect them first"); return -1; }
  return gather_exact(d, rows, n_rows, all_rows, n_all);
}

int jdaGatherResults(void* dist, const jdaResult* results, int n, int frame_offset, float** all_rows, int* n_all) {
  g_err.clear();
  Dist* d = (Dist*)dist;
  if (!d || !results || n < 0) { fail("bad arguments"); return -1; }
  // the row format of jdaResultsPack (libjda.so), restated here so that this library does not link it
  long long total = 0;
  for (int i = 0; i < n; i++) total += results[i].n;
  std::vector<float> rows((size_t)total * d->row_floats);
  float* o = rows.data();
  for (int i = 0; i < n; i++) {
    const jdaResult& r = results[i];
    const int dim = 2 * r.landmark_n;
    if (5 + dim != d->row_floats) { fail("row_floats does not match 5 + 2*landmark_n"); return -1; }
    for (int j = 0; j < r.n; j++) {
      o[0] = (float)(frame_offset + i);
      o[1] = (float)r.bboxes[3 * j]; o[2] = (float)r.bboxes[3 * j + 1]; o[3] = (float)r.bboxes[3 * j + 2];
      o[4] = r.scores[j];
      std::memcpy(o + 5, r.shapes + (size_t)j * dim, dim * sizeof(float));
      o += d->row_floats;
    }
  }
  return jdaDistGatherRows(dist, rows.data(), (int)total, all_rows, n_all);
}

int jdaDistPending(void* dist) {
  Dist* d = (Dist*)dist;
  return d ? d->head - d->tail : 0;
}

int jdaDistGatherStart(void* dist, const float* rows, int n_rows) {
  g_err.clear();
  Dist* d = (Dist*)dist;
  if (!d || n_rows < 0 || (n_rows > 0 && !rows)) { fail("bad arguments"); return -1; }
  if (d->head - d->tail >= 2) { fail("two gathers are already in flight: collect one first"); return -1; }
  D_HIP(hipSetDevice(d->device));
  Slot& s = d->slot[d->head & 1];
  const size_t rowb = (size_t)d->row_floats * sizeof(float);
  const size_t blk = (size_t)(1 + d->block_rows) * rowb;
  s.rows.assign(rows, rows + (size_t)n_rows * d->row_floats);
  s.h_block[0] = (float)n_rows;                                   // counts up to 2^24 are exact in fp32
  const int carried = n_rows <= d->block_rows ? n_rows : 0;
  if (carried) std::memcpy(s.h_block + d->row_floats, rows, (size_t)carried * rowb);
  D_HIP(hipMemcpyAsync(s.d_block, s.h_block, (size_t)(1 + carried) * rowb, hipMemcpyHostToDevice, d->stream));
  D_NCCL(ncclAllGather(s.d_block, s.d_all, (size_t)(1 + d->block_rows) * d->row_floats, ncclFloat, d->comm, d->stream));
  // every rank needs the counts (first row of every block); rank 0 also the rows: one D2H of everything on rank 0,
  // of the first rows elsewhere
  if (d->rank == 0) {
    D_HIP(hipMemcpyAsync(s.h_all, s.d_all, blk * d->world, hipMemcpyDeviceToHost, d->stream));
  } else {
    D_HIP(hipMemcpy2DAsync(s.h_all, rowb, s.d_all, blk, rowb, (size_t)d->world, hipMemcpyDeviceToHost, d->stream));
  }
  D_HIP(hipEventRecord(s.done, d->stream));
  s.busy = true;
  d->head++;
  return 0;
}

int jdaDistGatherCollect(void* dist, float** all_rows, int* n_all) {
  g_err.clear();
  Dist* d = (Dist*)dist;
  if (!d || !all_rows || !n_all) { fail("bad arguments"); return -1; }
  *all_rows = nullptr; *n_all = 0;
  if (d->head == d->tail) { fail("no gather in flight"); return -1; }
  D_HIP(hipSetDevice(d->device));
  Slot& s = d->slot[d->tail & 1];
  D_HIP(hipEventSynchronize(s.done));
  d->tail++;
  s.busy = false;
  const size_t rowf = (size_t)d->row_floats;
  const size_t blkf = (size_t)(1 + d->block_rows) * rowf;
  std::vector<int> cnt(d->world);
  bool over = false;
  long long total = 0;
  for (int r = 0; r < d->world; r++) {
    const float c = d->rank == 0 ? s.h_all[(size_t)r * blkf] : s.h_all[(size_t)r * rowf];
    cnt[r] = (int)c;
    over = over || cnt[r] > d->block_rows;
    total += cnt[r];
  }
  if (over) {
    // some rank had more rows than a block carries: every rank saw the same counts and takes the exact path
    // (any later gather already started stays queued behind it on the communicator's stream)
    return gather_exact(d, s.rows.data(), (int)(s.rows.size() / rowf), all_rows, n_all);
  }
  if (d->rank != 0) return 0;
  float* out = (float*)std::malloc(std::max<size_t>(1, (size_t)total * rowf * sizeof(float)));
  if (!out) { fail("out of memory"); return -1; }
  size_t off = 0;
  for (int r = 0; r < d->world; r++) {
    if (cnt[r]) std::memcpy(out + off * rowf, s.h_all + (size_t)r * blkf + rowf, (size_t)cnt[r] * rowf * sizeof(float));
    off += (size_t)cnt[r];
  }
  *all_rows = out; *n_all = (int)total;
  return 0;
}

void jdaDistFree(float* rows) { std::free(rows); }

}  // extern "C"
