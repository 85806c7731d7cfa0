// TEST INFRASTRUCTURE, never product: a stand-in for the eight RCCL entry points libjda_dist.so calls, so that
// jda_amd/csrc/dist.cpp -- the counts / offsets / grouping logic of the multi-GPU gather -- can run with SEVERAL ranks on a
// box with ONE GPU (RCCL itself refuses two ranks on one device).  Loaded with LD_PRELOAD in front of the real librccl by
// tests/test_dist_stub.py; libjda_dist.so itself is the shipped binary, untouched.
//
// Ranks are processes on the same device; a "collective" is: wait for the stream, copy the device buffer to a POSIX
// shared-memory slot, barrier, copy what the others left into the device buffer, barrier.  Synchronous where RCCL is
// asynchronous: stronger ordering, same data movement -- what is under test is who sends what to whom at which offset.
//   hipcc -O1 -fPIC -shared tests/c/rccl_stub.cpp -o tests/c/librccl_stub.so -lpthread -lrt
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {
constexpr size_t kSlot = 32u << 20;          // bytes a rank can put up per step
struct Ctl {
  std::atomic<int> init;                     // 0 nobody, 1 being initialised, 2 ready
  pthread_barrier_t bar;
  int world;
};
struct Msg { int dst; int pad; size_t off, bytes; };
struct Dir { int n; int pad; Msg m[62]; };    // a rank's outbox directory, at the head of its slot
struct Comm {
  int rank, world;
  char name[64];
  Ctl* ctl; unsigned char* base; size_t map_bytes;
  unsigned char* slot(int r) const { return base + sizeof(Ctl) + 4096 + (size_t)r * kSlot; }
};
struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t st; Comm* c; };
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;
Comm* g_comm = nullptr;                       // the tests make one communicator per process

size_t esize(ncclDataType_t t) {
  switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt64: case ncclUint64: case ncclFloat64: return 8; default: return 4; }
}
void wait_all(Comm* c) { pthread_barrier_wait(&c->ctl->bar); }

ncclResult_t run_ops(Comm* c) {
  Dir* dir = (Dir*)c->slot(c->rank);
  dir->n = 0;
  size_t off = sizeof(Dir);
  for (const Op& o : g_ops) {
    if (!o.send) continue;
    if (dir->n >= 62 || off + o.bytes > kSlot) return ncclInternalError;
    if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpy((unsigned char*)dir + off, o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    dir->m[dir->n++] = Msg{o.peer, 0, off, o.bytes};
    off += (o.bytes + 15) & ~(size_t)15;
  }
  wait_all(c);
  std::vector<int> cursor(c->world, 0);
  ncclResult_t rc = ncclSuccess;
  for (const Op& o : g_ops) {
    if (o.send) continue;
    const Dir* sd = (const Dir*)c->slot(o.peer);
    int& k = cursor[o.peer];
    while (k < sd->n && sd->m[k].dst != c->rank) k++;
    if (k >= sd->n || sd->m[k].bytes != o.bytes) { rc = ncclInvalidUsage; break; }     // (a recv nobody sent, or of another size)
    if (hipStreamSynchronize(o.st) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
    if (o.bytes && hipMemcpy(o.buf, (const unsigned char*)sd + sd->m[k].off, o.bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
    k++;
  }
  wait_all(c);
  return rc;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  static std::atomic<int> n{0};
  std::memset(id->internal, 0, sizeof id->internal);
  std::snprintf(id->internal, sizeof id->internal, "/jda_rccl_stub_%d_%d", (int)getpid(), n.fetch_add(1));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  Comm* c = new Comm();
  c->rank = rank; c->world = nranks;
  std::snprintf(c->name, sizeof c->name, "%s", id.internal);
  c->map_bytes = sizeof(Ctl) + 4096 + (size_t)nranks * kSlot;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return ncclSystemError;
  c->base = (unsigned char*)mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) return ncclSystemError;
  c->ctl = (Ctl*)c->base;
  int expect = 0;
  if (c->ctl->init.compare_exchange_strong(expect, 1)) {
    pthread_barrierattr_t a; pthread_barrierattr_init(&a); pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&c->ctl->bar, &a, (unsigned)nranks);
    c->ctl->world = nranks;
    c->ctl->init.store(2);
  } else {
    while (c->ctl->init.load() != 2) usleep(100);
  }
  wait_all(c);
  *comm = (ncclComm_t)c;
  g_comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclSuccess;
  wait_all(c);
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->base, c->map_bytes);
  if (g_comm == c) g_comm = nullptr;
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t st) {
  Comm* c = (Comm*)comm;
  const size_t bytes = count * esize(t);
  if (bytes > kSlot) return ncclInternalError;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slot(c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  wait_all(c);
  for (int r = 0; r < c->world; r++)
    if (hipMemcpy((unsigned char*)recv + (size_t)r * bytes, c->slot(r), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  wait_all(c);
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (--g_depth > 0) return ncclSuccess;
  // every rank of dist.cpp closes its group, also one with nothing to send: all of them meet at the two barriers
  if (!g_comm) return ncclInvalidUsage;
  const ncclResult_t rc = run_ops(g_comm);
  g_ops.clear();
  return rc;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
  g_ops.push_back(Op{true, const_cast<void*>(buf), count * esize(t), peer, st, (Comm*)comm});
  return ncclSuccess;
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
  g_ops.push_back(Op{false, buf, count * esize(t), peer, st, (Comm*)comm});
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success (stub)" : "error (rccl stub)"; }

}  // extern "C"
