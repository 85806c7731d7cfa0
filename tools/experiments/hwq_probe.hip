// Which of a process's streams share a hardware queue?  S streams created one after the other; for every ordered pair (i, j)
// a 300-us spin kernel goes to stream i and a time-stamp kernel to stream j right behind it: the stamp lands before the spin's
// end only if the two streams sit on different hardware queues (the runtime sets the barrier bit of every packet, so packets
// of one queue run one after the other).  Prints the classes of streams that serialise.
//   hipcc --offload-arch=gfx950 -O2 -o hwq_probe tools/experiments/hwq_probe.hip && ./hwq_probe [streams] [use_null_first] [prio_pattern]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(long long ticks, unsigned long long* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  *out = wall_clock64();
}
__global__ void stamp(unsigned long long* out) { *out = wall_clock64(); }
int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 8;
  const int use_null = argc > 2 ? atoi(argv[2]) : 1;
  const int prio = argc > 3 ? atoi(argv[3]) : 0;
  unsigned long long* h; CK(hipHostMalloc((void**)&h, 16, hipHostMallocDefault));
  if (use_null) { stamp<<<1, 1>>>(h); CK(hipDeviceSynchronize()); }
  int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  std::vector<hipStream_t> st(S);
  for (int i = 0; i < S; i++) {
    if (prio) { const int pr[3] = {0, hi, lo}; CK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, pr[i % 3])); }
    else CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    stamp<<<1, 1, 0, st[i]>>>(h); CK(hipStreamSynchronize(st[i]));      // used once, as a lane would be
  }
  printf("priority range %d..%d; %d streams%s%s\n", lo, hi, S, use_null ? ", null stream used first" : "", prio ? ", priorities 0/hi/lo in turn" : "");
  std::vector<int> cls(S, -1); int ncls = 0;
  for (int i = 0; i < S; i++) {
    printf("stream %d serialises with:", i);
    for (int j = 0; j < S; j++) {
      if (i == j) continue;
      h[0] = h[1] = 0;
      spin<<<1, 1, 0, st[i]>>>(30000, h);          // wall_clock64 ticks at 100 MHz: 300 us
      stamp<<<1, 1, 0, st[j]>>>(h + 1);
      CK(hipStreamSynchronize(st[i])); CK(hipStreamSynchronize(st[j]));
      const bool serial = h[1] >= h[0];
      if (serial) printf(" %d", j);
      if (serial && cls[i] < 0 && cls[j] >= 0) cls[i] = cls[j];
    }
    if (cls[i] < 0) cls[i] = ncls++;
    printf("\n");
  }
  printf("classes:"); for (int i = 0; i < S; i++) printf(" %d", cls[i]); printf("  (%d hardware queues in use by these streams)\n", ncls);
  return 0;
}
