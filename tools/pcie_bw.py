"""H2D copy rate of this box (pinned and pageable, one and two concurrent copies): the ceiling of the host-frame path."""
import time, torch
dev = torch.device("cuda", 0)
n = 256 << 20
pin = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
pag = torch.empty(n, dtype=torch.uint8)
dst = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
st = [torch.cuda.Stream() for _ in range(2)]
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
one = t(lambda: dst[0].copy_(pin[0], non_blocking=True))
print("pinned, one copy of 256 MiB: %.1f GB/s" % (n / one / 1e9))
def two():
    for i in range(2):
        with torch.cuda.stream(st[i]): dst[i].copy_(pin[i], non_blocking=True)
both = t(two)
print("pinned, two concurrent copies: %.1f GB/s aggregate" % (2 * n / both / 1e9))
p = t(lambda: dst[0].copy_(pag))
print("pageable, one copy: %.1f GB/s" % (n / p / 1e9))
small = torch.empty(78643200, dtype=torch.uint8).pin_memory(); d2 = torch.empty(78643200, dtype=torch.uint8, device=dev)
s = t(lambda: d2.copy_(small, non_blocking=True), 10)
print("pinned, one 78.6 MB batch: %.2f ms = %.1f GB/s -> at most %.2e windows/s through the host-frame path" % (s * 1e3, 78643200 / s / 1e9, 9790720 / s))
