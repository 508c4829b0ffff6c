# tools/dev/h2d_rate.py — host-to-device copy rates on this box: pageable vs pinned source, one 128-MiB copy vs 16-MiB pieces,
# and a host memcpy into a pinned buffer (what a staging thread would do).
import time, torch, numpy as np
n = 128 << 20
src = torch.empty(n, dtype=torch.uint8); src.random_(0, 255)
pin = torch.empty(n, dtype=torch.uint8).pin_memory(); pin.copy_(src)
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, k=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
ms = t(lambda: dst.copy_(src, non_blocking=True)); print("pageable, one copy      %.2f ms  %.1f GB/s" % (ms, n / ms / 1e6))
ms = t(lambda: dst.copy_(pin, non_blocking=True)); print("pinned,   one copy      %.2f ms  %.1f GB/s" % (ms, n / ms / 1e6))
P = 16 << 20
def pieces(s):
    for o in range(0, n, P): dst[o:o + P].copy_(s[o:o + P], non_blocking=True)
ms = t(lambda: pieces(src)); print("pageable, 16-MiB pieces %.2f ms  %.1f GB/s" % (ms, n / ms / 1e6))
ms = t(lambda: pieces(pin)); print("pinned,   16-MiB pieces %.2f ms  %.1f GB/s" % (ms, n / ms / 1e6))
t0 = time.perf_counter()
for _ in range(5): pin.copy_(src)
ms = (time.perf_counter() - t0) / 5 * 1e3; print("host memcpy pageable -> pinned (torch, %d threads) %.2f ms  %.1f GB/s" % (torch.get_num_threads(), ms, n / ms / 1e6))
torch.set_num_threads(1)
t0 = time.perf_counter()
for _ in range(5): pin.copy_(src)
ms = (time.perf_counter() - t0) / 5 * 1e3; print("host memcpy pageable -> pinned (1 thread) %.2f ms  %.1f GB/s" % (ms, n / ms / 1e6))
