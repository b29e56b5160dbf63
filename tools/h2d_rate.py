"""H2D / D2H rate of pinned copies of the callback path's sizes (torch as the plumbing)."""
import time
import torch

def rate(nbytes, reps=200, d2h=False):
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(10):
            (h.copy_(d, non_blocking=True) if d2h else d.copy_(h, non_blocking=True))
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            (h.copy_(d, non_blocking=True) if d2h else d.copy_(h, non_blocking=True))
        s.synchronize()
        dt = (time.perf_counter() - t0) / reps
    return dt

for nb in (131072, 1 << 20, 4 << 20, 16 << 20, 64 << 20):
    a, b = rate(nb), rate(nb, d2h=True)
    print("%9d B: H2D %.1f us (%.1f GB/s)  D2H %.1f us (%.1f GB/s)" % (nb, 1e6 * a, nb / a / 1e9, 1e6 * b, nb / b / 1e9))
