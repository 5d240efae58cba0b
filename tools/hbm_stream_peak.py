"""Empirical HBM stream bandwidth of the box (quote the roofline against nominal AND measured peak).
Copy (read + write) and read-only (sum) over buffers well past the 256 MB Infinity Cache."""
import torch
n = 1 << 30                         # 1 Gi floats = 4 GiB per buffer
x = torch.empty(n, device="cuda").normal_()
y = torch.empty_like(x)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
t = timed(lambda: y.copy_(x)); print("copy  (4 GiB read + 4 GiB write): %.2f TB/s" % (2 * 4 * n / t / 1e12))
t = timed(lambda: x.sum());    print("read  (4 GiB, sum reduction)    : %.2f TB/s" % (4 * n / t / 1e12))
t = timed(lambda: y.fill_(1.)); print("write (4 GiB fill)              : %.2f TB/s" % (4 * n / t / 1e12))
m = 8 * 16 * 16384                   # the bench's [8,16,128,128] tensors: 8.4 MB, cache resident
a, b = torch.randn(m, device="cuda"), torch.empty(m, device="cuda")
t = timed(lambda: b.copy_(a), 50);   print("copy  8.4 MB tensor (cache resident, incl. launch): %.2f TB/s, %.1f us" % (2 * 4 * m / t / 1e12, t * 1e6))
