"""Empirical HBM streaming ceilings on this device (context for the roofline fractions): device-to-device copy and a
read-only reduction, 2 GiB buffers, HIP-event timing."""
import torch
n = 1 << 29   # 2 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timed(lambda: b.copy_(a))
print("copy  : %.2f TB/s (read + write bytes)" % (2 * a.numel() * 4 / t / 1e12))
t = timed(lambda: a.sum())
print("read  : %.2f TB/s" % (a.numel() * 4 / t / 1e12))
t = timed(lambda: b.fill_(1.0))
print("write : %.2f TB/s" % (a.numel() * 4 / t / 1e12))
