"""How fast does this card take streaming writes?  (The training sweep's hidden stores: 1.07 GB per level-0 launch at 256 tiles.)
fill / copy / read-only reduction of a 4 GiB fp32 buffer, HIP-event timed."""
import torch
n = 1 << 30
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timed(lambda: a.fill_(1.0)); print("fill      : %.2f TB/s written" % (4 * n / t / 1e12))
t = timed(lambda: a.zero_());    print("memset    : %.2f TB/s written" % (4 * n / t / 1e12))
t = timed(lambda: b.copy_(a));   print("copy      : %.2f TB/s read + %.2f TB/s written" % (4 * n / t / 1e12, 4 * n / t / 1e12))
t = timed(lambda: a.sum());      print("reduction : %.2f TB/s read" % (4 * n / t / 1e12))
