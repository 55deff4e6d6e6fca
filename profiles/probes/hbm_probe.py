"""What this MI355X sustains on plain streaming reads / copies (torch kernels): the practical ceiling behind roofline.peak."""
import torch, time
dev = torch.device("cuda:0")
n = 4 * 1024**3 // 4          # 4 GiB of fp32
x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
y = torch.empty_like(x)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timeit(lambda: x.sum())
print("read  (sum)  %.2f TB/s" % (x.numel() * 4 / t / 1e12))
t = timeit(lambda: y.copy_(x))
print("copy  r+w    %.2f TB/s (read+write bytes)" % (2 * x.numel() * 4 / t / 1e12))
t = timeit(lambda: torch.add(x, 1.0, out=y))
print("add   r+w    %.2f TB/s" % (2 * x.numel() * 4 / t / 1e12))
xi = x.view(torch.int32)
t = timeit(lambda: xi.max())
print("read  (max)  %.2f TB/s" % (x.numel() * 4 / t / 1e12))
