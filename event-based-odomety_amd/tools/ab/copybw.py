import torch, time
for mb in (256, 1200, 2400):
    n = mb * 1024 * 1024 // 8
    x = torch.ones(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("copy %5d MB: %.3f ms  %.2f TB/s (read + write)" % (mb, ms, 2 * n * 8 / ms / 1e9))
    z = torch.zeros(1, device="cuda")
    e0.record()
    for _ in range(20): y.zero_()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("fill %5d MB: %.3f ms  %.2f TB/s (write)" % (mb, ms, n * 8 / ms / 1e9))
    e0.record()
    for _ in range(20): s = x.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("sum  %5d MB: %.3f ms  %.2f TB/s (read)" % (mb, ms, n * 8 / ms / 1e9))
