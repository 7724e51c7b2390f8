import torch, time
dev="cuda"
for gb in (2, 8):
    n = int(gb * 1e9 / 4)
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for name, fn, bytes_ in (("copy", lambda: b.copy_(a), 2 * n * 4), ("read-only sum", lambda: a.sum(), n * 4), ("fill", lambda: b.fill_(1.0), n * 4), ("mul_ in place", lambda: a.mul_(1.0001), 2*n*4)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%d GB %-14s %.3f ms  %.2f TB/s" % (gb, name, ms, bytes_ / ms / 1e9))
    del a, b
