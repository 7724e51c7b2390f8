import sys, time, torch
sys.path.insert(0, '/root/repo')
from pats_amd import ops
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
st = ops.masked_stream([c for c in range(n_cu) if c // 32 < 3])
x = torch.randn(8192, 8192, device='cuda')
for name, s in (("default", torch.cuda.current_stream()), ("masked", st), ("plain side", torch.cuda.Stream())):
    with torch.cuda.stream(s):
        for _ in range(3): y = x @ x
        torch.cuda.synchronize()
        for _ in range(20): y = x @ x          # deep queue
        t0 = time.perf_counter()
        for _ in range(100): torch.cuda.is_current_stream_capturing()
        t1 = time.perf_counter()
        for _ in range(100): ops._stream()
        t2 = time.perf_counter()
        for _ in range(100): torch.empty(1 << 20, dtype=torch.uint8, device='cuda')
        t3 = time.perf_counter()
        torch.cuda.synchronize()
    print(name, "is_capturing %.1f us  raw stream %.1f us  torch.empty %.1f us" % ((t1 - t0) * 1e4, (t2 - t1) * 1e4, (t3 - t2) * 1e4))
