import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pats_amd import ops, synth
inp = synth.roofline_inputs()
d0, d1, ns = (torch.from_numpy(inp[k]).cuda() for k in ("d0", "d1", "ns"))
alpha = torch.tensor(float(inp["alpha"]), device="cuda")
S = ops.cost(d0, d1)
def run(): return ops.log_optimal_transport(S, alpha, ns, 200)
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t = timeit(run)
print("direct: %.3f ms, %.0f sweeps/s" % (t, 200e3 / t))
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): run()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    tg = timeit(g.replay)
    ref = run()
    print("graph: %.3f ms, %.0f sweeps/s, equal %s" % (tg, 200e3 / tg, bool(torch.equal(out, ref))))
except Exception as e:
    print("graph capture failed:", repr(e)[:400])
