#!/usr/bin/env python3
"""Host profile of pipeline.forward_chunks_device (cProfile, 10 pairs) for STREAMS in the environment (default 2)."""
import sys, os, cProfile, pstats, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench, latency
from pats_amd import batch, ops, pipeline
dev = torch.device('cuda', 0)
h, w, if_local, outdoor, _, label = bench.WORKLOADS['megadepth']
cap = batch.Capacities(1, h, w, if_local=if_local)
gen = torch.Generator(device=dev); gen.manual_seed(1)
base = bench.BenchNets(ops, dev, gen, cap, h, w, batch=batch, channels_last=False)
nets = latency.PipelineNets(base, ops)
left, right = base.lefts[0:1], base.rights[0:1]
S = int(os.environ.get("STREAMS", "2"))
run = lambda: pipeline.forward_path(left, right, nets, if_local=True, device_counts=True, streams=S)
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): run()
print("streams %d: %.3f ms per pair (wall, no profiler)" % (S, (time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): run()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
