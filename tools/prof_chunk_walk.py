import sys, cProfile, pstats, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench, latency
from pats_amd import batch, ops, pipeline
dev = torch.device('cuda', 0)
h, w, if_local, outdoor, _, label = bench.WORKLOADS['megadepth']
cap = batch.Capacities(1, h, w, if_local=if_local)
gen = torch.Generator(device=dev); gen.manual_seed(1)
base = bench.BenchNets(ops, dev, gen, cap, h, w, batch=batch, channels_last=False)
nets = latency.PipelineNets(base, ops)
left, right = base.lefts[0:1], base.rights[0:1]
run = lambda: pipeline.forward_path(left, right, nets, if_local=True, device_counts=True, streams=4)
for _ in range(3): run()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): run()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
