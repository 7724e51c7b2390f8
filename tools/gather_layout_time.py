import torch, time, sys
sys.path.insert(0, "/root/repo")
from pats_amd import ops
dev = "cuda"
R = 4096
g = torch.Generator(device=dev); g.manual_seed(1)
m0 = torch.randn((2*R,64,48,48), device=dev, generator=g); m1 = torch.randn((2*R,64,24,24), device=dev, generator=g); m2 = torch.randn((2*R,128,12,12), device=dev, generator=g)
title = torch.randn((R,8), device=dev); rub = torch.randn((R,264), device=dev)
out = torch.empty((2,R,264,145), device=dev)
def tm(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
a = tm(lambda: ops.fine_descriptors([m0,m1,m2], title, rub, out=out))
ref = out.clone()
c = [x.contiguous(memory_format=torch.channels_last) for x in (m0,m1,m2)]
b = tm(lambda: ops.fine_descriptors(c, title, rub, out=out))
print("fine  R=%d: nchw %.3f ms  nhwc %.3f ms  equal %s  (scaled to 20224 rows: %.2f / %.2f ms)" % (R, a, b, torch.equal(ref, out), a*20224/R, b*20224/R))
del m0, m1, m2, c, out, ref
P = 22000
ff0 = torch.randn((R,128,52,52), device=dev, generator=g); ff1 = torch.randn((R,128,52,52), device=dev, generator=g)
mk0 = torch.rand((P,2), device=dev, generator=g)*96; mk1 = torch.rand((P,2), device=dev, generator=g)*96
b_ids = torch.sort(torch.randint(0, R, (P,), device=dev, generator=g))[0]
kenc = torch.randn((128,64), device=dev); rub3 = torch.randn((R,128,144), device=dev)
o = (torch.empty((P,128,65), device=dev), torch.empty((P,128,65), device=dev))
a = tm(lambda: ops.third_descriptors(ff0, ff1, mk0, mk1, b_ids, kenc, rub3, out=o))
r0, r1 = o[0].clone(), o[1].clone()
c0, c1 = ff0.contiguous(memory_format=torch.channels_last), ff1.contiguous(memory_format=torch.channels_last)
b = tm(lambda: ops.third_descriptors(c0, c1, mk0, mk1, b_ids, kenc, rub3, out=o))
print("third P=%d: nchw %.3f ms  nhwc %.3f ms  equal %s  (scaled to 110136: %.2f / %.2f ms)" % (P, a, b, torch.equal(r0, o[0]) and torch.equal(r1, o[1]), a*110136/P, b*110136/P))
