import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import models
from mdmm.harness import GradBucket, GraphedElboStep, elbo_step
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
T, B = 20, 64
g = torch.Generator().manual_seed(3)
x = {'x': torch.randn(T, B, 1, generator=g).to(dev), 'y': torch.randn(T, B, 1, generator=g).to(dev)}
mask = torch.ones(T, B, 1, dtype=torch.bool, device=dev)
def run(graph):
    junk = [torch.randn(1 << 24, device=dev) * 1e3 for _ in range(16)]; del junk      # dirty free blocks
    torch.manual_seed(0)
    m = models.MultiDMM(['x', 'y'], [1, 1], h_dim=32, z_dim=32, device=dev)
    m.noise = PhiloxNoise(seed=9)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, capturable=True)
    bucket = GradBucket(m.parameters())
    if graph:
        step = GraphedElboStep(m, opt, bucket, x, mask, [T] * B, 1.0, {'x': .5, 'y': .5}, train_particles=8, warmup=int(os.environ.get('WARM', '2')))
    else:
        step = lambda: elbo_step(m, opt, bucket, x, mask, [T] * B, 1.0, {'x': .5, 'y': .5}, train_particles=8)
    out = [float(step()) for _ in range(4)]
    torch.cuda.synchronize()
    for k, (mean, std, seen) in {}.items():
        print('      dbg', k, 'seen min/max/sum', float(seen.min()), float(seen.max()), float(seen.sum()), 'mean finite', bool(torch.isfinite(mean).all()), float(mean.abs().sum()))
    return out
for graph in (False, True):
    r = [run(graph) for _ in range(3)]
    print('graph' if graph else 'eager')
    for a in r: print('   ', ['%.4f' % v for v in a])
