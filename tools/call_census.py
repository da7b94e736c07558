"""GPU box: every library call of one eager cfg3 step by tag -- launches and HIP-event milliseconds (tools for the glue work:
which column sums, folds and small kernels a step still launches).  usage: python tools/call_census.py [cfg3|cfg4|cfg5]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
import bench
from mdmm import models, ops
from mdmm.harness import GradBucket, elbo_step
from mdmm.noise import PhiloxNoise
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg3']
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = cfg.model(models, dev)
m.noise = PhiloxNoise(seed=1)
opt = torch.optim.Adam(m.parameters(), lr=cfg.lr, fused=True)
bucket = GradBucket(m.parameters())
x, tg, mask, lengths = cfg.batch(cfg.T, cfg.B, 1234, dev)
kw = dict(targets=tg, n_points_global=sum(lengths))
if cfg.name != 'cfg4':
    kw['train_particles'] = 25
step = lambda: elbo_step(m, opt, bucket, x, mask, lengths, 1.0, cfg.rec, **kw)
for _ in range(2):
    step()
torch.cuda.synchronize()
ops.TIMER = ops.KernelTimer()
step()
torch.cuda.synchronize()
tot_n = tot_ms = 0
for k, (n, ms) in sorted(ops.TIMER.summary().items(), key=lambda kv: -kv[1][1]):
    print('%-56s x%-3d %8.3f ms' % (k, n, ms)); tot_n += n; tot_ms += ms
print('library calls: %d, %.2f ms of HIP-event spans (eager, single step)' % (tot_n, tot_ms))
