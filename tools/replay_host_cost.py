"""GPU box: host time of one GraphedElboStep replay call (enqueue only) against the step's device time, cfg3 B = 256."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
import bench
from mdmm import models
from mdmm.harness import GradBucket, GraphedElboStep
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
cfg = bench.CONFIGS['cfg3']
torch.manual_seed(0)
model = cfg.model(models, dev)
model.noise = PhiloxNoise(seed=1000)
opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, capturable=True, fused=True)
bucket = GradBucket(model.parameters())
x, tg, mask, lengths = cfg.batch(cfg.T, cfg.B, 1234, dev)
step = GraphedElboStep(model, opt, bucket, x, mask, lengths, 1.0, cfg.rec, warmup=3, targets=tg, n_points_global=sum(lengths), train_particles=25)
for _ in range(3):
    step()
torch.cuda.synchronize()
host, n = [], 30
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter(); step(); host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print('RIDER=%s CAPTURE=%s: %.3f ms/step wall; host enqueue median %.3f ms, max %.3f ms; all enqueued after %.1f ms of %.1f'
      % (os.environ.get('MDMM_RIDER', '1'), os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'), 1e3 * t_all / n, 1e3 * host[n // 2], 1e3 * host[-1], 1e3 * t_enq, 1e3 * t_all))
