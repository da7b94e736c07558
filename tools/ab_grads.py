"""GPU box: gradients of one eager cfg3 step with an environment switch off / on (same weights, same noise): the largest
relative difference per parameter.  usage: python tools/ab_grads.py VAR [B=32]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import bench
import torch
from mdmm import models
from mdmm.noise import PhiloxNoise
var = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device('cuda:0')
cfg = bench.CONFIGS['cfg3']
res = []
for flag in ('0', '1'):
    os.environ[var] = flag
    torch.manual_seed(0)
    model = cfg.model(models, dev)
    model.noise = PhiloxNoise(seed=5)
    inputs, targets, mask, lengths = cfg.batch(cfg.T, B, 1, dev)
    loss = model.step(inputs, mask, 1.0, cfg.rec, targets=targets, lengths=lengths, train_particles=bench.TRAIN_PARTICLES)
    loss.backward()
    torch.cuda.synchronize()
    res.append((float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
print('loss', res[0][0], res[1][0])
worst = []
for k, g0 in res[0][1].items():
    g1 = res[1][1][k]
    d = float((g0 - g1).abs().max() / (g0.abs().max() + 1e-30))
    worst.append((d, k))
for d, k in sorted(worst, reverse=True)[:12]:
    print('%.3e  %s' % (d, k))
