"""GPU box: run one eager ELBO step with every torch.empty() on the GPU pre-filled with NaN (float types): any
output or workspace element a kernel reads without having written it shows up as a non-finite loss / gradient.
(Graph replay recycles allocator blocks; eager steps on a fresh process mostly see zeros and hide such reads.)
usage: [PARTICLES=100] python tools/poison_check.py cfg3|cfg4|cfg5|cfg2 [lengths comma-separated]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bench
from oracle import mdmm_oracle as orc
from mdmm import models
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
cfg = bench.CONFIGS[name]
default = '128,97,64' if name == 'cfg5' else ('100,100,60,33' if name == 'cfg2' else '40,40,40,31,17,6')
lengths = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else default).split(',')]
if name == 'cfg5':
    inputs, targets, mask, _ = cfg.batch(cfg.T, len(lengths), 77, 'cpu', lengths=lengths)
else:
    inputs, targets, mask, _ = cfg.batch(cfg.T, len(lengths), 77, 'cpu')
    for d in (inputs, targets):
        for k in d:
            for b, n in enumerate(lengths):
                d[k][n:, b] = float('nan')
    mask = orc.len_to_mask(lengths)
to = lambda d: {k: v.to(dev) for k, v in d.items()}
x, tg, mask = to(inputs), to(targets), mask.to(dev)
torch.manual_seed(0)
model = cfg.model(models, dev)
model.noise = PhiloxNoise(seed=4321)

_empty, _empty_like = torch.empty, torch.empty_like
def poisoned(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.is_floating_point() and t.numel():
        t.fill_(float('nan'))
    elif t.is_cuda and t.dtype == torch.uint8 and t.numel():
        t.fill_(0xFF)                     # bf16 / fp32 NaN patterns for raw byte workspaces
    return t
def poisoned_like(t0, **k):
    t = _empty_like(t0, **k)
    if t.is_cuda and t.is_floating_point() and t.numel():
        t.fill_(float('nan'))
    return t
torch.empty, torch.empty_like = poisoned, poisoned_like
kw = dict(targets=tg, train_particles=int(os.environ.get('PARTICLES', bench.TRAIN_PARTICLES)))        # PARTICLES=100: the quad geometry
loss = model.step(x, mask, 1.0, cfg.rec, lengths=lengths, **kw)
(loss / sum(lengths)).backward()
torch.cuda.synchronize()
torch.empty, torch.empty_like = _empty, _empty_like
bad = [(k, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for k, p in model.named_parameters()
       if p.grad is not None and not torch.isfinite(p.grad).all()]
print(name, 'lengths', lengths, 'loss', float(loss.detach()), 'non-finite gradients:', bad if bad else 'none', flush=True)
