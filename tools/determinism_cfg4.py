"""GPU box: is the eager cfg4 step (MultiDKS, modality chains on their own streams) the same function of its inputs every
time?  Three runs on the same weights, batch and Philox stream; every gradient against the first run's.
usage: [MDMM_DKS_STREAMS=0] python tools/determinism_cfg4.py [B=256]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import torch
import bench
from oracle import mdmm_oracle as orc
from mdmm import models
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = bench.CONFIGS['cfg4']
lengths = sorted([40] * (B - B // 5) + [int(n) for n in np.random.RandomState(3).randint(5, 40, B // 5)], reverse=True)
inputs, targets, mask, _ = cfg.batch(cfg.T, B, 77, 'cpu')
for d in (inputs, targets):
    for k in d:
        for b, n in enumerate(lengths):
            d[k][n:, b] = float('nan')
mask = orc.len_to_mask(lengths)
to = lambda d: {k: v.to(dev) for k, v in d.items()}
x, tg, mask = to(inputs), to(targets), mask.to(dev)
torch.manual_seed(0)
model = cfg.model(models, dev)
runs = []
from mdmm import ops as _ops
_orig_stash, _seen = _ops._lazy_stash, []
def _spy(dx, entry):
    _seen[-1].append((tuple(entry['x'].shape), entry['means'].detach().clone(), entry['part'].detach().clone(), entry['dyn'].detach().clone()))
    return _orig_stash(dx, entry)
if os.environ.get('SPY') == '1':
    _ops._lazy_stash = _spy
_spies = []
for r in range(3):
    _seen.append([])
    if os.environ.get('SPY') == '2':
        _ops.DEBUG_SPY = []
        _spies.append(_ops.DEBUG_SPY)
    model.noise = PhiloxNoise(seed=4321)
    for p in model.parameters():
        p.grad = None
    junk = [torch.randn(1 << 22, device=dev) for _ in range(r)]      # (another allocator state every run)
    loss = model.step(x, mask, 1.0, cfg.rec, targets=tg, lengths=lengths)
    (loss / sum(lengths)).backward()
    torch.cuda.synchronize()
    del junk
    runs.append((float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
print('streams', os.environ.get('MDMM_DKS_STREAMS', '1'), 'B', B, 'losses', [r[0] for r in runs])
for r in (1, 2):
    worst = sorted(((float((runs[r][1][k] - runs[0][1][k]).norm() / (runs[0][1][k].norm() + 1e-30)), k) for k in runs[0][1]), reverse=True)[:4]
    print(' run %d vs run 0: worst gradients' % r, [(('%.2e' % e), k) for e, k in worst])
if os.environ.get('SHOW'):
    k = os.environ['SHOW']
    a, b = runs[0][1][k].flatten().double(), runs[1][1][k].flatten().double()
    d = (b - a)
    print(k, 'shape', tuple(runs[0][1][k].shape), 'norm', float(a.norm()), 'diff norm', float(d.norm()), 'nonzero diffs', int((d != 0).sum()), 'of', d.numel())
    idx = d.abs().argsort(descending=True)[:12]
    print(' largest diffs (index, a, b - a):', [(int(i), '%.5e' % float(a[i]), '%.3e' % float(d[i])) for i in idx])
    print(' ratio b/a of the largest elements:', [('%.8f' % float(b[i] / a[i])) for i in a.abs().argsort(descending=True)[:8]])

if os.environ.get('SPY') == '1':
    torch.cuda.synchronize()
    for i, (e0, e1) in enumerate(zip(_seen[0], _seen[1])):
        print(' stash %d x%s: means equal %s (max |d| %.3e), partial sums equal %s, dyn equal %s' % (
            i, e0[0], torch.equal(e0[1], e1[1]), float((e0[1] - e1[1]).abs().max()), torch.equal(e0[2], e1[2]), torch.equal(e0[3], e1[3])))

if os.environ.get('SPY') == '2':
    torch.cuda.synchronize()
    for i, (e0, e1) in enumerate(zip(_spies[0], _spies[1])):
        if e0[0] > 100:
            if e0[0] - 100 <= 4:
                names = ('big', 'x', 'dyn', 'means', 'stats', 'gamma')
                print(' BEFORE lazy wgrad %d cb=%d:' % (i, e0[0] - 100), ', '.join('%s %s' % (n, 'same' if torch.equal(u, v) else 'DIFFERENT (%d of %d)' % (int((u != v).sum()), u.numel()))
                                                                           for n, u, v in zip(names, e0[1:], e1[1:])))
            continue
        if e0[0] > 4:
            continue
        names = ('gw', 'big', 'x', 'dyn', 'means', 'ws')
        print(' lazy wgrad %d cb=%d:' % (i, e0[0]), ', '.join('%s %s' % (n, 'same' if torch.equal(u, v) else 'DIFFERENT (%d of %d)' % (int((u != v).sum()), u.numel()))
                                                          for n, u, v in zip(names, e0[1:], e1[1:])))
