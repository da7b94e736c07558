"""GPU box: loss trajectory of cfg3 (B sequences) over N Adam steps with the plug-in precision switches at
fp32 and at bf16 (same seeds, same Philox noise): does bf16 operand / activation storage move training?
usage: python tools/train_traj.py [B=64] [N=60]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
import bench
from mdmm import models
from mdmm.harness import GradBucket, elbo_step
from mdmm.noise import PhiloxNoise
kw = dict(B=64, N=60)
for a in sys.argv[1:]:
    k, v = a.split('='); kw[k] = int(v)
dev = torch.device('cuda:0')
cfg = bench.Cfg3
curves = {}
for name, (sw, cv, ac) in (('all fp32 (sweeps fp32 operands)', (torch.float32, torch.float32, torch.float32)),
                           ('bf16 sweeps only', (torch.bfloat16, torch.float32, torch.float32)),
                           ('bf16 sweeps + conv operands', (torch.bfloat16, torch.bfloat16, torch.float32)),
                           ('bf16 sweeps + conv operands + activations', (torch.bfloat16, torch.bfloat16, torch.bfloat16))):
    torch.manual_seed(0)
    m = cfg.model(models, dev)
    m.sweep_dtype, m.conv_dtype, m.act_dtype = sw, cv, ac
    m.noise = PhiloxNoise(seed=1000)
    opt = torch.optim.Adam(m.parameters(), lr=cfg.lr, fused=True)
    bucket = GradBucket(m.parameters())
    inputs, targets, mask, lengths = cfg.batch(cfg.T, kw['B'], 1234, dev)
    losses = []
    for i in range(kw['N']):
        loss = elbo_step(m, opt, bucket, inputs, mask, lengths, 1.0, cfg.rec, targets=targets,
                         n_points_global=sum(lengths), train_particles=25)
        losses.append(float(loss) / sum(lengths))
    curves[name] = losses
    print('%-44s loss/point at steps 0, 10, 20, 40, %d: %s' % (name, kw['N'] - 1, ' '.join('%.2f' % losses[i] for i in (0, 10, 20, 40, kw['N'] - 1))), flush=True)
ref = curves['all fp32 (sweeps fp32 operands)']
for name, c in curves.items():
    dev_ = [abs(a - b) / abs(b) for a, b in zip(c, ref)]
    print('%-44s relative deviation from the fp32 trajectory: median %.1e, max %.1e (step %d)'
          % (name, sorted(dev_)[len(dev_) // 2], max(dev_), dev_.index(max(dev_))))
