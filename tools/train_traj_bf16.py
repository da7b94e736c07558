"""GPU box: fifty Adam steps of the cfg3 model at B = 32, fp32 operands against bf16 operands (same weights, batch and
Philox streams): the two ELBO curves and their relative distance per step.  usage: python tools/train_traj_bf16.py [lr=1e-3]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch, numpy as np
import bench
from mdmm import models
from mdmm.noise import PhiloxNoise
from mdmm.harness import GradBucket, elbo_step
dev = torch.device('cuda:0')
def mk(dtype, state=None):
    torch.manual_seed(0)
    cfg = bench.Cfg3 if dtype is torch.bfloat16 else bench.Cfg3F32
    m = cfg.model(models, dev)
    if state is not None: m.load_state_dict(state)
    m.noise = PhiloxNoise(seed=2024)
    return m
state = {k: v.detach().clone() for k, v in mk(torch.float32).state_dict().items()}
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
curves = {}
for dtype in (torch.float32, torch.bfloat16):
    m = mk(dtype, state)
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    bucket = GradBucket(m.parameters())
    x, tg, mask, lengths = bench.Cfg3.batch(40, 32, 32, dev)
    ls = []
    for _ in range(50):
        ls.append(float(elbo_step(m, opt, bucket, x, mask, lengths, 1.0, bench.Cfg3.rec, targets=tg, n_points_global=sum(lengths), train_particles=25)))
    curves[dtype] = np.array(ls)
hi, lo = curves[torch.float32], curves[torch.bfloat16]
for i in range(50):
    print(i, '%.5e %.5e %.3e' % (hi[i], lo[i], abs(lo[i]-hi[i])/abs(hi[i])))
