"""GPU box: evaluation forward (trainer.py:264-323): MAP smoother over a 200-particle filter, cfg2 sizes."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from bench import synth_batch
from mdmm import models, ops
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
B = int(os.environ.get('B', 1024))
m = models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32, device=dev).eval()
m.noise = PhiloxNoise(seed=1)
inputs, targets, mask, lengths = synth_batch(100, B, 1234, dev)
for K in (1, 25, 200):
    with torch.no_grad():
        m(inputs, lengths=lengths, sample=False, flt_particles=K); torch.cuda.synchronize()
        ops.TIMER = ops.KernelTimer()
        t0 = time.perf_counter()
        for _ in range(3):
            infer, prior, recon = m(inputs, lengths=lengths, sample=False, flt_particles=K)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        top = {k: round(v[1] / 3, 3) for k, v in sorted(ops.TIMER.summary().items(), key=lambda kv: -kv[1][1])[:3]}
        ops.TIMER = None
    print('eval forward B=%d flt_particles=%d: %.2f ms  (%.0f sequences/s)  %s' % (B, K, dt * 1e3, B / dt, top))
