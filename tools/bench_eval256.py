"""GPU box: evaluation forward (trainer.py:264-323: MAP smoother over a K-particle filter) at the Weizmann
latent size z = h = 256 (GaussianMLP-free: the sweeps only, via ops.bfvi_sweep), T = 40, B sequences."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops
dev = torch.device('cuda:0')
B, T, D = int(os.environ.get('B', 256)), 40, 256
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device=dev)
shapes = [(D, D), (D,), (D, D), (D,), (D, D), (D,), (D, D), (D,), (D, D), (D,), (D, D), (D,)]
gtf = [0.06 * g(*s) for s in shapes]
z0m, z0s = g(D) * 0.1, g(D) * 0.1
experts = [ops.ExpertSpec(g(T, B, D), g(T, B, D).abs() + 0.3, (torch.rand(T, B, device=dev) > 0.1).float(), 1, False)
           for _ in range(3)]
for K in (1, 25, 100, 128, 200):
    for prec in (torch.bfloat16,):
        cfg = ops.SweepCfg(T, B, D, D, P=1, K=K, reverse=True, sample=K > 1, seed=7, precision=prec, need_samples=False)
        with torch.no_grad():
            ops.bfvi_sweep(cfg, gtf, z0m, z0s, experts); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ops.bfvi_sweep(cfg, gtf, z0m, z0s, experts)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print('filter sweep z=256 B=%d K=%d: %.2f ms (wide family: %s)' % (B, K, dt * 1e3, ops.wide_shape(cfg)), flush=True)
