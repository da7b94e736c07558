"""MultiVRNN.forward + backward on Spirals-shaped batches: the scan kernels (csrc/vrnn.hip) against the
model's own step-by-step route (stock modules per step, product of experts on its kernel).

  python tools/bench_vrnn.py [--B 1024] [--T 100] [--h 16] [--z 16] [--steps 5]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'multimodal-dmm_amd'))
from mdmm import models, ops   # noqa: E402
from mdmm.noise import PhiloxNoise   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=1024)
    ap.add_argument('--T', type=int, default=100)
    ap.add_argument('--h', type=int, default=16)
    ap.add_argument('--z', type=int, default=16)
    ap.add_argument('--layers', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--mode', default='use_inputs')
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    names, dims = ['spiral-x', 'spiral-y'], [1, 1]
    m = models.MultiVRNN(names, dims, h_dim=args.h, z_dim=args.z, n_layers=args.layers, recur_mode=args.mode,
                         device=dev)
    m.noise = PhiloxNoise(seed=1)
    x = {k: torch.randn(args.T, args.B, 1, device=dev) for k in names}
    for k in names:
        x[k][torch.rand(args.T, args.B, device=dev) < 0.1] = float('nan')
    lengths = [args.T] * args.B

    def step(scan):
        m.zero_grad(set_to_none=True)
        infer, prior, recon = m(x, lengths=lengths, scan=scan)
        loss = m.kld_loss(infer, prior) + sum(recon[0][k].square().mean() + recon[1][k].mean() for k in names)
        loss.backward()
        return loss

    for scan, tag in ((True, 'scan kernels'), (False, 'step by step')):
        if args.only and args.only != ('scan' if scan else 'steps'):
            continue
        for _ in range(2):
            step(scan)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(scan)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print('%-14s %9.2f ms per forward + backward   %10.0f sequences/s' % (tag, dt * 1e3, args.B / dt))
        if scan:
            ops.TIMER = ops.KernelTimer()
            step(True)
            torch.cuda.synchronize()
            for k, v in sorted(ops.TIMER.summary().items(), key=lambda kv: -kv[1][1])[:6]:
                print('    %-28s %8.3f ms in %d launches' % (k, v[1], v[0]))
            ops.TIMER = None


if __name__ == '__main__':
    main()
