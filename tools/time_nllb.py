"""GPU box: the stacked-passes Bernoulli loss at the cfg5 per-GPU size (T*B = 65,536 rows, 2 passes, half the rows masked out
by NaN observations): video-shaped bf16 logits (rows of 12,288 = whole float4s) and audio-shaped fp32 logits (rows of 12,810:
the one-element path), forward / backward ms and the bytes they imply."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops
dev = torch.device('cuda:0')
T, B, P = 128, 512, 2
def run(name, shape, dt, fast=False):
    inner = 1
    for s in shape: inner *= s
    x = torch.rand(T, B, *shape, device=dev)
    x[torch.rand(T, B, device=dev) < 0.5] = float('nan')
    mask = torch.ones(T, B, dtype=torch.bool, device=dev)
    lg = (torch.randn(P * T * B, *shape, device=dev)).to(dt).requires_grad_()
    def once():
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(); loss = ops.nll_bernoulli_logits(lg, x, mask, 2, 1.0, None, passes=P, fast=fast); b.record()
        g, = torch.autograd.grad(loss, lg); c.record(); torch.cuda.synchronize()
        return a.elapsed_time(b), b.elapsed_time(c)
    ts = [once() for _ in range(4)][1:]
    f, bw = min(t[0] for t in ts), min(t[1] for t in ts)
    eb = lg.element_size()
    live = float((~torch.isnan(x).flatten(2).any(-1)).float().mean())
    fb = T * B * inner * (P * eb + 4) / 1e9
    bb = T * B * inner * (2 * P * eb + 4) / 1e9
    print('%-28s fwd %.3f ms (%.1f GB if every row were read: %.2f TB/s)  bwd %.3f ms (%.1f GB: %.2f TB/s)  live rows %.2f'
          % (name, f, fb, fb / f, bw, bb, bb / bw, live))
run('video bf16 (3,64,64)', (3, 64, 64), torch.bfloat16)
run('audio fp32 (10,1281)', (10, 1281), torch.float32)
run('audio fp32 (10,1281), fast', (10, 1281), torch.float32, True)
run('audio-like fp32 (10,1280)', (10, 1280), torch.float32)
