"""GPU box: the audio plug-ins' 1-D convolution kernels (csrc/conv1d.hip) alone, at the three layer shapes of the stock pyramids
and N frames: ms and the bytes they have to move against the time (forward = up or down, input gradient, weight gradient).
usage: python tools/time_conv1d.py [N=16384]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
import torch.nn as nn
from mdmm import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device('cuda:0')


def best(fn, rep=5):
    ts = []
    for _ in range(rep):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


for kind, c_in, c_out, length in (('conv', 10, 4, 1281), ('conv', 4, 8, 641), ('conv', 8, 16, 321),
                                  ('deconv', 16, 8, 161), ('deconv', 8, 4, 321), ('deconv', 4, 10, 641)):
    tr = kind == 'deconv'
    layer = (nn.ConvTranspose1d(c_in, c_out, 3, 2, 1) if tr else nn.Conv1d(c_in, c_out, 3, 2, 1)).to(dev)
    x = torch.randn(N, c_in, length, device=dev, requires_grad=True)
    y = ops.conv1d_tiles(layer, x)
    gy = torch.randn_like(y)
    nb = (x.numel() + y.numel()) * 4
    t_f = best(lambda: ops.conv1d_tiles(layer, x.detach()))
    t_dx = best(lambda: torch.autograd.grad(ops.conv1d_tiles(layer, x), x, gy))
    t_dw = best(lambda: torch.autograd.grad(ops.conv1d_tiles(layer, x.detach()), layer.weight, gy))
    print('%-6s %2d -> %2d, L = %4d: forward %.3f ms (%.2f TB/s)  forward + input gradient %.3f ms  forward + weight gradient %.3f ms'
          '   [in + out = %.2f GB]' % (kind, c_in, c_out, length, t_f, nb / t_f / 1e9, t_dx, t_dw, nb / 1e9))
