"""GPU box: per-kernel time of the backward of every layer of the image pyramids (own kernels,
bf16-stored activations), by HIP events.  usage: python tools/time_conv_parts.py [N=10240]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch, torch.nn as nn
from mdmm import ops
N = int(sys.argv[1].split('=')[1]) if len(sys.argv) > 1 else 10240
dev = torch.device('cuda:0')
layers = [('Deconv 64->32  8->16', nn.ConvTranspose2d(64, 32, 4, 2, 1), (64, 8, 8)),
          ('Deconv 32->16 16->32', nn.ConvTranspose2d(32, 16, 4, 2, 1), (32, 16, 16)),
          ('Deconv 16->3  32->64', nn.ConvTranspose2d(16, 3, 4, 2, 1), (16, 32, 32)),
          ('Conv   16->32 32->16', nn.Conv2d(16, 32, 3, 2, 1), (16, 32, 32)),
          ('Conv   32->64 16->8 ', nn.Conv2d(32, 64, 3, 2, 1), (32, 16, 16))]
for name, layer, shp in layers:
    layer = layer.to(dev)
    x = torch.randn(N, *shp, device=dev).to(torch.bfloat16).requires_grad_()
    with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
        y = ops.conv_tiles(layer, x)
    gy = torch.randn_like(y)
    for _ in range(2):
        torch.autograd.grad(y, [x, layer.weight], gy, retain_graph=True)
    torch.cuda.synchronize()
    ops.TIMER = ops.KernelTimer()
    for _ in range(5):
        with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
            y = ops.conv_tiles(layer, x)
        torch.autograd.grad(y, [x, layer.weight], gy)
    torch.cuda.synchronize()
    t, ops.TIMER = ops.TIMER.summary(), None
    print(name, '  '.join('%s %.3f ms' % (k, v[1] / v[0]) for k, v in sorted(t.items())), flush=True)
