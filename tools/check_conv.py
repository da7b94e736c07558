"""GPU box: csrc/conv_tiles.hip against torch's convolutions on the same bf16-rounded operands
(products of bf16 numbers are exact in fp32, so only the summation order differs), forward, input
gradient and weight gradient of every layer shape of the image pyramids; then timings against the
library at the cfg3 frame count.  usage: python tools/check_conv.py [time=1] [N=10240]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch, torch.nn as nn
from mdmm import ops

kw = dict(time=1, N=10240, act=0)
for a in sys.argv[1:]:
    k, v = a.split('='); kw[k] = int(v)
dev = torch.device('cuda:0')
rb = lambda t: t.to(torch.bfloat16).to(torch.float32)      # noqa: E731


def err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


layers = [('Deconv 64->32  8->16', lambda: nn.ConvTranspose2d(64, 32, 4, 2, 1), (64, 8, 8)),
          ('Deconv 32->16 16->32', lambda: nn.ConvTranspose2d(32, 16, 4, 2, 1), (32, 16, 16)),
          ('Deconv 16->3  32->64', lambda: nn.ConvTranspose2d(16, 3, 4, 2, 1), (16, 32, 32)),
          ('Deconv 16->1  32->64', lambda: nn.ConvTranspose2d(16, 1, 4, 2, 1), (16, 32, 32)),
          ('Conv    3->16 64->32', lambda: nn.Conv2d(3, 16, 3, 2, 1), (3, 64, 64)),
          ('Conv    1->16 64->32', lambda: nn.Conv2d(1, 16, 3, 2, 1), (1, 64, 64)),
          ('Conv   16->32 32->16', lambda: nn.Conv2d(16, 32, 3, 2, 1), (16, 32, 32)),
          ('Conv   32->64 16->8 ', lambda: nn.Conv2d(32, 64, 3, 2, 1), (32, 16, 16))]
torch.manual_seed(0)
for name, mk, shp in layers:
    for N in (1, 5, 300):
        layer = mk().to(dev)
        x = torch.randn(N, *shp, device=dev, requires_grad=True)
        with ops.conv_operands(torch.bfloat16):
            assert ops.conv_tiles_supported(layer, x), name
            y = ops.conv_tiles(layer, x)
        gy = torch.randn_like(y)
        gx, gw, gb = torch.autograd.grad(y, [x, layer.weight, layer.bias], gy)
        # reference: the same roundings, fp32 arithmetic
        xr = rb(x.detach()).requires_grad_()
        wr = rb(layer.weight.detach()).requires_grad_()
        fn = torch.nn.functional.conv_transpose2d if isinstance(layer, nn.ConvTranspose2d) else torch.nn.functional.conv2d
        yr = fn(xr, wr, layer.bias.detach(), 2, 1)
        # input gradient contracts rounded gy with rounded w; weight gradient rounded gy with rounded x
        gxr, = torch.autograd.grad(fn(xr, wr, None, 2, 1), xr, rb(gy), retain_graph=False)
        gwr, = torch.autograd.grad(fn(xr, wr, None, 2, 1), wr, rb(gy))
        print('%-22s N=%-4d fwd %.1e  dgrad %.1e  wgrad %.1e  bias %.1e' % (
            name, N, err(y, yr), err(gx, gxr), err(gw, gwr), err(gb, gy.sum((0, 2, 3)))), flush=True)

if kw['time']:
    N = kw['N']
    act = torch.bfloat16 if kw['act'] else torch.float32
    for name, mk, shp in layers:
        layer = mk().to(dev)
        x = torch.randn(N, *shp, device=dev, requires_grad=True)
        xo = x.detach().to(act).requires_grad_() if (kw['act'] and shp[0] > 4) else x
        res = []
        for own in (False, True):
            def run():
                if own:
                    with ops.conv_operands(torch.bfloat16, act=act):
                        y = ops.conv_tiles(layer, xo)
                else:
                    y = layer(x)
                return y
            y = run(); gy = torch.randn_like(y)
            t = []
            for what in ('fwd', 'bwd'):
                for _ in range(2):
                    y = run()
                    if what == 'bwd':
                        torch.autograd.grad(y, [xo if own else x, layer.weight], gy.to(y.dtype))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    y = run()
                    if what == 'bwd':
                        torch.autograd.grad(y, [xo if own else x, layer.weight], gy.to(y.dtype))
                e1.record(); torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1) / 5)
            res.append((t[0], t[1] - t[0]))
        gb_f = (xo.numel() * xo.element_size() + y.numel() * y.element_size()) / 1e9
        print('%-22s N=%d  library fwd %.3f bwd %.3f ms | own fwd %.3f bwd %.3f ms | fwd traffic %.2f GB = %.3f ms at 4 TB/s'
              % (name, N, res[0][0], res[0][1], res[1][0], res[1][1], gb_f, gb_f / 4.0), flush=True)

# ---- the bf16-operand GEMM (csrc/gemm_tiles.hip) against torch on the rounded operands
print('linear tiles:')
for (m, k, n) in [(512, 32, 32), (1000, 36, 40), (2048, 256, 4096), (10240, 4096, 256), (4096, 9216, 256), (640, 260, 132)]:
    x = torch.randn(m, k, device=dev, requires_grad=True)
    lin = nn.Linear(k, n).to(dev)
    assert ops.linear_tiles_supported(x, lin.weight), (m, k, n)
    y = ops.linear_tiles(x, lin.weight, lin.bias)
    gy = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, [x, lin.weight, lin.bias], gy)
    xr, wr = rb(x.detach()).requires_grad_(), rb(lin.weight.detach()).requires_grad_()
    yr = torch.nn.functional.linear(xr, wr, lin.bias.detach())
    gxr = rb(gy) @ wr.detach()
    gwr = rb(gy).t() @ xr.detach()
    print('  M=%-6d K=%-5d N=%-5d fwd %.1e dgrad %.1e wgrad %.1e' % (m, k, n, err(y, yr), err(gx, gxr), err(gw, gwr)), flush=True)
    if kw['time'] and m >= 2048:
        for own in (False, True):
            f = (lambda: ops.linear_tiles(x, lin.weight, lin.bias)) if own else (lambda: lin(x))
            t = []
            for what in ('fwd', 'bwd'):
                for _ in range(2):
                    y = f()
                    if what == 'bwd':
                        torch.autograd.grad(y, [x, lin.weight], gy)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    y = f()
                    if what == 'bwd':
                        torch.autograd.grad(y, [x, lin.weight], gy)
                e1.record(); torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1) / 5)
            print('      %s fwd %.3f ms  bwd %.3f ms' % ('own    ' if own else 'library', t[0], t[1] - t[0]), flush=True)
