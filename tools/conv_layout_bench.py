"""GPU box: the six conv layers of the Weizmann plug-in stacks, forward + backward through MIOpen,
NCHW vs channels_last, fp32 (and bf16 with AMP=1).  usage: python tools/conv_layout_bench.py [N=2560]"""
import os, sys, time
import torch, torch.nn as nn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
dev = torch.device('cuda:0')
dt = torch.bfloat16 if os.environ.get('AMP') == '1' else torch.float32
layers = [('E1 conv 3->16 64->32', nn.Conv2d(3, 16, 3, 2, 1), (3, 64, 64)),
          ('E2 conv 16->32 32->16', nn.Conv2d(16, 32, 3, 2, 1), (16, 32, 32)),
          ('E3 conv 32->64 16->8', nn.Conv2d(32, 64, 3, 2, 1), (32, 16, 16)),
          ('D1 deconv 64->32 8->16', nn.ConvTranspose2d(64, 32, 4, 2, 1), (64, 8, 8)),
          ('D2 deconv 32->16 16->32', nn.ConvTranspose2d(32, 16, 4, 2, 1), (32, 16, 16)),
          ('D3 deconv 16->3 32->64', nn.ConvTranspose2d(16, 3, 4, 2, 1), (16, 32, 32))]
for name, layer, shp in layers:
    for fmt in (torch.contiguous_format, torch.channels_last):
        l = layer.to(dev).to(dt).to(memory_format=fmt)
        x = torch.randn(N, *shp, device=dev, dtype=dt).to(memory_format=fmt).requires_grad_()
        def run():
            y = l(x)
            y.backward(torch.ones_like(y))
        try:
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            y = l(x)
            gb = (x.numel() + y.numel()) * x.element_size() * 3 / 1e9
            print('%-26s %-14s fwd+bwd %.3f ms  (N=%d; x3 activation traffic %.2f GB -> %.2f ms at 4 TB/s)'
                  % (name, 'NCHW' if fmt is torch.contiguous_format else 'channels_last', ms, N, gb, gb / 4.0), flush=True)
        except Exception as e:
            print(name, fmt, 'failed', repr(e)[:200], flush=True)
