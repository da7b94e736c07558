"""GPU box: does a decoder call get cheaper per frame when two passes share it?  Forward + backward of the stock
ImageDecoder (own kernels, bf16 operands and activations) on N and on 2 N frames, device time by HIP events per
library call.  usage: python tools/time_decoder_batch.py [N=10240]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops
from mdmm.models import common as C
N = int(sys.argv[1].split('=')[1]) if len(sys.argv) > 1 else 10240
dev = torch.device('cuda:0')
dec = C.ImageDecoder(256, n_channels=3).to(dev).train()
for n in (N, 2 * N, N):
    z = torch.randn(n, 256, device=dev, requires_grad=True)
    def run():
        with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
            out = dec(z, logits=True)
        out = out[0] if isinstance(out, tuple) else out
        g = torch.ones_like(out)
        torch.autograd.grad(out, [z] + list(dec.parameters()), g, allow_unused=True)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    ops.TIMER = ops.KernelTimer()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    t, ops.TIMER = ops.TIMER.summary(), None
    tot = sum(v[1] for v in t.values()) / 5
    print('N = %6d: own calls %.3f ms per fwd+bwd = %.3f us per frame; wall %.3f ms' % (n, tot, tot / n * 1e3, e0.elapsed_time(e1) / 5))
    if n == 2 * N:
        for k, v in sorted(t.items(), key=lambda kv: -kv[1][1])[:12]:
            print('      %-28s %.3f ms / call' % (k, v[1] / v[0]))
