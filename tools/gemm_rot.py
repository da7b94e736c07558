"""GPU box: staggered contraction start of csrc/gemm_tiles.hip (MDMM_GEMM_ROT) on the Linear heads' shapes."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops

dev = torch.device('cuda:0')
M, K, N = 10240, 4096, 256


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ops.TIMER = ops.KernelTimer()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    own, ops.TIMER = ops.TIMER.summary(), None
    return sum(v[1] for v in own.values()) / n * 1e3


x = torch.randn(M, K, device=dev).bfloat16()
w = torch.randn(N, K, device=dev) * 0.02
g = torch.randn(M, N, device=dev)
wb, gb = w.bfloat16(), g.bfloat16()
z = torch.randn(M, N, device=dev)
zb = z.bfloat16()
w2 = torch.randn(K, N, device=dev) * 0.02
w2b = w2.bfloat16()
gy = torch.randn(M, K, device=dev).bfloat16()
cases = [
    ('enc fwd   (M x 4096 -> 256), bf16 W', lambda: ops._gemm_bf16(ops._rows(x), False, wb, False, M, N, K)),
    ('enc dgrad (M x 256 -> 4096), fp32 W^T', lambda: ops._gemm_bf16(ops._rows(g), False, ops._rows(w), True, M, K, N, out_dtype=torch.bfloat16)),
    ('enc wgrad (256 x 4096 over M), fp32 g', lambda: ops._gemm_bf16(ops._rows(g), True, ops._rows(x), True, N, K, M)),
    ('dec fwd   (M x 256 -> 4096), bf16 W', lambda: ops._gemm_bf16(ops._rows(z), False, w2b, False, M, K, N, out_dtype=torch.bfloat16)),
    ('dec dgrad (M x 4096 -> 256), fp32 W^T', lambda: ops._gemm_bf16(gy, False, ops._rows(w2), True, M, N, K)),
    ('dec wgrad (4096 x 256 over M), bf16 g', lambda: ops._gemm_bf16(gy, True, ops._rows(z), True, K, N, M)),
]
for tag, fn in cases:
    r = []
    for rot in ('0', '1', '0', '1'):
        os.environ['MDMM_GEMM_ROT'] = rot
        r.append(t(fn))
    print('%-42s  plain %6.1f %6.1f us   staggered %6.1f %6.1f us' % (tag, r[0], r[2], r[1], r[3]))
