"""GPU box: what bounds a csrc/gemm_heads.hip kernel -- the kernel without its stores / without its products
(MDMM_GEMM_MODE), next to plain fills and copies of the output's size."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops
dev = torch.device('cuda:0')


def ev(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in ((10240, 40960) if 'expand' in sys.argv else ()):
    N = 4096
    a = torch.randn(M, 256, device=dev).bfloat16()
    w = (torch.randn(N, 256, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    src = torch.randn(M, N, device=dev).bfloat16()
    print('M = %d: fill %.1f us, copy %.1f us' % (M, ev(lambda: out.zero_()), ev(lambda: out.copy_(src))))
    for mode, tag in ((0, 'whole kernel'), (1, 'no stores'), (2, 'quarter products'), (3, 'no stage loads'), (0, 'whole kernel')):
        os.environ['MDMM_GEMM_MODE'] = str(mode)
        print('   expand %-18s %6.1f us (back to back)' % (tag, ev(lambda: ops._gemm_bf16(a, False, w, False, M, N, 256, None, out_dtype=torch.bfloat16))))
    os.environ['MDMM_GEMM_MODE'] = '0'


for M in (10240, 40960):
    N, K = 256, 4096
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    print('M = %d contract:' % M)
    for mode, tag in ((0, 'whole kernel'), (1, 'no A loads'), (2, 'no B loads'), (3, 'quarter products'), (0, 'whole kernel')):
        os.environ['MDMM_GEMM_MODE'] = str(mode)
        ops.TIMER = ops.KernelTimer()
        tt = ev(lambda: ops._gemm_bf16(a, False, w, False, M, N, K, None))
        ops.TIMER = None
        print('   contract %-18s %6.1f us (back to back, with the fold)' % (tag, tt))
    os.environ['MDMM_GEMM_MODE'] = '0'
