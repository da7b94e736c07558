"""GPU box: the plug-in Linear heads' three products on csrc/gemm_tiles.hip against the library's bf16
GEMM on the same shapes (headroom of the own kernel).  usage: python tools/gemm_headroom.py"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops

dev = torch.device('cuda:0')
M, K, N = 10240, 4096, 256


def t(fn, n=20):
    """device time per call: HIP events around every launch (own kernels: ops.KernelTimer; library calls:
    events around the call), so that a host-bound loop does not hide the kernel"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ops.TIMER = ops.KernelTimer()
    spans = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        spans.append((e0, e1))
    torch.cuda.synchronize()
    own, ops.TIMER = ops.TIMER.summary(), None
    if own:
        return sum(v[1] for v in own.values()) / n * 1e3
    return sum(a.elapsed_time(b) for a, b in spans) / n * 1e3


x = torch.randn(M, K, device=dev).bfloat16()
w = torch.randn(N, K, device=dev) * 0.02
g = torch.randn(M, N, device=dev)
wb, gb = w.bfloat16(), g.bfloat16()
def ab(tag, fn):
    for raw in (1, 0, 1, 0):
        os.environ['MDMM_GEMM_NO_RAW'] = '0' if raw else '1'
        print('  own      %7.1f us (%s, bf16 operands %s)' % (t(fn), tag, 'moved raw' if raw else 'converted'))
    os.environ['MDMM_GEMM_NO_RAW'] = '0'


print('enc head  y = x W^T   (%d x %d -> %d)' % (M, K, N))
ab('fp32 W', lambda: ops._gemm_bf16(ops._rows(x), False, ops._rows(w), False, M, N, K))
ab('bf16 W', lambda: ops._gemm_bf16(ops._rows(x), False, wb, False, M, N, K))
print('  own      %7.1f us (fp32 W)' % t(lambda: ops._gemm_bf16(ops._rows(x), False, ops._rows(w), False, M, N, K)))
print('  own      %7.1f us (bf16 W)' % t(lambda: ops._gemm_bf16(ops._rows(x), False, wb, False, M, N, K)))
print('  library  %7.1f us (bf16 in, bf16 out)' % t(lambda: torch.nn.functional.linear(x, wb)))
print('dgrad     dx = g W')
print('  own      %7.1f us (fp32 W)' % t(lambda: ops._gemm_bf16(ops._rows(g), False, ops._rows(w), True, M, K, N, out_dtype=torch.bfloat16)))
print('  own      %7.1f us (bf16 W, bf16 g)' % t(lambda: ops._gemm_bf16(gb, False, wb, True, M, K, N, out_dtype=torch.bfloat16)))
print('  library  %7.1f us' % t(lambda: gb @ wb))
print('wgrad     dW = g^T x')
print('  own      %7.1f us (fp32 g)' % t(lambda: ops._gemm_bf16(ops._rows(g), True, ops._rows(x), True, N, K, M)))
print('  own      %7.1f us (bf16 g)' % t(lambda: ops._gemm_bf16(gb, True, ops._rows(x), True, N, K, M)))
print('  library  %7.1f us' % t(lambda: gb.t() @ x))
z = torch.randn(M, N, device=dev)
w2 = torch.randn(K, N, device=dev) * 0.02
print('dec head  y = z W^T   (%d x %d -> %d)' % (M, N, K))
print('  own      %7.1f us (fp32 W)' % t(lambda: ops._gemm_bf16(ops._rows(z), False, ops._rows(w2), False, M, K, N, out_dtype=torch.bfloat16)))
w2b = w2.bfloat16()
print('  own      %7.1f us (bf16 W)' % t(lambda: ops._gemm_bf16(ops._rows(z), False, w2b, False, M, K, N, out_dtype=torch.bfloat16)))
print('  library  %7.1f us' % t(lambda: torch.nn.functional.linear(z.bfloat16(), w2.bfloat16())))
