"""GPU box: the shape-specialised GEMMs of csrc/gemm_heads.hip against torch on the same bf16 operands, and their
device time (HIP events around each call) next to the generic tile kernel's (MDMM_GEMM_GENERIC=1) and the
library's.  usage: python tools/check_heads.py [expand|contract|wgrad ...]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops

dev = torch.device('cuda:0')
which = sys.argv[1:] or ['expand', 'contract', 'wgrad']


def t(fn, n=20):
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


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def both(tag, fn, ref, lib):
    os.environ['MDMM_GEMM_GENERIC'] = '0'
    out = fn()
    torch.cuda.synchronize()
    e = rel(out, ref)
    os.environ['MDMM_GEMM_GENERIC'] = '1'
    e_g = rel(fn(), ref)
    t_g = t(fn)
    os.environ['MDMM_GEMM_GENERIC'] = '0'
    t_h = t(fn)
    print('%-46s err %.2e (generic %.2e)   heads %6.1f us   generic %6.1f us   library %6.1f us' %
          (tag, e, e_g, t_h, t_g, t(lib)), flush=True)
    return e


torch.manual_seed(0)
worst = 0.0
if 'expand' in which:
    for M, N in ((10240, 4096), (10240, 512), (1000, 4096), (37, 256), (6 * 40 + 3, 4096), (40960, 4096)):
        a = torch.randn(M, 256, device=dev).bfloat16()
        w = (torch.randn(N, 256, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        ref = (a.float() @ w.float().t() + bias)
        worst = max(worst, both('expand  %6d x 256 -> %4d (bf16 out)' % (M, N),
                                lambda: ops._gemm_bf16(a, False, w, False, M, N, 256, bias, out_dtype=torch.bfloat16),
                                ref.bfloat16(), lambda: torch.nn.functional.linear(a, w, bias.bfloat16())))
if 'contract' in which:
    for M, N, K, odt in ((10240, 256, 4096, torch.float32), (10240, 512, 4096, torch.float32), (10240, 256, 4096, torch.bfloat16),
                         (1000, 256, 4096, torch.float32), (243, 256, 512, torch.float32), (40960, 256, 4096, torch.float32),
                         (4096, 256, 9216, torch.float32)):
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        bias = torch.randn(N, device=dev)
        ref = (a.float() @ w.float().t() + bias).to(odt)
        worst = max(worst, both('contract %6d x %4d -> %3d (%s out)' % (M, K, N, 'bf16' if odt is torch.bfloat16 else 'fp32'),
                                lambda: ops._gemm_bf16(a, False, w, False, M, N, K, bias, out_dtype=odt),
                                ref, lambda: torch.nn.functional.linear(a, w, bias.bfloat16())))
if 'wgrad' in which:
    for M, I, J in ((10240, 256, 4096), (10240, 4096, 256), (1000, 256, 4096), (10240, 256, 512), (4099, 4096, 256),
                    (40960, 256, 4096), (10240, 512, 256)):
        g_ = torch.randn(M, I, device=dev).bfloat16()
        x = torch.randn(M, J, device=dev).bfloat16()
        ref = g_.float().t() @ x.float()
        worst = max(worst, both('wgrad   %5d x %4d over %6d rows' % (I, J, M),
                                lambda: ops._gemm_bf16(g_, True, x, True, I, J, M), ref, lambda: g_.t() @ x))
print('worst relative error %.3e' % worst)
