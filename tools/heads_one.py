"""GPU box: one shape of a csrc/gemm_heads.hip kernel, a few calls (for rocprofv3 --pmc runs).
usage: python3 tools/heads_one.py contract|expand|wgrad [M]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops
dev = torch.device('cuda:0')
kind = sys.argv[1]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 10240
if kind == 'contract':
    a = torch.randn(M, 4096, device=dev).bfloat16()
    w = (torch.randn(256, 4096, device=dev) * 0.02).bfloat16()
    fn = lambda: ops._gemm_bf16(a, False, w, False, M, 256, 4096, None)
elif kind == 'expand':
    a = torch.randn(M, 256, device=dev).bfloat16()
    w = (torch.randn(4096, 256, device=dev) * 0.05).bfloat16()
    fn = lambda: ops._gemm_bf16(a, False, w, False, M, 4096, 256, None, out_dtype=torch.bfloat16)
else:
    g = torch.randn(M, 256, device=dev).bfloat16()
    x = torch.randn(M, 4096, device=dev).bfloat16()
    fn = lambda: ops._gemm_bf16(g, True, x, True, 256, 4096, M)
for _ in range(5):
    fn()
torch.cuda.synchronize()
