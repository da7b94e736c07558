"""GPU box, diagnostic build (-DWIDE_STAMPS, MDMM_LIB=.../ab_stamps/libmdmm_hip.so): where one step of
the wide forward sweep spends its cycles.  usage: python tools/wide_stamps.py [K=25] [P=4] [B=256] [T=40] [bf16=1]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
kw = dict(K=25, P=4, B=256, T=40, bf16=1, bwd=0)
for a in sys.argv[1:]:
    k, v = a.split('='); kw[k] = int(v)
dev = torch.device('cuda:0')
stamps = torch.zeros(8 * 32, dtype=torch.int64, device=dev)
os.environ['MDMM_STAMP_PTR'] = '%x' % stamps.data_ptr()
from mdmm import ops
K, P, B, T = kw['K'], kw['P'], kw['B'], kw['T']
D = H = 256
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device=dev)
shapes = [(H, D), (H,), (D, H), (D,), (D, D), (D,), (H, D), (H,), (D, H), (D,), (D, D), (D,)]
gtf = [0.06 * g(*s) for s in shapes]
z0m, z0s = g(D) * 0.1, g(D) * 0.1
experts = []
for m in range(P - 1):
    experts.append(ops.ExpertSpec(g(T, B, D), g(T, B, D).abs() + 0.3, (torch.rand(T, B, device=dev) > 0.1).float(),
                                  1 | (1 << (m + 1)), False))
cfg = ops.SweepCfg(T, B, D, H, P=P, K=K, reverse=False, sample=True, seed=7,
                   precision=torch.bfloat16 if kw['bf16'] else torch.float32)
if not kw['bwd']:
    with torch.no_grad():
        for _ in range(3):
            ops.bfvi_sweep(cfg, gtf, z0m, z0s, experts)
    torch.cuda.synchronize()
    s = stamps.cpu().view(8, 32)
    names = {0: 'step start', 1: 'P1 gemm', 2: 'P1 store', 3: 'barrier', 4: 'P2 gemm', 5: 'barrier', 6: 'P3 gemm+store',
             7: 'barrier', 8: 'P4 gemm+barrier+store', 9: 'P5a exp+gemm', 10: 'barrier', 16: 'P5b gemm', 11: 'softplus+poe',
             12: 'moments', 13: 'fuse+sample', 14: 'barrier', 15: 'Z store+barrier'}
    order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 16, 11, 12, 13, 14, 15]
else:
    for t in gtf + [z0m, z0s]:
        t.requires_grad_()
    for e in experts:
        e.mean.requires_grad_(); e.std.requires_grad_()
    for _ in range(2):
        outs = ops.bfvi_sweep(cfg, gtf, z0m, z0s, experts)
        sum((o * o).mean() for o in outs if o.numel()).backward()
    torch.cuda.synchronize()
    s = stamps.cpu().view(8, 32)
    names = {0: 'step start', 1: '(A) fuse adjoint', 2: 'R1 particles+store+spill', 3: 'barrier', 4: 'R2 2 gemms+stores+spills',
             5: 'barrier', 6: 'R3 3 gemms+store+spill', 7: 'barrier', 8: 'R4 gemm', 9: 'E elementwise', 10: 'E stores+spills',
             11: 'barrier', 12: 'D1 gemm+store+spill', 13: 'barrier', 14: 'D2 2 gemms+store+spills', 15: 'barrier+store+barrier',
             16: 'D3 3 gemms+adj', 17: 'barrier'}
    order = list(range(18))
    if K == 1:
        names.update({18: '  (A) arguments + pair table', 19: '  (A) requests issued'})
        order = [0, 18, 19] + list(range(1, 18))
    if os.environ.get('MDMM_FWD_PARK', '1') != '0' and 1 < K <= 25 and kw['bf16']:
        # sweep_wide_bwd4.hip (the forward kept its park)
        names = {0: 'step start', 1: 'pair 0: fuse adjoint, barrier, E', 2: 'pair 1: fuse adjoint, E', 3: 'pair 2', 4: 'pair 3',
                 5: 'mask loads, barrier', 6: 'D1 2 gemms + masks', 7: 'barrier, GN / GHG stores+spills, barrier',
                 8: 'D2 2 gemms + mask', 9: 'barrier, GHN store+spill, barrier', 10: 'D3 sums over the particles',
                 11: '  D1 gemm 1 (Ws^T G3)', 12: '  D1 gemm 2 (W2g^T GG)', 13: '  D3 noise loads issued', 14: '  D3 gemm 1 (W1n^T GHN)',
                 15: '  D3 gemm 2 (Wl^T Glin)'}
        order = [0, 1, 2, 3, 4, 5, 11, 12, 6, 7, 8, 9, 13, 14, 15, 10]
for w in (0, 7):
    print('wave %d (cycles since step start; delta)' % w)
    prev = int(s[w, 0])
    for k in order[1:]:
        v = int(s[w, k])
        print('  %-28s %8d  +%d' % (names[k], v - int(s[w, 0]), v - prev))
        prev = v
