"""GPU box: one sweep (forward + backward) in isolation, single stream, HIP-event timing per launch.
usage: python tools/bench_sweep.py [K=25] [P=3] [B=1024] [T=100] [D=32] [H=32] [inv=0] [rev=1] [n=5] [bf16=0]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops
kw = dict(K=25, P=3, B=1024, T=100, D=32, H=32, inv=0, rev=1, n=5, bf16=0)
for a in sys.argv[1:]:
    k, v = a.split('='); kw[k] = int(v)
K, P, B, T, D, H = kw['K'], kw['P'], kw['B'], kw['T'], kw['D'], kw['H']
dev = torch.device('cuda:0')
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device=dev)
shapes = [(H, D), (H,), (D, H), (D,), (D, D), (D,), (H, D), (H,), (D, H), (D,), (D, D), (D,)]
gtf = [((0.3 if D <= 32 else 0.06) * g(*s)).requires_grad_() for s in shapes]
z0m, z0s = g(D).mul(0.1).requires_grad_(), g(D).mul(0.1).requires_grad_()
experts = []
for m in range(P - 1 if P > 1 else 1):          # modality m is part of pass 0 and pass m + 1
    bits = 1 | (1 << (m + 1)) if P > 1 else 1
    mask = (torch.rand(T, B, device=dev) > 0.1).float()
    experts.append(ops.ExpertSpec(g(T, B, D).requires_grad_(), (g(T, B, D).abs() + 0.3).requires_grad_(), mask, bits, False))
if kw['inv']:                                   # the smoother also fuses the filter posterior, one per pass
    experts.append(ops.ExpertSpec(g(P, T, B, D).requires_grad_(), (g(P, T, B, D).abs() + 0.3).requires_grad_(),
                                  torch.ones(T, B, device=dev), (1 << P) - 1, True))
cfg = ops.SweepCfg(T, B, D, H, P=P, K=K, reverse=bool(kw['rev']), sample=K > 1, use_inv_prior=bool(kw['inv']), seed=7,
                   precision=torch.bfloat16 if kw['bf16'] else torch.float32)
def run():
    outs = ops.bfvi_sweep(cfg, gtf, z0m, z0s, experts)
    loss = sum((o * o).mean() for o in outs if o.numel())
    loss.backward()
run(); torch.cuda.synchronize()
ops.TIMER = ops.KernelTimer()
for _ in range(kw['n']):
    run()
torch.cuda.synchronize()
for k, (n, ms) in sorted(ops.TIMER.summary().items()):
    print('%s %-34s %8.3f ms/launch (%d launches)' % (os.environ.get('TAG', ''), k, ms / n, n))
