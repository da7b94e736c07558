"""GPU box: cfg4 (MultiDKS) at 32 sequences: where do the bf16-operand gradients of the first encoder blocks' BatchNorm
bias differ from the oracle's -- rounding noise on an ill-conditioned sum, or a wrong term?  fp32-operand HIP vs oracle,
bf16 vs oracle, bf16 vs fp32, per parameter with the gradient's own scale."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bench
import helpers
from oracle import mdmm_oracle as orc
import test_replay_gpu as trg
from mdmm import models, ops
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
cfg = bench.CONFIGS[name]
g = torch.Generator().manual_seed(9)
lengths = sorted([cfg.T] * 20 + torch.randint(5, cfg.T, (12,), generator=g).tolist(), reverse=True)
K, T, B, D = bench.TRAIN_PARTICLES, cfg.T, len(lengths), cfg.D
x_cpu, tg_cpu, mask_cpu, x, tg, mask = trg._ragged_batch(cfg, lengths, dev)
n_points = sum(lengths)
kw = dict(train_particles=K) if name == 'cfg3' else {}
grads = {}
for mode in ('bf16', 'fp32'):
    torch.manual_seed(1)
    m = cfg.model(models, dev)
    if os.environ.get('PERTURB_BN', '1') == '1':
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                    mod.bias.uniform_(-0.2, 0.2)
    if mode == 'fp32':
        m.sweep_dtype = m.conv_dtype = m.act_dtype = torch.float32
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    m.noise = PhiloxNoise(seed=66)
    loss = m.step(x, mask, 1.0, cfg.rec, targets=tg, lengths=lengths, **kw)
    (loss / n_points).backward()
    torch.cuda.synchronize()
    grads[mode] = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters() if p.grad is not None}
    print(mode, 'loss', float(loss))
o = cfg.oracle(orc) if name == 'cfg3' else trg._oracle_dks(cfg)
o.load_state_dict(sd); o.train()
noise = PhiloxNoise(seed=66)
P = 1 + cfg.M
if name == 'cfg3':
    draws = [noise.normal((50, 1, D), dev).cpu(), noise.normal((50, 1, D), dev).cpu()]
    sweeps = []
    for k in (1, K, 1):
        s_, off = noise.stream()
        sweeps.append(ops.philox_normal(s_, off, (P, T, k, B, D), dev).cpu())
    for p in range(P):
        draws += [sweeps[0][p, t] for t in reversed(range(T))]
    for p in range(P):
        draws += [sweeps[1][p, t] for t in reversed(range(T))]
        draws += [sweeps[2][p, t] for t in range(T)]
else:
    s_, off = noise.stream()
    eps = ops.philox_normal(s_, off, (T, P, B, D), dev).cpu()
    draws = [eps[t, p] for p in range(P) for t in range(T)]
o.noise = orc.ReplayNoise(draws)
oloss = o.step(x_cpu, mask_cpu, 1.0, cfg.rec, targets=tg_cpu, lengths=lengths, **kw)
(oloss / n_points).backward()
print('oracle loss', float(oloss))
og = {k: p.grad.double() for k, p in o.named_parameters() if p.grad is not None}
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
rows = []
for k in og:
    if k not in grads['bf16']:
        continue
    rows.append((rel(grads['bf16'][k], og[k]), rel(grads['fp32'][k], og[k]), rel(grads['bf16'][k], grads['fp32'][k]), float(og[k].norm()),
                 float(og[k].abs().max()), k))
rows.sort(reverse=True)
print('%10s %10s %10s %10s %10s  %s' % ('bf16/orc', 'fp32/orc', 'bf16/fp32', '|g|', 'max|g|', 'parameter'))
for r in rows[:30]:
    print('%10.3e %10.3e %10.3e %10.3e %10.3e  %s' % r)
