"""GPU box: which torch (at::native / rocclr) kernels one eager cfg2 step launches, grouped by op,
input shapes and the autograd node they run under -- the list item 6 of the round-1 review asks to
empty.  usage: python tools/glue_census.py [cfg2|cfg3]"""
import os, sys, collections
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

which = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
dev = torch.device('cuda:0')
from mdmm import models
from mdmm.harness import GradBucket, elbo_step
from mdmm.noise import PhiloxNoise
cfg = bench.Cfg2 if which == 'cfg2' else bench.Cfg3
torch.manual_seed(0)
model = cfg.model(models, dev)
model.noise = PhiloxNoise(seed=1000)
optimizer = torch.optim.Adam(model.parameters(), lr=cfg.lr, fused=True)
bucket = GradBucket(model.parameters())
inputs, targets, mask, lengths = cfg.batch(cfg.T, cfg.B, 1234, dev)
kw = dict(targets=targets, n_points_global=sum(lengths), train_particles=bench.TRAIN_PARTICLES)
step = lambda: elbo_step(model, optimizer, bucket, inputs, mask, lengths, 1.0, cfg.rec, **kw)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = list(prof.events())
# enclosing autograd node / our Function of every CPU op: walk up the cpu_parent chain
def owner(e):
    p = e.cpu_parent
    names = []
    while p is not None:
        n = p.name
        if 'Backward' in n or n.startswith('autograd::') or n.startswith('_') or 'Fn' in n:
            names.append(n.replace('autograd::engine::evaluate_function: ', ''))
        p = p.cpu_parent
    return names[-1] if names else '(forward)'
rows = collections.defaultdict(lambda: [0, 0.0])
ours = 0.0
total = 0.0
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    t = sum(k.duration for k in e.kernels)
    total += t
    if not e.name.startswith('aten::'):
        ours += t
        continue
    key = (e.name.replace('aten::', ''), str(e.input_shapes)[:70], owner(e)[:44])
    rows[key][0] += len(e.kernels); rows[key][1] += t
print('%s: device time of one step %.3f ms, aten ops %.3f ms (%.1f%%)' % (which, total / 1e3, (total - ours) / 1e3, 100 * (total - ours) / total))
for (name, shapes, own), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%8.1f us x%-3d %-18s %-70s %s' % (t, n, name, shapes, own))
