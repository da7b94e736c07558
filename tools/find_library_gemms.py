"""GPU box: which Python call sites still reach a library GEMM in one eager cfg3 step (forward side; the backward of
each is the library's too).  Wraps F.linear / addmm / mm / matmul / bmm and prints shapes with the calling line."""
import os, sys, traceback, collections
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import bench
import torch
import torch.nn.functional as F
from mdmm import models
from mdmm.harness import GradBucket, elbo_step
from mdmm.noise import PhiloxNoise
seen = collections.Counter()


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        shapes = [tuple(t.shape) for t in a if torch.is_tensor(t)]
        st = [fr for fr in traceback.extract_stack(limit=8) if 'mdmm' in fr.filename and 'find_library' not in fr.filename]
        where = '%s:%d' % (os.path.basename(st[-1].filename), st[-1].lineno) if st else '?'
        seen[(name, str(shapes), where)] += 1
        return orig(*a, **k)
    setattr(mod, name, f)


for mod, name in ((F, 'linear'), (torch, 'addmm'), (torch, 'mm'), (torch, 'matmul'), (torch, 'bmm'), (torch, 'einsum')):
    wrap(mod, name)
_mm = torch.Tensor.__matmul__


def mm(self, other):
    st = [fr for fr in traceback.extract_stack(limit=8) if 'mdmm' in fr.filename]
    seen[('@', str([tuple(self.shape), tuple(other.shape)]), '%s:%d' % (os.path.basename(st[-1].filename), st[-1].lineno) if st else '?')] += 1
    return _mm(self, other)


torch.Tensor.__matmul__ = mm
dev = torch.device('cuda:0')
cfg = bench.CONFIGS['cfg3']
torch.manual_seed(0)
model = cfg.model(models, dev)
model.noise = PhiloxNoise(seed=1)
opt = torch.optim.Adam(model.parameters(), lr=cfg.lr)
bucket = GradBucket(model.parameters())
inputs, targets, mask, lengths = cfg.batch(cfg.T, 32, 1, dev)
elbo_step(model, opt, bucket, inputs, mask, lengths, 1.0, cfg.rec, targets=targets, train_particles=bench.TRAIN_PARTICLES)
torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
