"""GPU box: eager cfg4-shaped steps over several noise seeds; where do non-finite GTF bias gradients come from?"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bench
from oracle import mdmm_oracle as orc
from mdmm import models, ops
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
cfg = bench.CONFIGS['cfg4']
lengths = [40, 40, 40, 31, 17, 6]
inputs, targets, mask, _ = cfg.batch(cfg.T, len(lengths), 77, 'cpu')
for d in (inputs, targets):
    for k in d:
        for b, n in enumerate(lengths):
            d[k][n:, b] = float('nan')
mask = orc.len_to_mask(lengths)
to = lambda d: {k: v.to(dev) for k, v in d.items()}
x, tg, mask = to(inputs), to(targets), mask.to(dev)
torch.manual_seed(0)
model = cfg.model(models, dev)
if os.environ.get('F32') == '1':
    model.sweep_dtype = torch.float32
orig = ops.PackedGtf.unpack_grads
def spy(self, G, X, like, contract=None):
    if G is not None:
        badG = (~torch.isfinite(G)).nonzero()
        badX = (~torch.isfinite(X)).nonzero()
        if len(badG) or len(badX):
            B = len(lengths)
            print('   G non-finite at (row, col):', [(int(r), int(c), 't=%d b=%d' % (int(r) // B, int(r) % B)) for r, c in badG[:8]], 'count', len(badG),
                  '| X non-finite:', len(badX), 'G shape', tuple(G.shape), flush=True)
            r = int(badG[0][0]) if len(badG) else int(badX[0][0])
            print('   row', r, 'G[row] finite frac', float(torch.isfinite(G[r]).float().mean()), 'max|G| finite', float(G[torch.isfinite(G)].abs().max()), flush=True)
    return orig(self, G, X, like, contract)
ops.PackedGtf.unpack_grads = spy
for seed in range(int(os.environ.get('SEEDS', 12))):
    model.zero_grad(set_to_none=True)
    model.noise = PhiloxNoise(seed=int(os.environ.get('SEED0', 100)) + (0 if os.environ.get('ADV') else seed))
    if os.environ.get('ADV'):
        model.noise.device_counter(dev).fill_(seed * PhiloxNoise.STRIDE)
    loss = model.step(x, mask, 1.0, cfg.rec, lengths=lengths, targets=tg)
    (loss / sum(lengths)).backward()
    torch.cuda.synchronize()
    bad = [k for k, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print('seed', seed, 'loss', float(loss.detach()), 'bad', bad, flush=True)
