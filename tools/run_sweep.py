"""GPU box: run the cfg2 ELBO step a few times (for rocprofv3 --pmc runs)."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from bench import synth_batch
from mdmm import models
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32, device=dev)
m.noise = PhiloxNoise(seed=1)
inputs, targets, mask, lengths = synth_batch(100, 1024, 1234, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    loss = m.step(inputs, mask, 1.0, {'spiral-x': .5, 'spiral-y': .5}, targets=targets, lengths=lengths)
    (loss / 102400).backward()
    m.zero_grad()
torch.cuda.synchronize()
print('ok', float(loss))
