"""GPU box: which tensors the elementwise add / mul / copy / cat kernels of one eager cfg2 step
touch (shape histogram per op, forward vs autograd thread), to find avoidable gradient traffic."""
import os, sys, collections
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synth_batch
from mdmm import models
from mdmm.harness import GradBucket
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32, device=dev)
m.noise = PhiloxNoise(seed=1)
inputs, targets, mask, lengths = synth_batch(100, 1024, 1234, dev)
bucket = GradBucket(m.parameters())
rec = {'spiral-x': .5, 'spiral-y': .5}
def fb():
    l = m.step(inputs, mask, 1.0, rec, targets=targets, lengths=lengths); (l / 102400).backward()
fb(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    fb(); torch.cuda.synchronize()
hist = collections.Counter(); dur = collections.Counter()
main_tid = None
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels: continue
    if main_tid is None: main_tid = e.thread
    key = (e.name.replace('aten::', ''), 'fwd' if e.thread == main_tid else 'bwd', str(e.input_shapes)[:90])
    hist[key] += len(e.kernels); dur[key] += sum(k.duration for k in e.kernels)
for key, n in sorted(hist.items(), key=lambda kv: -dur[kv[0]])[:60]:
    print('%4d x %-22s %s %8.1f us  %s' % (n, key[0], key[1], dur[key], key[2]))
