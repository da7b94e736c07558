"""GPU box, diagnostic: per-phase cycle totals of the cooperative backward kernel.
Build first:  hipcc ... -DMDMM_STAMPS -shared -o gpurun_out/libmdmm_stamps.so csrc/*.hip   (tools/stamps.sh)"""
import ctypes, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
from mdmm import native
native.LIB_PATH = os.path.join(R, 'gpurun_out', 'libmdmm_stamps.so')
import torch
from bench import synth_batch
from mdmm import models
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32, device=dev)
m.noise = PhiloxNoise(seed=1)
inputs, targets, mask, lengths = synth_batch(100, 1024, 1234, dev)
for _ in range(3):
    loss = m.step(inputs, mask, 1.0, {'spiral-x': .5, 'spiral-y': .5}, targets=targets, lengths=lengths)
    (loss / 102400).backward()
    m.zero_grad()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
L = native.lib()
L.mdmm_debug_stamps.argtypes = [ctypes.c_void_p]
print('rc', L.mdmm_debug_stamps(buf))
names = sys.argv[1:] or []
for w in range(2):
    v = list(buf[16 * w:16 * w + 16])
    tot = sum(v)
    print('wave', 4 * w, 'total ticks', tot)
    for i, x in enumerate(v):
        if x:
            print('   %2d %-28s %12d  %5.1f%%   per step %8.1f' % (i, names[i] if i < len(names) else '', x, 100.0 * x / tot, x / 297.0))
