"""Step-by-step run of mdmm.batch.pad_and_merge on the GPU (debugging aid)."""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'multimodal-dmm_amd'))
import numpy as np
import torch

def say(*a):
    print(*a, flush=True)

dev = torch.device('cuda:0')
say('empty'); out = torch.empty((12, 7, 3), dtype=torch.float32, device=dev)
torch.cuda.synchronize(); say('ok')
flat = np.random.rand(48, 3).astype(np.float32)
say('from_numpy.to'); flat_d = torch.from_numpy(flat).to(dev, non_blocking=True); torch.cuda.synchronize(); say('ok')
off = np.concatenate([[0], np.cumsum([5, 9, 9, 3, 12, 1, 9])[:-1]]).astype(np.int64)
say('as_tensor int64', off.dtype, off.flags)
t = torch.as_tensor(np.asarray(off, dtype=np.int64)); say('cpu ok', t)
t = t.to(dev); torch.cuda.synchronize(); say('to ok')
t2 = torch.as_tensor(np.asarray(off, dtype=np.int64), device=dev); torch.cuda.synchronize(); say('as_tensor(device) ok')
t3 = torch.as_tensor(np.asarray([4, 1, 2, 6, 0, 3, 5], dtype=np.int32), device=dev); say('i32 ok')
from mdmm import native, batch
say('lib'); L = native.lib(); say('lib ok', L.mdmm_version())
items = [np.random.rand(n, 3).astype(np.float32) for n in (5, 9, 9, 3, 12, 1, 9)]
say('pad_and_merge'); x = batch.pad_and_merge(items, device=dev); torch.cuda.synchronize(); say('ok', x.shape, torch.isnan(x).sum().item())
