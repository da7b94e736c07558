"""ImageDecoder forward + backward at the cfg3 per-term size (10,240 frames), single stream: run under
rocprofv3 --kernel-trace --stats for isolated kernel durations (tools/_run2.sh).  argv[1]: repetitions."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'multimodal-dmm_amd'))
import torch
import mdmm  # noqa: F401
from mdmm import ops
from mdmm.models import common as C

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
torch.manual_seed(0)
dec = C.ImageDecoder(256, n_channels=3).to(dev).train()
z = torch.randn(10240, 256, device=dev, requires_grad=True)
for i in range(reps + 1):
    with ops.conv_operands(torch.bfloat16, torch.bfloat16):
        out = dec(z, logits=True)[0]
    gy = torch.randn_like(out)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    torch.autograd.grad(out, [z] + list(dec.parameters()), gy)
    t1.record()
    torch.cuda.synchronize()
    if i:
        print('backward %.3f ms' % t0.elapsed_time(t1))
