import os, sys
sys.path.insert(0, '' + os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'multimodal-dmm_amd') + '')
import torch, mdmm
from mdmm import ops
from mdmm.models import common as C
dev = torch.device('cuda:0')
dec = C.ImageDecoder(256, n_channels=3).to(dev).train()
z = torch.randn(528, 256, device=dev, requires_grad=True)
orig = ops.lazy_bn_ok
def spy(x_pre):
    fn = getattr(x_pre, 'grad_fn', None)
    r = orig(x_pre)
    print('lazy_bn_ok:', r, type(fn).__name__, getattr(fn, 'lazy_consumer', None), tuple(x_pre.shape), x_pre.dtype, x_pre.is_contiguous())
    return r
ops.lazy_bn_ok = spy
ops.TIMER = ops.KernelTimer()
orig_sup = ops.bn_deconv_supported
def spy2(p, l):
    r = orig_sup(p, l)
    print('bn_deconv_supported', r, p.x_pre.dtype, ops.ACT_STORAGE, ops.conv_tiles_supported(l, p.x_pre), p.bn.training, ops.BN_GROUPS)
    return r
ops.bn_deconv_supported = spy2
orig_ct = ops.conv_tiles
def spy3(layer, x, bias=True, stats_for=None):
    print('conv_tiles', tuple(x.shape), x.dtype, ops.BN_DEFER, ops.ACT_STORAGE)
    return orig_ct(layer, x, bias=bias, stats_for=stats_for)
ops.conv_tiles = spy3
with ops.conv_operands(torch.bfloat16, torch.bfloat16), ops.bn_groups(1):
    out = dec(z, logits=True)[0]
g = torch.autograd.grad(out, [z] + list(dec.parameters()), torch.randn_like(out), allow_unused=True)
print(sorted(ops.TIMER.spans))
