"""The conv plug-ins of models.common against golden G9 -- VALUES recorded from the reference's own
ImageEncoder / ImageDecoder / AudioEncoder / AudioDecoder (/root/reference/models/common.py:70-290;
tests/golden/make_golden.py::g9_plugins): training-mode outputs, the gradient of a fixed linear
functional with respect to every parameter and the input, the BatchNorm running statistics after that
forward, and the evaluation-mode outputs.  A wiring difference (layer order, final Sigmoid, padding,
channel schedule, which statistics a mode uses) shows here; the state-dict goldens (G7) only pin names.

CPU: the product classes on stock torch ops.  GPU (-m gpu): the same classes on the own kernels --
fp32 (fused BatchNorm + ReLU, the audio pyramids' conv1d kernels) at fp32 tolerances, and with
conv_operands(bfloat16) (own bf16-operand convolutions and Linear heads) at operand-rounding ones."""
import pytest
import torch

from helpers import Golden, rel_err

NAMES = ['image_enc', 'image_enc_feat', 'image_dec', 'image_dec_mask', 'audio_enc', 'audio_enc_feat', 'audio_dec']
SPECS = {
    'image_enc': ('ImageEncoder', dict(z_dim=8, n_channels=3)),
    'image_enc_feat': ('ImageEncoder', dict(z_dim=8, gauss_out=False, n_channels=1)),
    'image_dec': ('ImageDecoder', dict(z_dim=8, n_channels=3)),
    'image_dec_mask': ('ImageDecoder', dict(z_dim=8, n_channels=1)),
    'audio_enc': ('AudioEncoder', dict(z_dim=8)),
    'audio_enc_feat': ('AudioEncoder', dict(z_dim=8, gauss_out=False)),
    'audio_dec': ('AudioDecoder', dict(z_dim=8)),
}


def _run(name, device, ctx=None, tol_out=1e-5, tol_grad=1e-4, tol_run=1e-5, tol_gx=None):
    from contextlib import nullcontext
    from mdmm.models import common as C
    g = Golden('g9_plugins.npz')
    cls, kw = SPECS[name]
    m = getattr(C, cls)(**kw)
    m.load_state_dict(g.sub(name + '/sd'))
    m.to(device)
    x = g.t(name + '/x').to(device).requires_grad_()
    w = [t.to(device) for t in g.seq(name + '/w')]
    want = g.seq(name + '/train')
    m.train()
    with (ctx() if ctx else nullcontext()):
        res = m(x)
        res = list(res) if isinstance(res, tuple) else [res]
        assert len(res) == len(want)
        for i, (a, b) in enumerate(zip(res, want)):
            assert tuple(a.shape) == tuple(b.shape)
            e = rel_err(a.float(), b)
            assert e < tol_out, '%s train output %d: %.3e' % (name, i, e)
        sum((r.float() * w_).sum() for r, w_ in zip(res, w)).backward()
    e = rel_err(x.grad, g.t(name + '/gx'))
    assert e < (tol_gx or tol_grad), '%s input gradient: %.3e' % (name, e)
    grads = g.sub(name + '/grads')
    gmax = max(float(v.abs().max()) for v in grads.values())
    for k, p in m.named_parameters():
        ref = grads[k]
        if float(ref.abs().max()) < 1e-5 * gmax:         # conv biases in front of a BatchNorm: zero up to rounding
            continue
        got = p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(ref)
        e = float((got - ref).norm() / (ref.norm() + 1e-30))
        assert e < tol_grad, '%s grad %s: %.3e' % (name, k, e)
    sd = m.state_dict()
    for k, ref in g.sub(name + '/running').items():
        e = rel_err(sd[k].float(), ref.float())
        assert e < tol_run, '%s %s: %.3e' % (name, k, e)
    m.eval()
    with torch.no_grad(), (ctx() if ctx else nullcontext()):
        ev = m(x)
    ev = list(ev) if isinstance(ev, tuple) else [ev]
    for i, (a, b) in enumerate(zip(ev, g.seq(name + '/eval'))):
        e = rel_err(a.float(), b)
        assert e < tol_out, '%s eval output %d: %.3e' % (name, i, e)


@pytest.mark.parametrize('name', NAMES)
def test_plugins_match_reference_cpu(name):
    _run(name, torch.device('cpu'))


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_plugins_match_reference_gpu_fp32(name):
    """fp32 on the GPU: library convolutions (2-D) / own conv1d kernels, own fused BatchNorm + ReLU."""
    _run(name, torch.device('cuda:0'), tol_out=2e-5, tol_grad=2e-4, tol_run=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('act', [torch.float32, torch.bfloat16], ids=['act_fp32', 'act_bf16'])
@pytest.mark.parametrize('name', NAMES)
def test_plugins_match_reference_gpu_bf16_operands(name, act):
    """conv_operands(bfloat16): the own bf16-operand convolution / GEMM kernels (csrc/conv_tiles.hip,
    gemm_tiles.hip), activations stored as fp32 or bf16; tolerances = operand (and storage) rounding.
    The reference's values are the yardstick, not torch on the same rounded operands."""
    from mdmm import ops
    tol_o = 2e-2 if act is torch.bfloat16 else 1e-2
    # Three frames are a tiny batch for BatchNorm statistics: operand rounding moves the per-channel gradients
    # by up to 1.1e-1 here (parameters), and the gradient with respect to the input FRAMES of an encoder -- three
    # bf16 layers and two normalisations deep, never formed in a model (frames do not require it) -- by 2e-1.
    _run(name, torch.device('cuda:0'), ctx=lambda: ops.conv_operands(torch.bfloat16, act), tol_out=tol_o,
         tol_grad=1.5e-1, tol_run=1e-2, tol_gx=3e-1 if 'enc' in name else None)
