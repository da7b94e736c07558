"""-m gpu: the audio plug-ins' stacks as one autograd node each (mdmm.audio on csrc/audio_chain.hip, through the C ABI
mdmm_audio_fwd / mdmm_audio_bwd) against the stock modules of the reference's plug-in classes run by plain PyTorch on
the CPU in fp64 -- the same arithmetic the oracle's cfg5 model runs (bench.Cfg5.oracle builds it from these classes;
reference: common.py:177-290, losses.py:23-42, dmm.py:164-177).  Values, every parameter gradient, the input gradient,
the BatchNorm running statistics and the `seen` flags; fp32 activations tight, bf16 activations at bf16 bounds."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import helpers  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _models():
    from mdmm.models import common
    return common


def _rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _decoder_reference(dec64, z, target, mask, passes, pass_w, weight):
    """sum_p w_p * BCE(sum) of the stock module called pass by pass (dgts.py:132-145), fp64 on the CPU."""
    rows = target.shape[0] * target.shape[1]
    total = 0
    zs = z.reshape(passes, rows, -1)
    tg = target.reshape(rows, 10, 1281)
    for p in range(passes):
        probs = dec64(zs[p])[0]
        on = mask.reshape(rows, 1, 1).bool() & ~torch.isnan(tg)
        t0 = torch.where(on, tg, torch.zeros_like(tg))
        term = F.binary_cross_entropy(probs, t0, reduction='none')
        total = total + pass_w[p] * (term * on).sum()
    return weight * total


@pytest.mark.parametrize('act', ['fp32', 'bf16'])
@pytest.mark.parametrize('passes', [1, 2])
def test_audio_decoder_nll_node(act, passes, dev):
    from mdmm import ops, audio
    C = _models()
    torch.manual_seed(3)
    dec = C.AudioDecoder(32).to(dev).train()
    with torch.no_grad():                       # (non-trivial affine parameters and running statistics)
        for m in dec.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(dec).double().cpu().train()
    t_max, b_dim = 5, 3
    rows = t_max * b_dim
    g = torch.Generator().manual_seed(5)
    z = torch.randn(passes * rows, 32, generator=g)
    target = torch.rand(t_max, b_dim, 10, 1281, generator=g)
    target[3:, 1] = float('nan')                 # padding
    target[1, 0, 2, 100:140] = float('nan')      # interior NaN: that element alone is unobserved
    mask = torch.ones(t_max, b_dim)
    mask[3:, 1] = 0
    mask[4, 2] = 0                               # masked row with finite observations
    pass_w = [0.5, 1.0][:passes]
    weight = 0.7
    bf16 = act == 'bf16'

    zr = z.double().requires_grad_(True)
    loss_ref = _decoder_reference(ref, zr, target.double(), mask.double(), passes, pass_w, weight)
    up = 1.3
    (up * loss_ref).backward()

    zd = z.to(dev).requires_grad_(True)
    total = ops.LossSum(dev)
    with ops.conv_operands(torch.bfloat16 if bf16 else None, act=torch.bfloat16 if bf16 else torch.float32):
        blocks = audio.decoder_plan(dec)
        assert blocks is not None
        dec.nll(zd, blocks, target.to(dev), mask.to(dev), weight, total, passes, pass_w, bf16)
    loss = total.total()
    (up * loss).backward()
    torch.cuda.synchronize()

    tol_v, tol_g = (3e-3, 6e-2) if bf16 else (2e-6, 3e-5)
    assert abs(float(loss) - float(loss_ref)) < tol_v * abs(float(loss_ref)), (float(loss), float(loss_ref))
    worst = {}
    for (k, p), (_, q) in zip(dec.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        if 'deconv.bias' in k and 'deconv_stack.2' not in k:      # a bias in front of a BatchNorm: exactly zero
            assert float(p.grad.abs().max()) == 0.0, k
            continue
        worst[k] = _rel(p.grad, q.grad)
    worst['z'] = _rel(zd.grad, zr.grad)
    for k, e in worst.items():
        helpers.note('audio_dec[%s,%d].grad.%s' % (act, passes, k), e)
    bad = {k: e for k, e in worst.items() if e > tol_g}
    assert not bad, bad
    # running statistics: updated pass by pass, as the stock module's calls would
    for (k, b), (_, r) in zip(dec.named_buffers(), ref.named_buffers()):
        if 'num_batches' in k:
            assert int(b) == int(r) == passes, k
        else:
            assert _rel(b, r) < (2e-3 if bf16 else 1e-5), (k, _rel(b, r))


@pytest.mark.parametrize('act', ['fp32', 'bf16'])
def test_audio_encoder_node(act, dev):
    from mdmm import ops, audio
    C = _models()
    torch.manual_seed(4)
    enc = C.AudioEncoder(24).to(dev).train()
    with torch.no_grad():
        for m in enc.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(enc).double().cpu().train()
    n = 11
    g = torch.Generator().manual_seed(6)
    x = torch.rand(n, 10, 1281, generator=g)
    x[2] = float('nan')
    x[5, 3, 7] = float('nan')
    x[9, :, 1200:] = float('nan')
    bf16 = act == 'bf16'
    cm = torch.randn(n, 24, generator=g).double()
    cs = torch.randn(n, 24, generator=g).double()

    x0 = torch.where(torch.isnan(x), torch.zeros_like(x), x).double()
    mean_r, std_r = ref(x0)
    ((mean_r * cm).sum() + (std_r * cs).sum()).backward()
    seen_r = ~torch.isnan(x).flatten(1).any(1)

    with ops.conv_operands(torch.bfloat16 if bf16 else None, act=torch.bfloat16 if bf16 else torch.float32):
        blocks = audio.encoder_plan(enc)
        assert blocks is not None
        mean, std, seen = enc.encode_frames(x.to(dev), blocks)
    ((mean.double() * cm.to(dev)).sum() + (std.double() * cs.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    assert torch.equal(seen.cpu() > 0, seen_r)
    tol_v, tol_g = (2e-2, 6e-2) if bf16 else (1e-5, 5e-5)
    assert _rel(mean, mean_r) < tol_v and _rel(std, std_r) < tol_v, (_rel(mean, mean_r), _rel(std, std_r))
    worst = {}
    for (k, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()):
        if 'conv.bias' in k and 'conv_stack.2' not in k:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        worst[k] = _rel(p.grad, q.grad)
    for k, e in worst.items():
        helpers.note('audio_enc[%s].grad.%s' % (act, k), e)
    bad = {k: e for k, e in worst.items() if e > tol_g}
    assert not bad, bad
    for (k, b), (_, r) in zip(enc.named_buffers(), ref.named_buffers()):
        if 'num_batches' in k:
            assert int(b) == int(r) == 1, k
        else:
            assert _rel(b, r) < (2e-3 if bf16 else 1e-5), (k, _rel(b, r))



def _run_512(dev, act, padded, monkeypatch):
    """Decoder node (512 frames, one pass) and encoder node (512 frames) on the chip; every gradient by name, and the
    launches' tags."""
    from mdmm import ops, audio
    if not padded:
        monkeypatch.setattr(audio, 'padded_rows', lambda n: False)
    C = _models()
    torch.manual_seed(5)
    dec = C.AudioDecoder(256).to(dev).train()
    enc = C.AudioEncoder(256).to(dev).train()
    with torch.no_grad():
        for m in list(dec.modules()) + list(enc.modules()):
            if isinstance(m, nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    t_max, b_dim = 8, 64
    rows = t_max * b_dim
    g = torch.Generator().manual_seed(7)
    z = torch.randn(rows, 256, generator=g) * 0.5
    target = torch.rand(t_max, b_dim, 10, 1281, generator=g)
    target[6:, 3] = float('nan')
    mask = torch.ones(t_max, b_dim)
    mask[6:, 3] = 0
    x = torch.rand(rows, 10, 1281, generator=g)
    x[5] = float('nan')
    cm = torch.randn(rows, 256, generator=g).double()
    bf16 = act == 'bf16'
    zd = z.to(dev).requires_grad_(True)
    total = ops.LossSum(dev)
    ops.TIMER = timer = ops.KernelTimer()
    try:
        with ops.conv_operands(torch.bfloat16 if bf16 else None, act=torch.bfloat16 if bf16 else torch.float32):
            assert audio.padded_rows(rows) == padded
            dec.nll(zd, audio.decoder_plan(dec), target.to(dev), mask.to(dev), 1.0, total, 1, [1.0], bf16)
            loss = total.total()
            loss.backward()
            mean, std, seen = enc.encode_frames(x.to(dev), audio.encoder_plan(enc))
            ((mean.double() * cm.to(dev)).sum() + std.double().sum()).backward()
        torch.cuda.synchronize()
    finally:
        ops.TIMER = None
    out = {'loss': loss.detach(), 'mean': mean.detach(), 'std': std.detach(), 'seen': seen.detach(), 'dec.z': zd.grad}
    out.update({'dec.' + k: p.grad for k, p in dec.named_parameters()})
    out.update({'enc.' + k: p.grad for k, p in enc.named_parameters()})
    data = dict(z=z, target=target, mask=mask, x=x, cm=cm)
    return out, set(timer.spans), (dec, enc), data


_WIDE = {'linear_fwd[256x2816]', 'linear_fwd[2816x256]', 'linear_wgrad[256x2816]'}
_BEHIND_A_NORM = ('dec.deconv_stack.0.deconv.bias', 'dec.deconv_stack.1.deconv.bias',
                  'enc.conv_stack.0.conv.bias', 'enc.conv_stack.1.conv.bias')


def test_audio_padded_feature_rows_are_a_layout(dev, monkeypatch):
    """512 frames and bf16 activations: the 2576-wide feature side travels as rows of 2816 (zeros behind each frame) so that
    z_to_feat and the encoder heads run on the shape-specialised head kernels (audio.padded_rows; mdmm_audio_t.in_stride /
    out_stride).  The stacks' own results are THE SAME BITS either way (the stride is an address, not arithmetic); the
    Linear layers around them differ by their kernels' summation order alone."""
    a, tags_a, _, _ = _run_512(dev, 'bf16', True, monkeypatch)
    b, tags_b, _, _ = _run_512(dev, 'bf16', False, monkeypatch)
    assert _WIDE <= tags_a and not (_WIDE & tags_b), (sorted(tags_a), sorted(tags_b))
    assert torch.equal(a['seen'], b['seen']) and torch.equal(a['loss'], b['loss'])
    for k in a:
        if k in _BEHIND_A_NORM or k in ('seen', 'loss'):
            continue
        assert a[k].shape == b[k].shape, k
        if 'stack' in k:
            assert torch.equal(a[k], b[k]), k
        else:
            e = _rel(a[k], b[k])
            helpers.note('audio_512[padded vs generic].%s' % k, e)
            assert e < 1e-5, (k, e)


@pytest.mark.parametrize('act', ['fp32', 'bf16'])
def test_audio_nodes_at_512_frames(act, dev, monkeypatch):
    """The same two nodes at 512 frames against the fp64 modules.  The gradients in front of the FIRST BatchNorm's adjoint
    are sums of 164,352 terms per channel that cancel (the adjoint removes the mean, so the terms of the next sum add to
    about zero): every fp32 rounding of a per-channel mean is repeated in all of them and grows with sqrt(n) against the
    sum -- 3.6e-4 with fp32 activations where 15 frames give 3e-5 (PyTorch's CPU kernels accumulate these in fp64 and
    show 1e-6; the fp32 arithmetic of a GPU BatchNorm is ours).  Bounds: 2-3x the measured margins
    (profiles/r06_parity_measured.txt, audio_512[...])."""
    out, tags, (dec, enc), d = _run_512(dev, act, act == 'bf16', monkeypatch)
    bf16 = act == 'bf16'
    dref, eref = copy.deepcopy(dec).double().cpu().train(), copy.deepcopy(enc).double().cpu().train()
    for m in (dref, eref):
        m.zero_grad()
    # (the running statistics were updated by the run above; train-mode arithmetic does not read them)
    zr = d['z'].double().requires_grad_(True)
    loss_ref = _decoder_reference(dref, zr, d['target'].double(), d['mask'].double(), 1, [1.0], 1.0)
    loss_ref.backward()
    x = d['x']
    x0 = torch.where(torch.isnan(x), torch.zeros_like(x), x).double()
    mr, sr = eref(x0)
    ((mr * d['cm']).sum() + sr.sum()).backward()
    ref = {'dec.z': zr.grad}
    ref.update({'dec.' + k: p.grad for k, p in dref.named_parameters()})
    ref.update({'enc.' + k: p.grad for k, p in eref.named_parameters()})
    assert abs(float(out['loss']) - float(loss_ref)) < (3e-3 if bf16 else 2e-6) * abs(float(loss_ref))
    assert torch.equal(out['seen'].cpu() > 0, ~torch.isnan(x).flatten(1).any(1))
    tol_v = 2e-2 if bf16 else 1e-5
    assert _rel(out['mean'], mr) < tol_v and _rel(out['std'], sr) < tol_v
    first = ('dec.z', 'dec.z_to_feat', 'enc.conv_stack.0')          # in front of the first BatchNorm's adjoint
    bad = {}
    for k, r in ref.items():
        if k in _BEHIND_A_NORM:
            continue
        e = _rel(out[k], r)
        helpers.note('audio_512[%s].grad.%s' % (act, k), e)
        tol = (2e-1 if bf16 else 1.6e-3) if k.startswith(first) else (7e-2 if bf16 else 1e-4)
        if e > tol:
            bad[k] = e
    assert not bad, bad


def test_audio_plan_is_for_the_reference_stacks_only(dev):
    """mdmm.audio.decoder_plan / encoder_plan: the reference's own stacks (common.py:177-290: three stride-2 k3 layers on
    10 x 1281 frames, BatchNorm with momentum and running statistics) in training mode get the one-node route; anything
    else -- evaluation mode, a cumulative-average BatchNorm, another kernel size, another width -- goes layer by layer."""
    from mdmm import audio, ops
    C = _models()
    with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
        dec, enc = C.AudioDecoder(32).to(dev), C.AudioEncoder(32).to(dev)
        assert audio.decoder_plan(dec) is not None and audio.encoder_plan(enc) is not None
        dec.eval(); enc.eval()
        assert audio.decoder_plan(dec) is None and audio.encoder_plan(enc) is None
        odd = C.AudioDecoder(32).to(dev)
        odd.deconv_stack[0].net[1] = nn.BatchNorm1d(8, momentum=None).to(dev)       # cumulative average
        assert audio.decoder_plan(odd) is None
        odd = C.AudioEncoder(32).to(dev)
        odd.conv_stack[1].net[0] = nn.Conv1d(4, 8, 5, 2, 2).to(dev)                # another kernel size
        assert audio.encoder_plan(odd) is None
        odd = C.AudioEncoder(32).to(dev)
        odd.conv_stack[0].net[1] = nn.BatchNorm1d(4, affine=False).to(dev)
        assert audio.encoder_plan(odd) is None
    with ops.conv_operands(None, act=torch.float32):
        assert audio.decoder_plan(C.AudioDecoder(32).to(dev)) is not None      # fp32 activations: the same nodes


def test_audio_nodes_match_the_layer_by_layer_route(dev, monkeypatch):
    """MDMM_AUDIO_FUSED=0 selects the route of models.common (csrc/conv1d.hip + batchnorm.hip + reduce.hip): the same
    MultiDMM.step on a small vidTIMIT-shaped batch either way (fp32 operands: the two routes differ by summation order)."""
    import bench
    from mdmm import models
    cfg = bench.CONFIGS['cfg5']
    torch.manual_seed(0)
    model = cfg.model(models, dev)
    model.sweep_dtype = model.conv_dtype = model.act_dtype = torch.float32
    x, tg, mask, lengths = cfg.batch(6, 3, 77, dev)
    out = {}
    for fused in ('1', '0'):
        monkeypatch.setenv('MDMM_AUDIO_FUSED', fused)
        m2 = copy.deepcopy(model)
        from mdmm.noise import PhiloxNoise
        m2.noise = PhiloxNoise(seed=11)
        loss = m2.step(x, mask, 1.0, cfg.rec, targets=tg, lengths=lengths, train_particles=3)
        loss.backward()
        torch.cuda.synchronize()
        out[fused] = (float(loss), {k: p.grad.detach().clone() for k, p in m2.named_parameters() if p.grad is not None})
    la, lb = out['1'][0], out['0'][0]
    assert abs(la - lb) < 2e-6 * abs(lb), (la, lb)
    for k, gb in out['0'][1].items():
        ga = out['1'][1][k]
        if float(gb.abs().max()) == 0.0:
            assert float(ga.abs().max()) == 0.0, k
            continue
        assert _rel(ga, gb) < 2e-4, (k, _rel(ga, gb))
