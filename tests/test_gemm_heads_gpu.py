"""csrc/gemm_heads.hip (-m gpu): the shape-specialised GEMMs of the image plug-ins' Linear heads
(/root/reference/models/common.py:114-175: feat_to_z_mean / feat_to_z_std 4096 -> 256, z_to_feat 256 -> 4096,
applied to all T*B frames) behind mdmm_gemm_bf16 -- `expand` (256 -> N), `contract` (K -> 256) and `wgrad`
(the transposed-read weight gradient) against torch on the same bf16 operands, and against the generic tile
kernel of csrc/gemm_tiles.hip on the same call (MDMM_GEMM_GENERIC=1): ragged row counts, a row slice and
leading dimensions wider than the matrix, both output types, the split the library plans."""
import ctypes as C

import pytest
import torch

import helpers  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def _both(monkeypatch, fn):
    monkeypatch.setenv('MDMM_GEMM_GENERIC', '0')
    own = fn()
    monkeypatch.setenv('MDMM_GEMM_GENERIC', '1')
    gen = fn()
    monkeypatch.setenv('MDMM_GEMM_GENERIC', '0')
    return own, gen


@pytest.mark.parametrize('m,n,odt', [(10240, 4096, torch.bfloat16), (1000, 512, torch.bfloat16), (37, 256, torch.bfloat16),
                                     (243, 4096, torch.bfloat16), (64, 256, torch.bfloat16), (65, 768, torch.bfloat16),
                                     (10240, 768, torch.float32), (1000, 256, torch.float32), (65, 512, torch.float32)])
def test_expand_matches_torch(dev, monkeypatch, m, n, odt):
    from mdmm import ops
    torch.manual_seed(m + n)
    wide = torch.randn(m + 3, 256 + 8, device=dev).bfloat16()
    a = wide[3:, 8:]                                    # rows offset by 3, leading dimension 264
    w = (torch.randn(n, 256, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, device=dev)
    for b_ in (bias, None):
        ref = a.float() @ w.float().t() + (b_ if b_ is not None else 0.0)
        own, gen = _both(monkeypatch, lambda: ops._gemm_bf16(ops._rows(a), False, w, False, m, n, 256, b_, out_dtype=odt))
        assert own.dtype == odt and tuple(own.shape) == (m, n)
        # (bf16 output: a different summation order flips a last bit of one output in ~10^4; 4e-3 per flip)
        assert rel(own, ref.to(odt)) < 1e-4, rel(own, ref.to(odt))
        assert rel(own, gen) < 1e-4


@pytest.mark.parametrize('m,n,k,odt', [(10240, 256, 4096, torch.float32), (1000, 256, 4096, torch.bfloat16),
                                       (243, 256, 512, torch.float32), (130, 512, 1024, torch.float32),
                                       (4096, 256, 9216, torch.float32)])
def test_contract_matches_torch(dev, monkeypatch, m, n, k, odt):
    from mdmm import native, ops
    torch.manual_seed(m + n + k)
    a = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    bias = torch.randn(n, device=dev)
    ref = (a.float() @ w.float().t() + bias).to(odt)
    own, gen = _both(monkeypatch, lambda: ops._gemm_bf16(a, False, w, False, m, n, k, bias, out_dtype=odt))
    tol = 1e-4 if odt is torch.bfloat16 else 3e-6
    assert rel(own, ref) < tol, rel(own, ref)
    assert rel(own, gen) < tol
    # the planned split: about one workgroup per CU, never more slices than four-step pieces of the contraction
    g = native.Gemm()
    g.I, g.J, g.L, g.split, g.a_bf16, g.b_bf16 = m, n, k, 1, 1, 1
    g.a, g.b, g.c, g.lda, g.ldb, g.ldc = a.data_ptr(), w.data_ptr(), own.data_ptr(), k, k, n
    g.c_bf16 = int(odt is torch.bfloat16)
    s = native.lib().mdmm_gemm_split(C.byref(g))
    tiles = ((m + 127) // 128) * (n // 256)
    assert 1 <= s <= max(1, 256 // tiles) and s <= max(1, k // 64 // 4)


@pytest.mark.parametrize('m,i,j', [(10240, 256, 4096), (10240, 4096, 256), (1000, 256, 512), (4099, 512, 256), (513, 256, 128)])
def test_wgrad_matches_torch(dev, monkeypatch, m, i, j):
    from mdmm import ops
    torch.manual_seed(m + i + j)
    g_ = torch.randn(m, i, device=dev).bfloat16()
    x = torch.randn(m, j, device=dev).bfloat16()
    ref = g_.float().t() @ x.float()
    own, gen = _both(monkeypatch, lambda: ops._gemm_bf16(g_, True, x, True, i, j, m))
    assert own.dtype == torch.float32 and tuple(own.shape) == (i, j)
    assert rel(own, ref) < 5e-6, rel(own, ref)
    assert rel(own, gen) < 5e-6


@pytest.mark.parametrize('m,i,j', [(20480, 4096, 256), (10240, 256, 4096), (4099, 512, 256), (513, 256, 128), (1000, 256, 512)])
def test_wgrad_leaves_the_bias_gradient(dev, monkeypatch, m, i, j):
    """mdmm_gemm_t.colsum_a: the weight-gradient launch G^T X of an nn.Linear also leaves G^T 1 -- the sums over the rows of
    the A tiles it stages anyway (row slices folded with the product's slabs) -- equal to the column-sum kernel's on the
    same bf16 matrix; the product itself is unchanged bit for bit; the generic tile kernel declines (None)."""
    from mdmm import ops
    torch.manual_seed(m + i + j)
    g_ = (torch.randn(m, i, device=dev) * 3).bfloat16()
    x = torch.randn(m, j, device=dev).bfloat16()
    plain = ops._gemm_bf16(g_, True, x, True, i, j, m)
    own, cs = ops._gemm_bf16(g_, True, x, True, i, j, m, colsum_a=True)
    assert torch.equal(own, plain)
    assert cs is not None and cs.dtype == torch.float32 and tuple(cs.shape) == (i,)
    ref = g_.double().sum(0)
    assert float((cs.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max() + g_.float().abs().sum(0).max())
    assert rel(cs, ops.colsum(g_)) < 2e-6
    monkeypatch.setenv('MDMM_GEMM_GENERIC', '1')
    gen, none = ops._gemm_bf16(g_, True, x, True, i, j, m, colsum_a=True)
    assert none is None and rel(gen, own) < 5e-6


def test_heads_through_plug_linear(dev):
    """ops.plug_linear on a stock nn.Linear head under conv_operands(bfloat16, act = bfloat16): the autograd
    function hands the kernels bf16 operands (the 256-wide side rounded on the way, the weight and its transpose
    from the per-step pack) -- values and every gradient against torch on the rounded operands."""
    import torch.nn as nn
    from mdmm import ops
    torch.manual_seed(0)
    rb = lambda t: t.to(torch.bfloat16).float()      # noqa: E731
    m = 2000
    enc, dec = nn.Linear(4096, 256).to(dev), nn.Linear(256, 4096).to(dev)
    flat = torch.randn(m, 4096, device=dev).bfloat16().requires_grad_()
    z = torch.randn(m, 256, device=dev, requires_grad=True)
    with ops.conv_operands(torch.bfloat16, torch.bfloat16):
        y_e = ops.plug_linear(enc, flat)
        y_d = ops.plug_linear(dec, z, act_out=True)
    assert y_e.dtype == torch.float32 and y_d.dtype == torch.bfloat16
    ge, gd = torch.randn_like(y_e), torch.randn_like(y_d)
    g_flat, g_we, g_be = torch.autograd.grad(y_e, [flat, enc.weight, enc.bias], ge)
    g_z, g_wd, g_bd = torch.autograd.grad(y_d, [z, dec.weight, dec.bias], gd)
    we, wd = rb(enc.weight.detach()), rb(dec.weight.detach())
    assert rel(y_e, flat.detach().float() @ we.t() + enc.bias.detach()) < 2e-5
    assert rel(y_d, rb(z.detach()) @ wd.t() + dec.bias.detach()) < 8e-3
    assert g_flat.dtype == torch.bfloat16 and rel(g_flat, rb(ge) @ we) < 8e-3
    assert g_z.dtype == torch.float32 and rel(g_z, gd.float() @ wd) < 2e-5
    assert rel(g_we, rb(ge).t() @ flat.detach().float()) < 2e-5
    assert rel(g_wd, gd.float().t() @ rb(z.detach())) < 2e-5
    assert rel(g_be, ge.sum(0)) < 1e-5 and rel(g_bd, gd.float().sum(0)) < 1e-5


@pytest.mark.parametrize('generic', ['0', '1'])
def test_relu_epilogue(dev, monkeypatch, generic):
    """ops.plug_linear(..., relu=True): nn.ReLU behind z_to_feat (common.py:141-148) in the GEMM's epilogue
    (MDMM_GEMM_RELU), on the shape-specialised and on the generic kernel, bf16 and fp32 output, with a split
    contraction too; the adjoint masks the gradient with y > 0."""
    import torch.nn as nn
    from mdmm import ops
    monkeypatch.setenv('MDMM_GEMM_GENERIC', generic)
    torch.manual_seed(1)
    rb = lambda t: t.to(torch.bfloat16).float()      # noqa: E731
    for (k, n, act) in ((256, 4096, True), (256, 512, False), (4096, 256, False)):
        lin = nn.Linear(k, n).to(dev)
        x = torch.randn(1500, k, device=dev)
        if k > 256:
            x = x.bfloat16()
        x.requires_grad_()
        with ops.conv_operands(torch.bfloat16, torch.bfloat16):
            y = ops.plug_linear(lin, x, act_out=act, relu=True)
        ref = torch.relu(rb(x.detach()) @ rb(lin.weight.detach()).t() + lin.bias.detach())
        assert float(y.float().min()) >= 0.0
        assert rel(y, ref) < (8e-3 if act else 2e-5)
        gy = torch.randn_like(y)
        gx, gw, gb = torch.autograd.grad(y, [x, lin.weight, lin.bias], gy)
        gm = gy.float() * (y.float() > 0)
        assert rel(gb, gm.sum(0)) < 1e-5
        assert rel(gw, rb(gm).t() @ rb(x.detach())) < 2e-5
        assert rel(gx, rb(gm) @ rb(lin.weight.detach())) < (8e-3 if gx.dtype == torch.bfloat16 else 2e-5)


@pytest.mark.parametrize('k,n', [(256, 10), (10, 256), (256, 30), (37, 64)])
def test_thin_linear_padded(dev, k, n):
    """ops.linear_tiles_thin: the 10-wide layers of the Categorical modality's stock MLPs (common.py:9-41) on the
    own GEMM with the thin side zero-padded -- values and all three gradients against torch on the bf16-rounded
    operands; the padding rows / columns leave no trace."""
    import torch.nn as nn
    from mdmm import ops
    torch.manual_seed(k + n)
    rb = lambda t: t.to(torch.bfloat16).float()      # noqa: E731
    lin = nn.Linear(k, n).to(dev)
    x = torch.randn(2048, k, device=dev, requires_grad=True)
    assert ops.linear_tiles_thin_supported(x, lin.weight) and not ops.linear_tiles_supported(x, lin.weight)
    y = ops.linear_tiles_thin(x, lin.weight, lin.bias)
    assert tuple(y.shape) == (2048, n)
    assert rel(y, rb(x.detach()) @ rb(lin.weight.detach()).t() + lin.bias.detach()) < 2e-5
    gy = torch.randn(2048, n, device=dev)
    gx, gw, gb = torch.autograd.grad(y, [x, lin.weight, lin.bias], gy)
    assert tuple(gw.shape) == (n, k) and tuple(gx.shape) == (2048, k)
    assert rel(gx, rb(gy) @ rb(lin.weight.detach())) < 2e-5
    assert rel(gw, rb(gy).t() @ rb(x.detach())) < 2e-5
    assert rel(gb, gy.sum(0)) < 1e-5
