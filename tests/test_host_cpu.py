"""CPU tests (no GPU needed): the C-ABI library loads and exports every symbol the header
declares; the drop-in package keeps the reference's module tree / state_dict layout; the
host logic (weight packing, replay-noise scheduling, harness arithmetic) is right; and the
product refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import helpers
from helpers import Golden
from oracle import mdmm_oracle as orc


def test_library_exports_every_declared_symbol():
    from mdmm import native
    header = open(os.path.join(helpers.REPO, 'include', 'mdmm_hip.h')).read()
    declared = set(re.findall(r'\b(mdmm_[a-z0-9_]+)\s*\(', header))
    assert declared == set(native.SYMBOLS), declared ^ set(native.SYMBOLS)
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    L = native.lib()
    assert L.mdmm_version() == native.ABI_VERSION
    assert L.mdmm_pad(5) == 8 and L.mdmm_pad(32) == 32
    assert L.mdmm_sweep_spill_width_g(32, 32) == 2 * 32 + 4 * 32
    assert L.mdmm_sweep_spill_width_x(5, 20) == 2 * 8 + 2 * 20
    assert b'LDS' in L.mdmm_strerror(-2)


def test_binding_struct_layout_matches_header():
    """ctypes mirrors of the C structs: sizes derived independently from the header text."""
    from mdmm import native
    assert ctypes.sizeof(native.Gtf) == 12 * 8
    assert ctypes.sizeof(native.Expert) == 5 * 8 + 8 + 4 + 4
    n_int = 12
    expect = n_int * 4 + 8 + 16 + 4 * 8 + 96 + 8 * 56 + 5 * 8 + 5 * 8 + 3 * 8 + 2 * 8 + 8 + 16 + 8
    expect += 8 + 4 + 4 + 8 + 8          # wide family: gtf_frag, precision, reserved1, wide_ws, wide_ws_bytes
    expect += 8 + 8                      # fwd_park, fwd_park_bytes
    expect += 8 + 8 + 8 + 4 + 4          # fused KL term: kld_mask, kld_out, kld_scale_dev, kld_weight, reserved2
    assert ctypes.sizeof(native.Sweep) == expect


def test_argument_errors_are_reported_not_swallowed():
    from mdmm import native
    L = native.lib()
    s = native.Sweep()
    assert L.mdmm_bfvi_sweep_fwd(ctypes.byref(s), None) == -1        # MDMM_E_ARG, no launch
    s.T, s.B, s.D, s.H, s.K, s.P = 4, 2, 8, 8, 1, 9
    assert L.mdmm_bfvi_sweep_fwd(ctypes.byref(s), None) == -2        # too many passes
    with pytest.raises(native.MdmmError):
        native.check(-3, 'x')


def test_no_cpu_fallback():
    """A model on the CPU can hold weights but cannot run: the hot path must fail loudly."""
    from mdmm import models, native
    m = models.MultiDMM(['a', 'b'], [1, 1], z_dim=5, h_dim=20, device=torch.device('cpu'))
    x = {'a': torch.randn(4, 2, 1), 'b': torch.randn(4, 2, 1)}
    mask = torch.ones(4, 2, 1, dtype=torch.bool)
    with pytest.raises(native.MdmmError):
        m(x, lengths=[4, 4])
    with pytest.raises(native.MdmmError):
        m.step(x, mask, 1.0, {}, lengths=[4, 4])
    with pytest.raises(native.MdmmError):
        m.product_of_experts(torch.zeros(2, 3, 4), torch.ones(2, 3, 4))


def test_product_does_not_import_oracle():
    pkg = os.path.join(helpers.PKG_DIR, 'mdmm')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert 'oracle' not in src.replace('oracle/', '').lower() or f == 'noise.py' \
                    or 'import oracle' not in src, f
                assert 'from oracle' not in src and 'import oracle' not in src, f


# ------------------------------------------------------------------ state_dict layout --
def _ref_shapes(case):
    g = Golden('g7_state_dicts.npz')
    return {k: tuple(v.tolist()) for k, v in g.sub(case).items()}


def _shapes(model):
    return {k: tuple(v.shape) for k, v in model.state_dict().items()}


def test_state_dict_layout_matches_reference():
    from mdmm import models
    C = models.common
    cpu = torch.device('cpu')
    got = _shapes(models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=20, z_dim=5, device=cpu))
    assert got == _ref_shapes('spirals_dmm')
    assert list(got) == list(_ref_shapes('spirals_dmm'))            # same key order too
    got = _shapes(models.MultiDKS(['spiral-x', 'spiral-y'], [1, 1], h_dim=20, z_dim=5, device=cpu))
    assert got == _ref_shapes('spirals_dks')
    mods, dims = ['video', 'mask', 'action'], [(3, 64, 64), (1, 64, 64), 10]
    dists = ['Bernoulli', 'Bernoulli', 'Categorical']
    wz = models.MultiDMM(
        mods, dims, dists, encoders={'video': C.ImageEncoder(256, n_channels=3),
                                     'mask': C.ImageEncoder(256, n_channels=1)},
        decoders={'video': C.ImageDecoder(256, n_channels=3),
                  'mask': C.ImageDecoder(256, n_channels=1)}, h_dim=256, z_dim=256, device=cpu)
    assert _shapes(wz) == _ref_shapes('weizmann_dmm')     # incl. the aliased conv keys
    assert sum(p.numel() for p in wz.parameters()) == 7489294
    got = _shapes(models.MultiDKS(
        mods, dims, dists,
        encoders={'video': C.ImageEncoder(256, gauss_out=False, n_channels=3),
                  'mask': C.ImageEncoder(256, gauss_out=False, n_channels=1)},
        decoders={'video': C.ImageDecoder(256, n_channels=3),
                  'mask': C.ImageDecoder(256, n_channels=1)}, h_dim=256, z_dim=256, device=cpu))
    assert got == _ref_shapes('weizmann_dks')
    got = _shapes(models.MultiDMM(
        ['video', 'audio'], [(3, 64, 64), (10, 1281)], ['Bernoulli', 'Bernoulli'],
        encoders={'video': C.ImageEncoder(256), 'audio': C.AudioEncoder(256)},
        decoders={'video': C.ImageDecoder(256), 'audio': C.AudioDecoder(256)},
        h_dim=256, z_dim=256, device=cpu))
    assert got == _ref_shapes('vidtimit_dmm')


def test_default_init_matches_reference_rng_order():
    """Same seed -> same default initialisation as the reference (checkpoint-free parity):
    the golden state_dict of g4/z5 was created by the reference under manual_seed(0)."""
    from mdmm import models
    g = Golden('g4_step.npz')
    torch.manual_seed(0)
    m = models.MultiDMM(['a', 'b'], [1, 1], h_dim=20, z_dim=5, device=torch.device('cpu'))
    for k, v in g.sub('z5/sd').items():
        assert torch.equal(m.state_dict()[k], v), k


def test_models_registry_and_dropin_shim():
    import sys
    from mdmm import models
    assert models.names == {'vrnn': 'MultiVRNN', 'dmm': 'MultiDMM', 'dks': 'MultiDKS'}
    for cls in models.names.values():
        assert hasattr(models, cls)
    shim_dir = os.path.join(helpers.PKG_DIR, 'dropin')
    saved = sys.modules.pop('models', None)
    sys.path.insert(0, shim_dir)
    try:
        import importlib
        shim = importlib.import_module('models')
        assert shim.MultiDMM is models.MultiDMM
        from models.common import GaussianGTF, ImageEncoder   # noqa: F401
        assert getattr(shim, shim.names['dks']) is models.MultiDKS
    finally:
        sys.path.remove(shim_dir)
        for k in [k for k in sys.modules if k == 'models' or k.startswith('models.')]:
            del sys.modules[k]
        if saved is not None:
            sys.modules['models'] = saved


# ------------------------------------------------------------------------ host logic --
def test_gtf_packing_and_grad_unpacking_roundtrip():
    """PackedGtf lays the 12 raw tensors out as mdmm_gtf_t documents; unpack_grads inverts the
    layout: feed G = dL/d(pre-activations), X = activations of a torch GTF and recover
    autograd's parameter gradients."""
    from mdmm import ops
    from mdmm.models.common import GaussianGTF
    torch.manual_seed(1)
    for D, H in ((5, 20), (32, 32), (6, 10)):
        gtf = GaussianGTF(D, H, min_std=1e-3)
        params = ops.gtf_param_list(gtf)
        pk = ops.PackedGtf(params, D, H)
        Dp, Hp, F1 = pk.Dp, pk.Hp, pk.F1
        assert Dp % 4 == 0 and Hp % 4 == 0 and pk.buf.numel() == 2 * (F1 * Dp + 2 * Dp * Hp + Dp * Dp) + F1 + 3 * Dp
        view = lambda name, r, c: pk.buf[pk.offsets[name]:pk.offsets[name] + r * c].reshape(r, c)  # noqa: E731
        w_in = view('w_in', F1, Dp)
        assert torch.equal(w_in[:H, :D], gtf.z_to_gate[0].weight)
        assert torch.equal(w_in[Hp:Hp + H, :D], gtf.z_nonlin[0].weight)
        assert torch.equal(w_in[2 * Hp:2 * Hp + D, :D], gtf.z_lin.weight)
        assert torch.equal(view('wt_in', Dp, F1), w_in.t())
        assert torch.equal(view('w_gate', Dp, Hp)[:D, :H], gtf.z_to_gate[2].weight)
        assert torch.equal(view('wt_std', Dp, Dp), view('w_std', Dp, Dp).t())
        # gradient round trip
        n = 7
        z = torch.randn(n, D)
        a1 = gtf.z_to_gate[0](z); h1 = torch.relu(a1)
        ag = gtf.z_to_gate[2](h1); gate = torch.sigmoid(ag)
        lin = gtf.z_lin(z)
        a2 = gtf.z_nonlin[0](z); h2 = torch.relu(a2)
        nl = gtf.z_nonlin[2](h2)
        pre = gtf.z_to_std[0](nl)
        mean = (1 - gate) * lin + gate * nl
        std = torch.nn.functional.softplus(pre) + 1e-3
        cm, cs = torch.randn(n, D), torch.randn(n, D)
        loss = (mean * cm).sum() + (std * cs).sum()
        acts = [a1, ag, lin, a2, nl, pre]
        g_a1, g_ag, g_lin, g_a2, g_nl, g_pre = torch.autograd.grad(loss, acts, retain_graph=True)
        G = torch.zeros(n, F1 + 3 * Dp); X = torch.zeros(n, 2 * Dp + 2 * Hp)
        G[:, :H] = g_a1; G[:, Hp:Hp + H] = g_a2; G[:, 2 * Hp:2 * Hp + D] = g_lin
        G[:, F1:F1 + D] = g_ag; G[:, F1 + Dp:F1 + Dp + D] = g_nl; G[:, F1 + 2 * Dp:F1 + 2 * Dp + D] = g_pre
        X[:, :D] = z; X[:, Dp:Dp + H] = h1; X[:, Dp + Hp:Dp + Hp + H] = h2
        X[:, Dp + 2 * Hp:Dp + 2 * Hp + D] = nl
        ref_contract = lambda G_, g0, gc, X_, x0, xc: G_[:, g0:g0 + gc].t() @ X_[:, x0:x0 + xc]  # noqa: E731
        got = pk.unpack_grads(G.detach(), X.detach(), params, contract=ref_contract)
        want = torch.autograd.grad(loss, params)
        for a, b in zip(got, want):
            assert helpers.rel_err(a, b) < 1e-5


def test_replay_noise_schedule_counts():
    """The number of draws the fused step consumes equals what the reference drew."""
    from mdmm.models.dmm import _n_draws
    assert _n_draws(7, True, 1, False) == 7 and _n_draws(7, False, 25, False) == 7
    assert _n_draws(7, False, 1, True) == 1 and _n_draws(7, False, 1, False) == 0
    g = Golden('g4_step.npz')
    # z5: 2 prior-matching draws + P*T (bfilter) + P*2T (fsmooth), P = 3, T = 6
    assert len(g.seq('z5/eps')) == 2 + 3 * 6 + 3 * 12
    assert len(g.seq('z5_nouni/eps')) == 2 + 6 + 12


def test_harness_anneal_and_bucket():
    from mdmm import harness
    assert harness.anneal(0.0, 1.0, 24, 2400) == orc.anneal(0.0, 1.0, 24, 2400)
    # first batch of epoch 1 with 24 batches / epoch and kld_anneal=100
    assert harness.kld_multiplier(0, 1, 24, 1.0, 100) == pytest.approx(0.01)
    lin = torch.nn.Linear(3, 2)
    bucket = harness.GradBucket(lin.parameters())
    lin(torch.ones(4, 3)).sum().backward()
    assert bucket.flat.abs().sum() > 0
    assert lin.weight.grad.data_ptr() == bucket.flat.data_ptr()
    flat_before = bucket.flat.clone()
    lin(torch.ones(4, 3)).sum().backward()                 # accumulates in place
    assert torch.allclose(bucket.flat, 2 * flat_before)
    lin.zero_grad(set_to_none=True)
    lin(torch.ones(4, 3)).sum().backward()
    bucket.check_views()
    assert lin.weight.grad.data_ptr() == bucket.flat.data_ptr()
    assert torch.allclose(bucket.flat, flat_before)
    bucket.zero()
    assert float(lin.bias.grad.abs().sum()) == 0.0
    x = {'a': torch.arange(24.).reshape(3, 8, 1)}
    xs, ms, ls = harness.shard_batch(x, torch.ones(3, 8, 1), [3] * 8, 1, 4)
    assert xs['a'].shape == (3, 2, 1) and ls == [3, 3] and torch.equal(xs['a'], x['a'][:, 2:4])


def test_vrnn_scan_layout_and_limits():
    """Host side of the VRNN scan (csrc/vrnn.hip): the spill-row layout the weight gradients index, and
    which widths one CU's LDS takes."""
    from mdmm import native
    L = native.lib()
    a = native.Vrnn()
    a.T, a.B, a.H, a.Z, a.M, a.L = 10, 64, 18, 5, 2, 2          # h = 18 -> 20, z = 5 -> 8
    a.dims[0], a.dims[1] = 3, 2
    lay = native.VrnnLayout()
    assert L.mdmm_vrnn_layout(ctypes.byref(a), ctypes.byref(lay)) == 0
    assert (lay.Hp, lay.Zp, lay.dp[0], lay.dp[1]) == (20, 8, 4, 4)
    parts = (2 * 20 + 20 + 2 * 8 +                       # GRU states, prior hidden / mean / std
             2 * (4 + 20 + 20 + 8 + 8) + 8 +             # per modality: x, phi, encoder hidden, mean, std; z
             2 * (20 + 4 + 4 + 4) +                      # per modality: decoder hidden, mean, std, filled x
             2 * 20 + 20 +                               # recurrence features, phi_z
             2 * (60 + 60 + 20))                         # per GRU layer: input / state products, new state
    assert lay.rows == parts
    offs = [lay.h[0], lay.h[1], lay.ph, lay.pm, lay.ps, lay.z, lay.fz, lay.feat[0], lay.feat[1], lay.gi[1], lay.hn[1]]
    assert all(o % 4 == 0 and 0 <= o < lay.rows for o in offs)
    assert lay.feat[1] == lay.feat[0] + lay.Hp and lay.fz == lay.feat[1] + lay.Hp     # the GRU's input is one block
    assert L.mdmm_vrnn_supported(ctypes.byref(a), 0) == 1 and L.mdmm_vrnn_supported(ctypes.byref(a), 1) == 1
    a.H = a.Z = 256
    assert L.mdmm_vrnn_supported(ctypes.byref(a), 0) == 1 and L.mdmm_vrnn_supported(ctypes.byref(a), 1) == 0
    a.M = 5
    assert L.mdmm_vrnn_supported(ctypes.byref(a), 0) == 0
    assert L.mdmm_vrnn_fwd(ctypes.byref(a), None) < 0           # argument error, nothing launched


def test_vrnn_supported_refuses_more_modalities_than_the_descriptor_holds():
    """ops.vrnn_supported with five modalities: False (-> the model runs step by step), not an IndexError
    from filling the four-element dims array of the descriptor."""
    from mdmm import native, ops
    spec = dict(T=6, B=4, H=16, Z=16, M=native.VRNN_MAX_MODS + 1, L=1, dims=[2] * (native.VRNN_MAX_MODS + 1))
    assert ops.vrnn_supported(spec, backward=False) is False
    spec = dict(T=6, B=4, H=16, Z=16, M=2, L=native.VRNN_MAX_LAYERS + 1, dims=[2, 3])
    assert ops.vrnn_supported(spec, backward=True) is False


def test_conv1d_supported_checks_the_lds_budget():
    """mdmm_conv1d_supported: the stock audio pyramids fit one CU's LDS; a 16-channel, 2048-sample layer
    (262 KB for the down kernel) is reported unsupported so that the caller takes the library convolution."""
    from mdmm import native
    L = native.lib()
    a = native.Conv1d()
    a.N, a.S, a.CS, a.CB = 8, 641, 4, 10            # first encoder layer of AudioEncoder (10 x 1281 -> 4 x 641)
    assert L.mdmm_conv1d_supported(ctypes.byref(a)) == 1
    a.S, a.CS, a.CB = 161, 16, 8                    # last encoder layer
    assert L.mdmm_conv1d_supported(ctypes.byref(a)) == 1
    a.S, a.CS, a.CB = 2048, 8, 16
    assert L.mdmm_conv1d_supported(ctypes.byref(a)) == 0
    assert L.mdmm_conv1d_wgrad_ws_bytes(ctypes.byref(a)) == 0


def test_vrnn_block_padding_round_trip():
    """Weights of the VRNN scan are zero-padded per concatenated part (GRU gates x input blocks)."""
    from mdmm import ops
    torch.manual_seed(0)
    H, M = 5, 2
    w = torch.randn(3 * H, (M + 1) * H)
    p = ops._pad_blocks(w, [H] * 3, [H] * (M + 1))
    assert p.shape == (3 * 8, (M + 1) * 8)
    assert torch.equal(ops._unpad_blocks(p, [H] * 3, [H] * (M + 1)), w)
    assert float(p.abs().sum()) == pytest.approx(float(w.abs().sum()))          # padding is zeros
    assert torch.equal(p[8:8 + H, 16:16 + H], w[H:2 * H, 2 * H:3 * H])         # gate z, last input block
    same = torch.randn(8, 12)
    assert ops._pad_blocks(same, [8], [4, 8]) is same


def test_colsum_launch_geometry():
    from mdmm import native
    L = native.lib()
    assert 1 <= L.mdmm_colsum_splits(10240, 4096) <= 64
    assert L.mdmm_colsum_splits(100, 64) == 1
    assert L.mdmm_colsum(None, 0, 8, 8, 8, None, None, None) < 0


def test_gemm_split_and_conv_parts_planning():
    """Host-side planning entry points (no GPU work): mdmm_gemm_split for the plug-in heads' shapes (csrc/gemm_heads.hip)
    and for the generic tiles; the workgroup counts behind mdmm_conv_t.out_stats."""
    import ctypes as C
    from mdmm import native
    L = native.lib()
    g = native.Gemm()
    # encoder head forward 10,240 x 4096 -> 256, bf16 operands: 80 row tiles -> 3 slices (240 workgroups)
    g.I, g.J, g.L, g.a_bf16, g.b_bf16, g.lda, g.ldb, g.ldc = 10240, 256, 4096, 1, 1, 4096, 4096, 256
    g.a = g.b = g.c = 1 << 20
    assert L.mdmm_gemm_split(C.byref(g)) == 3
    # decoder head 256 -> 4096, bf16 in and out: weight-stationary kernel, never split
    g.I, g.J, g.L, g.lda, g.ldb, g.ldc, g.c_bf16 = 10240, 4096, 256, 256, 256, 4096, 1
    assert L.mdmm_gemm_split(C.byref(g)) == 1
    # weight gradient 4096 x 256 over 10,240 rows: 32 tiles x 8 slices
    g.I, g.J, g.L, g.ta, g.tb, g.lda, g.ldb, g.ldc, g.c_bf16 = 4096, 256, 10240, 1, 1, 4096, 256, 256, 0
    assert L.mdmm_gemm_split(C.byref(g)) == 8
    # fp32 operands: the generic tiles' rule (fewer than two tiles per CU -> slices of at least eight steps)
    g.a_bf16 = g.b_bf16 = 0
    s = L.mdmm_gemm_split(C.byref(g))
    assert 1 <= s <= 64 and s <= (10240 // 32) // 8
    # fp32 operands on the fp32 matrix instruction (MDMM_GEMM_F32): never a head kernel, 16-value steps
    g.I, g.J, g.L, g.ta, g.tb, g.lda, g.ldb, g.ldc = 10240, 256, 4096, 0, 0, 4096, 4096, 256
    g.flags = native.GEMM_F32
    s = L.mdmm_gemm_split(C.byref(g))
    assert 1 <= s <= 64 and s <= (4096 // 16) // 8
    g.split = s
    assert L.mdmm_gemm_ws_bytes(C.byref(g)) == (s * 10240 * 256 * 4 if s > 1 else 0)
    g.flags, g.split = 0, 1
    c = native.Conv()
    c.N, c.S, c.CS, c.CB, c.KS = 20480, 16, 32, 16, 4
    assert L.mdmm_conv_up_parts(C.byref(c)) == 512
    c.N = 100
    assert L.mdmm_conv_up_parts(C.byref(c)) == 100 and 1 <= L.mdmm_conv_down_parts(C.byref(c)) <= 100


def test_clip_flat_matches_clip_grad_norm():
    """harness.clip_flat_ (the captured step's gradient clipping) = torch.nn.utils.clip_grad_norm_ on the same
    gradients (trainer.py:240-241), clipping and not clipping."""
    from mdmm.harness import clip_flat_
    torch.manual_seed(0)
    for scale, max_norm in ((1.0, 0.5), (1e-3, 5.0)):
        ps = [torch.nn.Parameter(torch.zeros(n)) for n in (7, 300, 1025)]
        for p in ps:
            p.grad = torch.randn_like(p) * scale
        flat = torch.cat([p.grad for p in ps]).clone()
        norm = clip_flat_(flat, max_norm)
        ref = torch.nn.utils.clip_grad_norm_(ps, max_norm)
        assert abs(float(norm) - float(ref)) <= 1e-6 * float(ref)
        assert torch.allclose(flat, torch.cat([p.grad for p in ps]), rtol=1e-6, atol=0)


def test_loss_sum_with_a_device_weight():
    """ops.LossSum.scaled (the annealed KLD multiplier of a captured step, trainer.py:227-229): the total is
    acc + w * acc_sub, the sub-sum's terms receive w times the upstream gradient, the others the gradient itself
    (plain torch on the CPU: the terms' kernels only ever see those two device scalars)."""
    from mdmm import ops
    total = ops.LossSum(torch.device('cpu'))
    w = torch.tensor(0.25)
    sub = total.scaled(w)
    assert ops.weighted_into(total, 0.5) == (0.5, total)
    wt, into = ops.weighted_into(total, w)
    assert wt == 1.0 and into is not total and into.acc.dtype == torch.float64

    class Term(torch.autograd.Function):          # what a term's kernel pair does: adds into acc, scales by g
        @staticmethod
        def forward(ctx, x, acc, weight):
            acc += weight * x.double().sum()
            ctx.weight = weight
            return torch.empty(())

        @staticmethod
        def backward(ctx, g):
            return ctx.weight * g.expand(3), None, None

    a, b = torch.tensor([1., 2., 3.], requires_grad=True), torch.tensor([4., 5., 6.], requires_grad=True)
    total.handles.append(Term.apply(a, total.acc, 2.0))
    sub.handles.append(Term.apply(b, sub.acc, 1.0))
    loss = total.total()
    assert abs(float(loss) - (2.0 * 6.0 + 0.25 * 15.0)) < 1e-6
    loss.backward(gradient=torch.tensor(0.5))
    assert torch.allclose(a.grad, torch.full((3,), 0.5 * 2.0))
    assert torch.allclose(b.grad, torch.full((3,), 0.5 * 0.25))


def test_sweep_routing_by_shape_precision_and_switches(monkeypatch):
    """Which kernel family a sweep is sent to (ops.wide_shape; csrc/sweep_wide.hip `plan`, wide_sweep.h `quad_shape`):
    z = h = 256 only; any particle count forward; backward up to 32 rows with fp32 operands, up to 100 particles with
    bf16 ones -- above 64 only while the forward keeps its park; the two switches that pin the generic kernels."""
    import torch
    from mdmm import ops
    for k in ('MDMM_FORCE_GENERIC', 'MDMM_NO_WIDE', 'MDMM_FWD_PARK'):
        monkeypatch.delenv(k, raising=False)
    mk = lambda K, prec, D=256, **kw: ops.SweepCfg(8, 4, D, D, P=2, K=K, precision=prec, **kw)     # noqa: E731
    bf, f32 = torch.bfloat16, torch.float32
    assert not ops.wide_shape(mk(25, bf, D=32)) and not ops.wide_shape(mk(25, bf, trans_only=True))
    assert all(ops.wide_shape(mk(K, p)) for K in (1, 25, 64, 100, 200) for p in (bf, f32))          # forward: any K
    assert [ops.wide_shape(mk(K, f32), bwd=True) for K in (1, 25, 32, 33)] == [True, True, True, False]
    assert [ops.wide_shape(mk(K, bf), bwd=True) for K in (1, 25, 64, 65, 70, 100, 101)] == [True] * 6 + [False]
    monkeypatch.setenv('MDMM_FWD_PARK', '0')            # no park: the one-round backward (and its quad geometry) is off
    assert [ops.wide_shape(mk(K, bf), bwd=True) for K in (25, 64, 65, 100)] == [True, True, False, False]
    monkeypatch.delenv('MDMM_FWD_PARK')
    for sw in ('MDMM_FORCE_GENERIC', 'MDMM_NO_WIDE'):
        monkeypatch.setenv(sw, '1')
        assert not ops.wide_shape(mk(25, bf)) and not ops.wide_shape(mk(25, bf), bwd=True)
        monkeypatch.delenv(sw)
    with pytest.raises(ValueError):
        mk(25, torch.float16)


def test_audio_plans_and_padded_rows_are_decided_on_the_host():
    """mdmm.audio: no plan for a stack that is not on the GPU (there is no CPU path), and the 2,816-wide feature rows only
    with bf16 operands AND bf16 activations at 512 frames or more (a multiple of four).  (Which GPU stacks get a plan:
    tests/test_audio_chain_gpu.py::test_audio_plan_is_for_the_reference_stacks_only.)"""
    import torch
    from mdmm import audio, ops
    from mdmm.models import common as C
    dec, enc = C.AudioDecoder(32), C.AudioEncoder(32)
    with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
        assert audio.decoder_plan(dec) is None and audio.encoder_plan(enc) is None
        assert [audio.padded_rows(n) for n in (508, 512, 514, 516, 65536)] == [False, True, False, True, True]
    with ops.conv_operands(torch.bfloat16, act=torch.float32):
        assert not audio.padded_rows(512)
    with ops.conv_operands(None, act=torch.float32):
        assert not audio.padded_rows(512)
    assert audio.FEAT == 16 * 161 and audio.FEAT_PAD % 256 == 0 and audio.FEAT_PAD >= audio.FEAT
    w, b = audio.pad_linear_out(dec.z_to_feat[0])
    assert w.shape == (audio.FEAT_PAD, 32) and b.shape == (audio.FEAT_PAD,)
    assert torch.equal(w[:audio.FEAT], dec.z_to_feat[0].weight) and float(w[audio.FEAT:].abs().max()) == 0.0
    assert float(b[audio.FEAT:].abs().max()) == 0.0
    wi = audio.pad_linear_in(enc.feat_to_z_mean)
    assert wi.shape == (32, audio.FEAT_PAD) and torch.equal(wi[:, :audio.FEAT], enc.feat_to_z_mean.weight)
    assert float(wi[:, audio.FEAT:].abs().max()) == 0.0


def test_batchnorm_group_routing():
    """ops.bn_groups_for: a batch of n passes is normalised as n groups by ONE launch only when the rows divide, no
    statistics are synchronised and the running statistics use a momentum (not the cumulative average)."""
    import torch.nn as nn
    from mdmm import ops
    bn, cum = nn.BatchNorm2d(4), nn.BatchNorm2d(4, momentum=None)
    assert ops.bn_groups_for(12, bn) == 1
    with ops.bn_groups(3):
        assert ops.bn_groups_for(12, bn) == 3 and ops.bn_groups_for(13, bn) == 1 and ops.bn_groups_for(12, cum) == 1
    assert ops.BN_GROUPS == 1
