#!/usr/bin/env python
"""ELBO-step benchmark of the MI355X-native MDMM path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config cfg3|cfg2]

One "step" = the body of trainer.py:237-252 on one synthetic batch already resident in HBM:
loss = model.step(...); (loss / n_points).backward(); [all-reduce]; Adam.step(); zero_grad.

Workload (default, the configuration BASELINE.json's targets are quoted on): configs[2] "cfg3",
Weizmann-shaped synthetic -- video (3,64,64) + mask (1,64,64) Bernoulli, action Categorical(10),
conv encoders / decoders of weizmann.py:63-68, MultiDMM BFVI, z = h = 256, T = 40, 256 sequences
per GPU, 20 % burst NaN, 25 training particles; the dense contractions of the sweeps, convolutions
and projections run with bf16 operands and fp32 accumulation, the conv-chain activations are stored
as bf16 (MultiDGTS.sweep_dtype / conv_dtype / act_dtype); latents, statistics, reductions fp32.  The
step is replayed from HIP graphs (`--eager` opts out; `config.execution` says which, and names the
graph executor's stream count this script asks for).  `--config cfg2` is the
Spirals-synthetic z = h = 32 case of round 1; it and cfg4 (MultiDKS on the cfg3 batch) ride along as
`extra.cfg2` / `extra.cfg4` at N = 1.

`--gpus N` with N > 1 starts N ranks itself (torch.distributed.run, one process per GPU, RCCL)
unless it already runs under a launcher (WORLD_SIZE set, which must then equal N).  Every rank
gets its own shard (weak scaling); one all-reduce of the flat gradient bucket per step.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the sweep with the largest device time in the step (HIP events on the launch
                  stream, mdmm.ops.KernelTimer) against BOTH roofs: `bound` is the roof that takes
                  longer for that launch (flops / matrix-pipe peak of the operand type vs HBM bytes /
                  8 TB/s, with the measured PMC traffic of profiles/ standing in for the algorithmic
                  bytes where a record of this shape exists); `mfma` and `hbm` carry both fractions.
                  Algorithmic counts: SURVEY 8d (backward = 2 x forward); `executed_flops` = what the
                  kernels really do (recompute included);
  roofline_k1  -- the same for the largest K = 1 MAP sweep, the kernel the north star's HBM target names;
  config.replay_matches_eager -- graph-replayed step vs the same step eagerly (same weights, same
                  Philox stream): relative loss / gradient difference;
  cpu_baseline -- the CPU oracle (oracle/mdmm_oracle.py, a per-timestep torch-CPU port of the
                  reference) timed on this box's host cores on a bounded sample of the same
                  workload at 1 thread, all cores and a few counts between (rank 0, N = 1 only), with
                  its step-time ratio to the unmodified reference as measured in the build container;
  elbo_delta   -- |loss_hip - loss_oracle| / |loss_oracle| of one step on that sample, the kernels' noise
                  replayed into the oracle, for the timed precision mode and for fp32 operands;
  extra        -- cfg2, cfg4, cfg5 (at its per-GPU size) and cfg3 with fp32 operands ride along at N = 1, and the callers either side of the step
                  (batch_prep: collate + burst deletion on the device; eval: 200-particle evaluation forward + metrics
                  + decollate; tools/bench_callers.py).
"""
import argparse
import json
import os
import signal
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, 'multimodal-dmm_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# MIOpen benchmarks every applicable solver the first time it sees a convolution (a fresh box has no
# find cache).  Its reference ("naive") and im2col + GEMM solvers never win for the plug-in stacks'
# shapes but take 2.5 of the 3.5 minutes that search costs (a naive weight-gradient candidate runs
# 0.9 s per call): leave them out of the search.  The kernels selected for the timed steps are the
# same with or without this (Winograd / implicit-GEMM assembly kernels; 118-121 ms per step).
for _k in ('MIOPEN_DEBUG_CONV_GEMM', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD',
           'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
    os.environ.setdefault(_k, '0')

# Graph replay workaround of this ROCm (mdmm/__init__.py: the runtime's graph packet capture faults replayed steps);
# read when the runtime initialises, so set before torch touches the GPU.  An exported value wins.
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

TRAIN_PARTICLES = 25
F32_PEAK_TFLOPS = 157.3          # MI355X dense f32 (vector = f32-input MFMA), MI355X_MICROARCH.md
BF16_PEAK_TFLOPS = 2500.0        # dense bf16 MFMA


def gtf_flops(d, h):
    return 8 * d * h + 4 * d * d          # common.py:62-68, SURVEY 8a-3


def burst_nan(x, t_max, frac, g):
    """multiseq.py burst deletion, vectorised: one NaN burst of frac*T steps per sequence."""
    import torch
    b_dim = x.shape[1]
    burst = max(1, int(t_max * frac))
    start = torch.randint(0, t_max - burst + 1, (b_dim,), generator=g)
    idx = torch.arange(t_max).unsqueeze(1)
    hole = (idx >= start.unsqueeze(0)) & (idx < (start + burst).unsqueeze(0))
    x[hole] = float('nan')
    return x


# ---------------------------------------------------------------------------------------------
# cfg2: Spirals-synthetic, z = h = 32
# ---------------------------------------------------------------------------------------------
class Cfg2:
    name, T, B, D, H, M = 'cfg2', 100, 1024, 32, 32, 2
    dtype, peak, lr = 'f32', F32_PEAK_TFLOPS, 1e-3
    rec = {'spiral-x': .5, 'spiral-y': .5}
    workload = ('cfg2: Spirals-synthetic MultiDMM BFVI, M=2, z=32, h=32, T=100, B=%d per GPU, '
                '10%% burst NaN, train_particles=25')

    @staticmethod
    def batch(t_max, b_dim, seed, device):
        """x, y ~ N(0,1) (T,B,1); inputs = targets with a 10 % NaN burst."""
        import torch
        g = torch.Generator().manual_seed(seed)
        targets = {m: torch.randn(t_max, b_dim, 1, generator=g) for m in ('spiral-x', 'spiral-y')}
        inputs = {m: burst_nan(v.clone(), t_max, 0.1, g) for m, v in targets.items()}
        mask = torch.ones(t_max, b_dim, 1, dtype=torch.bool)
        dev = lambda d: {k: v.to(device) for k, v in d.items()}   # noqa: E731
        return dev(inputs), dev(targets), mask.to(device), [t_max] * b_dim

    @staticmethod
    def model(models, device):
        return models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32, device=device)

    @staticmethod
    def oracle(orc):
        return orc.OracleDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32)


def synth_batch(t_max, b_dim, seed, device):      # kept for the tools that import it
    return Cfg2.batch(t_max, b_dim, seed, device)


# ---------------------------------------------------------------------------------------------
# cfg3: Weizmann-shaped, z = h = 256, conv plug-ins
# ---------------------------------------------------------------------------------------------
class Cfg3:
    name, T, B, D, H, M = 'cfg3', 40, 256, 256, 256, 3
    dtype, peak, lr = 'bf16', BF16_PEAK_TFLOPS, 1e-4
    rec = {'video': 1.0, 'mask': 1.0, 'action': 10.0}
    workload = ('cfg3: Weizmann-shaped synthetic (video 3x64x64 + mask 1x64x64 Bernoulli, action '
                'Categorical(10)), MultiDMM BFVI, conv encoders/decoders, z=h=256, T=40, B=%d per GPU, '
                '20%% burst NaN, train_particles=25, sweep / conv / projection contractions with bf16 operands and fp32 '
                'accumulation, conv-chain activations stored as bf16; latents, statistics, reductions fp32')
    mods, dims = ['video', 'mask', 'action'], [(3, 64, 64), (1, 64, 64), 10]
    dists = ['Bernoulli', 'Bernoulli', 'Categorical']

    @staticmethod
    def batch(t_max, b_dim, seed, device):
        """video U(0,1), mask Bernoulli(.5), action constant over time; 20 % burst."""
        import torch
        g = torch.Generator().manual_seed(seed)
        tg = {'video': torch.rand(t_max, b_dim, 3, 64, 64, generator=g),
              'mask': (torch.rand(t_max, b_dim, 1, 64, 64, generator=g) < 0.5).float(),
              'action': torch.randint(0, 10, (1, b_dim, 1), generator=g).float().expand(t_max, b_dim, 1).contiguous()}
        x = {k: burst_nan(v.clone(), t_max, 0.2, g) for k, v in tg.items()}
        mask = torch.ones(t_max, b_dim, 1, dtype=torch.bool)
        dev = lambda d: {k: v.to(device) for k, v in d.items()}   # noqa: E731
        return dev(x), dev(tg), mask.to(device), [t_max] * b_dim

    @classmethod
    def _plugins(cls, C):
        enc = {'video': C.ImageEncoder(256, n_channels=3), 'mask': C.ImageEncoder(256, n_channels=1)}
        dec = {'video': C.ImageDecoder(256, n_channels=3), 'mask': C.ImageDecoder(256, n_channels=1)}
        return enc, dec

    @classmethod
    def model(cls, models, device):
        import torch
        enc, dec = cls._plugins(models.common)
        m = models.MultiDMM(cls.mods, cls.dims, cls.dists, encoders=enc, decoders=dec, h_dim=256,
                            z_dim=256, device=device)
        m.sweep_dtype = torch.bfloat16
        m.conv_dtype = torch.bfloat16
        m.act_dtype = torch.bfloat16
        return m

    @classmethod
    def oracle(cls, orc):
        from mdmm.models import common as C       # the plug-in conv stacks are plain torch modules
        enc, dec = cls._plugins(C)
        return orc.OracleDMM(cls.mods, cls.dims, cls.dists, encoders=enc, decoders=dec, h_dim=256,
                             z_dim=256)


class Cfg3F32(Cfg3):
    """cfg3 with every contraction on fp32 operands: the precision mode the 1e-5 parity tests run in."""
    name, dtype, peak = 'cfg3', 'f32', F32_PEAK_TFLOPS
    workload = ('cfg3 (fp32 operands): Weizmann-shaped synthetic, MultiDMM BFVI, conv encoders/decoders, z=h=256, T=40, '
                'B=%d per GPU, 20%% burst NaN, train_particles=25; every contraction with fp32 operands (sweeps on the '
                'f32-input MFMA, convolutions in the library), activations fp32')

    @classmethod
    def model(cls, models, device):
        import torch
        m = super().model(models, device)
        m.sweep_dtype = m.conv_dtype = m.act_dtype = torch.float32
        return m


class Cfg3F32Own(Cfg3F32):
    """... and its convolutions on the own fp32-operand path (MultiDGTS.conv_f32_own, csrc/conv_f32.hip)"""
    workload = Cfg3F32.workload.replace('convolutions in the library', 'convolutions on the own fp32-operand products')

    @classmethod
    def model(cls, models, device):
        m = super().model(models, device)
        m.conv_f32_own = True
        return m


class Cfg4(Cfg3):
    """BASELINE configs[3] at its per-GPU size (2048 sequences on 8 GPUs): the same Weizmann-shaped batch and
    plug-ins under MultiDKS, backward-RNN with skip updates (B-Skip), feat_to_z, uni_loss."""
    name = 'cfg4'
    workload = ('cfg4: Weizmann-shaped synthetic, MultiDKS backward RNN (B-Skip), feat_to_z, conv encoders/decoders, '
                'z=h=256, T=40, B=%d per GPU, 20%% burst NaN; recurrence / projection / conv contractions with bf16 '
                'operands and fp32 accumulation, conv-chain activations stored as bf16')

    @classmethod
    def model(cls, models, device):
        import torch
        C = models.common
        enc = {'video': C.ImageEncoder(256, gauss_out=False, n_channels=3),
               'mask': C.ImageEncoder(256, gauss_out=False, n_channels=1)}
        dec = {'video': C.ImageDecoder(256, n_channels=3), 'mask': C.ImageDecoder(256, n_channels=1)}
        m = models.MultiDKS(cls.mods, cls.dims, cls.dists, encoders=enc, decoders=dec, h_dim=256, z_dim=256,
                            feat_to_z=True, rnn_dir='bwd', rnn_skip=True, device=device)
        m.sweep_dtype = m.conv_dtype = m.act_dtype = torch.bfloat16
        return m


class Cfg5:
    """BASELINE configs[4] at its per-GPU size (4096 sequences on 8 GPUs): vidTIMIT-shaped video (3,64,64) +
    audio spectrogram slices (10,1281), both Bernoulli (vidTIMIT.py:50-69), every (t, b, modality) missing
    independently with p = 0.5, ragged lengths U{64..128} sorted descending, MultiDMM BFVI."""
    name, T, B, D, H, M = 'cfg5', 128, 512, 256, 256, 2
    dtype, peak, lr = 'bf16', BF16_PEAK_TFLOPS, 1e-4
    rec = {'video': 1.0, 'audio': 1.0}
    workload = ('cfg5: vidTIMIT-shaped synthetic (video 3x64x64 + audio 10x1281, Bernoulli), MultiDMM BFVI, conv '
                'encoders/decoders, z=h=256, T=128, B=%d per GPU, ragged lengths 64..128, 50%% independent missingness, '
                'train_particles=25; sweep / conv / projection contractions with bf16 operands and fp32 accumulation, '
                'image conv-chain activations stored as bf16')
    mods, dims = ['video', 'audio'], [(3, 64, 64), (10, 1281)]
    dists = ['Bernoulli', 'Bernoulli']

    @staticmethod
    def batch(t_max, b_dim, seed, device, lengths=None):
        import torch
        g = torch.Generator().manual_seed(seed)
        if lengths is None:
            lengths = sorted(torch.randint(t_max // 2, t_max + 1, (b_dim,), generator=g).tolist(), reverse=True)
            lengths[0] = t_max
        mask = torch.zeros(t_max, b_dim, 1, dtype=torch.bool)
        for b, n in enumerate(lengths):
            mask[:n, b] = True
        pad = ~mask.squeeze(-1)
        tg, x = {}, {}
        for k, shape in (('video', (3, 64, 64)), ('audio', (10, 1281))):     # one modality at a time on the host
            v = torch.rand(t_max, b_dim, *shape, generator=g)
            v[pad] = float('nan')
            tg[k] = v.to(device)
            v[torch.rand(t_max, b_dim, generator=g) < 0.5] = float('nan')
            x[k] = v.to(device)
            del v
        return x, tg, mask.to(device), lengths

    @classmethod
    def _plugins(cls, C):
        return ({'video': C.ImageEncoder(256), 'audio': C.AudioEncoder(256)},
                {'video': C.ImageDecoder(256), 'audio': C.AudioDecoder(256)})

    @classmethod
    def model(cls, models, device):
        import torch
        enc, dec = cls._plugins(models.common)
        m = models.MultiDMM(cls.mods, cls.dims, cls.dists, encoders=enc, decoders=dec, h_dim=256, z_dim=256,
                            device=device)
        m.sweep_dtype = m.conv_dtype = m.act_dtype = torch.bfloat16
        return m

    @classmethod
    def oracle(cls, orc):
        from mdmm.models import common as C
        enc, dec = cls._plugins(C)
        return orc.OracleDMM(cls.mods, cls.dims, cls.dists, encoders=enc, decoders=dec, h_dim=256, z_dim=256)


class Cfg4F32(Cfg4):
    """cfg4 with every contraction on fp32 operands (BASELINE configs[3] states no dtype)."""
    name, dtype, peak = 'cfg4', 'f32', F32_PEAK_TFLOPS
    workload = ('cfg4 (fp32 operands): Weizmann-shaped synthetic, MultiDKS backward RNN (B-Skip), feat_to_z, conv encoders/decoders, '
                'z=h=256, T=40, B=%d per GPU, 20%% burst NaN; every contraction with fp32 operands (recurrences and projections on '
                'the f32-input MFMA, convolutions in the library), activations fp32')

    @classmethod
    def model(cls, models, device):
        import torch
        m = super().model(models, device)
        m.sweep_dtype = m.conv_dtype = m.act_dtype = torch.float32
        return m


class Cfg5F32(Cfg5):
    """cfg5 with every contraction on fp32 operands (BASELINE configs[4] states no dtype)."""
    name, dtype, peak = 'cfg5', 'f32', F32_PEAK_TFLOPS
    workload = ('cfg5 (fp32 operands): vidTIMIT-shaped synthetic (video 3x64x64 + audio 10x1281, Bernoulli), MultiDMM BFVI, conv '
                'encoders/decoders, z=h=256, T=128, B=%d per GPU, ragged lengths 64..128, 50%% independent missingness, '
                'train_particles=25; every contraction with fp32 operands (sweeps on the f32-input MFMA, image convolutions in the '
                'library, audio stacks on csrc/audio_chain.hip with fp32 activations)')

    @classmethod
    def model(cls, models, device):
        import torch
        m = super().model(models, device)
        m.sweep_dtype = m.conv_dtype = m.act_dtype = torch.float32
        return m


CONFIGS = {'cfg2': Cfg2, 'cfg3': Cfg3, 'cfg4': Cfg4, 'cfg5': Cfg5}


# Step time of the oracle against the UNMODIFIED reference on the same shapes, weights and cores, measured in
# the build container (8 cores, tools/oracle_vs_reference_time.py; the reference cannot travel to the GPU box):
# < 1 means the port is the faster of the two, i.e. `cpu_baseline` is an optimistic stand-in for the reference.
ORACLE_OVER_REFERENCE_TIME = {'cfg2': 0.91, 'cfg3': 0.71}


def step_draws(noise, cfg, k_train, b_dim, device):
    """The eps tensors one MultiDMM.step draws from `noise` (a PhiloxNoise positioned where the step
    started), materialised in the oracle's call order (two prior-matching draws, bfilter pass by pass, then
    per pass the K-particle filter and the smoother): dmm.py:503-554."""
    from mdmm import ops
    d, t_max, p_pass = cfg.D, cfg.T, 1 + cfg.M
    draws = [noise.normal((50, 1, d), device).cpu(), noise.normal((50, 1, d), device).cpu()]
    sweeps = []
    for k in (1, k_train, 1):
        sd, off = noise.stream()
        sweeps.append(ops.philox_normal(sd, off, (p_pass, t_max, k, b_dim, d), device, noise.device_counter(device)).cpu())
    for p in range(p_pass):
        draws += [sweeps[0][p, t] for t in reversed(range(t_max))]
    for p in range(p_pass):
        draws += [sweeps[1][p, t] for t in reversed(range(t_max))]
        draws += [sweeps[2][p, t] for t in range(t_max)]
    return draws


def cpu_baseline(cfg, device=None, seconds_budget=45.0):
    """Oracle ELBO step (fwd + bwd + Adam) on the host cores, bounded sample of the workload; and, with a
    device, the ELBO delta of the HIP path against that oracle on the same batch (BASELINE metric, 2nd half)."""
    import torch
    from oracle import mdmm_oracle as orc
    b_dim = 32 if cfg is Cfg2 else 4
    torch.manual_seed(0)
    model = cfg.oracle(orc)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr)
    inputs, targets, mask, lengths = cfg.batch(cfg.T, b_dim, 1234, 'cpu')

    delta = None
    if device is not None:      # before the timed steps move the weights
        delta = elbo_delta(cfg, model, inputs, targets, mask, lengths, device)

    def one():
        loss = model.step(inputs, mask, 1.0, cfg.rec, targets=targets, lengths=lengths)
        (loss / sum(lengths)).backward()
        opt.step()
        opt.zero_grad()

    # Thousands of small ops per step: more threads is not always faster (on a 128-core host 128 threads
    # were 6x slower than 8).  One warm-up step, then single steps at a few thread counts in the order a
    # good one is likely to come early; 1 thread and every core are always among them.
    all_cores = torch.get_num_threads()
    order = [c for c in (8, 16, 32) if c < all_cores] + [1, all_cores]
    torch.set_num_threads(order[0])
    t_start = time.perf_counter()
    one()
    trial = {}
    for nt in order:
        # (1 thread and all cores are measured even when the budget is gone: they are the contract's two points)
        if nt not in (1, all_cores) and time.perf_counter() - t_start > 0.5 * seconds_budget:
            continue
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        one()
        trial[nt] = time.perf_counter() - t0
    best = min(trial, key=trial.get)
    torch.set_num_threads(best)
    t0, n = time.perf_counter(), 0
    while n < 3 or (n < 8 and time.perf_counter() - t_start < 0.5 * seconds_budget):      # >= 3 steps at the chosen count
        one()
        n += 1
    dt = (time.perf_counter() - t0) / n
    torch.set_num_threads(all_cores)
    ratio = ORACLE_OVER_REFERENCE_TIME.get(cfg.name)
    out = {'value': round(b_dim / dt, 3), 'unit': 'sequences/s', 'cores': best, 'kind': 'port',
           'by_threads': {str(k): round(b_dim / v, 3) for k, v in sorted(trial.items())},
           'oracle_over_reference_step_time': ratio,
           'steps_timed': n,
           'sample': '%d consecutive steps of the same %s step at B=%d at the best thread count of one trial step each '
                     '(sequences/s is flat in B on the CPU, BASELINE.md 2; the full batch would take minutes per step), '
                     'torch-CPU oracle; trial s/step by threads: %s; host has %d cores; best %d threads, %.2f s/step over '
                     'the timed steps.  The oracle takes %s x the unmodified reference\'s step time '
                     '(same shapes and weights, 8 cores of the build container, tools/oracle_vs_reference_time.py)'
                     % (n, cfg.name, b_dim, {k: round(v, 2) for k, v in sorted(trial.items())}, all_cores, best, dt, ratio)}
    return out, delta


def elbo_delta(cfg, oracle, inputs, targets, mask, lengths, device):
    """|loss_hip - loss_oracle| / |loss_oracle| of one ELBO step on the cpu_baseline's batch: same weights,
    the kernels' own Philox draws materialised and replayed into the oracle.  One oracle forward serves every
    precision mode (the draws do not depend on it)."""
    import torch
    from mdmm import models
    from mdmm.noise import PhiloxNoise
    from oracle import mdmm_oracle as orc
    to = lambda d: {k: v.to(device) for k, v in d.items()}      # noqa: E731
    b_dim = len(lengths)
    kw = dict(train_particles=TRAIN_PARTICLES)
    modes = {'f32': torch.float32} if cfg is Cfg2 else {'bf16': torch.bfloat16, 'f32': torch.float32}
    hip = {}
    for name, dt in modes.items():
        m = cfg.model(models, device)
        if cfg is not Cfg2:
            m.sweep_dtype = m.conv_dtype = m.act_dtype = dt
        m.load_state_dict(oracle.state_dict())
        m.noise = PhiloxNoise(seed=777)
        hip[name] = float(m.step(to(inputs), mask.to(device), 1.0, cfg.rec, targets=to(targets), lengths=lengths, **kw))
        del m
    own_noise = oracle.noise
    oracle.noise = orc.ReplayNoise(step_draws(PhiloxNoise(seed=777), cfg, TRAIN_PARTICLES, b_dim, device))
    was_training = oracle.training
    oracle.train()
    with torch.no_grad():
        ref = float(oracle.step(inputs, mask, 1.0, cfg.rec, targets=targets, lengths=lengths, **kw))
    oracle.train(was_training)
    oracle.noise = own_noise
    return {'rel': {k: round(abs(v - ref) / abs(ref), 9) for k, v in hip.items()}, 'loss_oracle': ref,
            'loss_hip': hip, 'sample': 'loss of one %s step at B=%d, same weights, the kernels\' Philox noise replayed '
                                       'into the CPU oracle (north star: 1e-4 relative)' % (cfg.name, b_dim)}


HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md; 6.3 TB/s is the measured copy rate)


def sweep_bytes(cfg, tag, b_dim):
    """Algorithmic HBM bytes of one sweep launch: per (pass, sequence, t) a forward sweep reads
    2 M_p expert tensors (+ 2 of the filter pass and nothing else for the smoother's extra experts) and writes
    4 or 5 outputs of D floats; backward = 2 x."""
    m_p = [cfg.M] + [1] * cfg.M if cfg.M > 1 else [1]                  # modalities present per pass
    k25 = 'K=%d' % TRAIN_PARTICLES in tag
    smoother = ',inv' in tag
    per = sum(2 * m + (7 if smoother else (2 if k25 else 5)) for m in m_p)
    fwd = per * cfg.T * cfg.D * 4 * b_dim
    return 2 * fwd if 'bwd' in tag else fwd


def load_traffic(tag, cfg, b_dim):
    """HBM bytes per launch from the PMC passes kept under profiles/ (tools/pmc_traffic.sh; counters cannot be
    collected from inside the benchmark process) -- only when the record was taken at THIS shape."""
    for name in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json'):
        path = os.path.join(REPO, 'profiles', name)
        if not os.path.exists(path):
            continue
        try:
            table = json.load(open(path))
        except ValueError:
            continue
        for key, rec in table.items():
            key = rec.get('key', key)               # (r04: one file for several shapes, the call's tag under 'key')
            shape = rec.get('shape', {'B': 256, 'T': 40, 'P': 4} if name.startswith('r02') else None)
            if key.startswith('sweep_wide') and cfg.dtype != 'bf16':
                continue                            # (the wide records were taken with bf16 operands)
            if key in tag and shape == {'B': b_dim, 'T': cfg.T, 'P': 1 + cfg.M}:
                return rec.get('bytes_per_launch'), rec.get('source')
    return None, None


def roofline_of(cfg, spans, b_dim, want_k1=False):
    """The sweep with the largest device time (or, want_k1, the largest K = 1 sweep: the kernel the north
    star's HBM target names) against both roofs: `bound` = the roof that takes longer for this launch, measured
    HBM traffic standing in for the algorithmic bytes where a PMC record of this shape exists."""
    cand = {t: v for t, v in spans.items() if t.startswith('sweep_') and (not want_k1 or 'K=1,' in t or 'K=1]' in t)}
    if not cand:
        return None
    tag, (n_launch, tot_ms) = max(cand.items(), key=lambda kv: kv[1][1])
    note = None
    if want_k1:
        # Both K = 1 backward sweeps of a step are the same kernel on the same shape (the filtering-mode one and the
        # smoother's); in the eager probe they run on different streams next to the K-particle backward, whose 256
        # workgroups take every CU: the HIP-event span of the one that queues behind it is 3-4 x its device time.
        # Take the one with the shorter span -- the one rocprofv3's per-kernel average agrees with.
        bwd = {t: v for t, v in cand.items() if 'bwd' in t}
        if len(bwd) > 1:
            tag, (n_launch, tot_ms) = min(bwd.items(), key=lambda kv: kv[1][1] / kv[1][0])
            note = 'the K = 1 backward call with the shorter HIP-event span of the step\'s two (the other one queues behind the K-particle backward on its stream: %s)' % (
                ', '.join('%s %.2f ms' % (t, v[1] / v[0]) for t, v in sorted(bwd.items())))
    p_pass = 1 + cfg.M
    k = TRAIN_PARTICLES if 'K=%d' % TRAIN_PARTICLES in tag else 1
    rows = p_pass * b_dim * (cfg.T - 1) * k          # transition rows of one sweep launch
    fwd = rows * gtf_flops(cfg.D, cfg.H)
    if 'bwd' in tag:
        # backward = 2 x forward (input + weight gradients); the kernels also recompute
        # the forward transition (executed = 3 x)
        flops, executed = 2 * fwd, 3 * fwd
    else:
        flops = executed = fwd
    avg_s = tot_ms / n_launch * 1e-3
    wide = tag.startswith('sweep_wide') and cfg.dtype == 'bf16'
    peak = BF16_PEAK_TFLOPS if wide else F32_PEAK_TFLOPS
    nbytes = sweep_bytes(cfg, tag, b_dim)
    traffic, source = load_traffic(tag, cfg, b_dim)
    # T_roof = max(algorithmic bytes / HBM peak, algorithmic flops / matrix peak); `bound` names the
    # larger term and `frac` = T_roof / T_measured.  The COUNTER view (measured HBM traffic instead of the
    # algorithmic bytes) is reported beside it as `traffic_bound`; it never decides the headline fields.
    t_mfma = flops / (peak * 1e12)
    t_hbm = nbytes / (HBM_PEAK_GBS * 1e9)
    bound = 'mfma' if t_mfma >= t_hbm else 'hbm'
    tf, gbs = flops / avg_s / 1e12, nbytes / avg_s / 1e9
    rf = {'bound': bound, 'kernel': tag}
    if traffic:
        t_traffic = traffic / (HBM_PEAK_GBS * 1e9)
        rf['traffic_bound'] = {'bound': 'mfma' if t_mfma >= t_traffic else 'hbm', 'hbm_time_ms': round(t_traffic * 1e3, 4),
                               'frac_of_launch': round(t_traffic / avg_s, 4)}
    if bound == 'mfma':
        rf.update({'achieved': round(tf, 3), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(tf / peak, 4)})
    else:
        rf.update({'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4)})
    rf.update({'traffic': traffic, 'launch_ms': round(avg_s * 1e3, 4), 'launches': n_launch,
               'mfma': {'achieved_tflops': round(tf, 3), 'peak_tflops': peak, 'frac': round(tf / peak, 4),
                        'flops_per_launch': flops, 'executed_flops': executed, 'operands': 'bf16' if wide else 'f32'},
               'hbm': {'algorithmic_bytes': nbytes, 'achieved_gbs': round(gbs, 1), 'frac': round(gbs / HBM_PEAK_GBS, 4),
                       'traffic_gbs': round(traffic / avg_s / 1e9, 1) if traffic else None,
                       'traffic_frac': round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                       'traffic_over_algorithmic': round(traffic / nbytes, 2) if traffic else None},
               'roof_times_ms': {'mfma': round(t_mfma * 1e3, 4), 'hbm': round(t_hbm * 1e3, 4)}})
    if source:
        rf['traffic_source'] = source
    if note:
        rf['choice'] = note
    if want_k1 and wide and cfg.D == 256:
        rf['design_bound'] = k1_design_bound(cfg, tag, b_dim, nbytes)
    return rf


# The ceiling of the K = 1 wide sweeps AS DESIGNED: a workgroup re-streams its direction's weight fragments
# from L2 once per time step -- 768 KB forward, 1.5 MB backward (bf16) -- through a CU port that delivers 61 B per clock
# (tools/mb/l2_stream.hip, profiles/r04z_l2_stream.txt: 5.4 us per 768 KB pass, the same with 1 ... 128 workgroups), and
# a launch is T - 1 dependent steps: no batch size makes a launch shorter than (T - 1) x 5.4 us (10.8 us backward); the
# weight-gradient contraction and the reduction behind a backward sweep are HBM streams of the spilled operands.
K1_STREAM_US = {'fwd': 5.4, 'bwd': 10.8}


def k1_design_bound(cfg, tag, b_dim, nbytes):
    which = 'bwd' if 'bwd' in tag else 'fwd'
    pairs = (1 + cfg.M) * b_dim
    n_wg = (pairs + 7) // 8 if pairs <= 8 * 256 else ((pairs + 15) // 16 if pairs <= 16 * 256 else (pairs + 31) // 32)
    rounds = (n_wg + 255) // 256                       # workgroups hold a CU's LDS: one per CU at a time
    t_stream = rounds * (cfg.T - 1) * K1_STREAM_US[which] * 1e-6
    frac = nbytes / t_stream / 1e9 / HBM_PEAK_GBS
    return {'frac': round(frac, 4), 'launch_ms_floor': round(t_stream * 1e3, 4), 'workgroups': n_wg, 'rounds': rounds,
            'basis': 'T - 1 = %d dependent steps x %.1f us (one pass of the %s weight fragments through a CU\'s 61 B/clk L2 '
                     'port, DESIGN 4.2e) per round of workgroups; algorithmic bytes of the launch over that time against the '
                     'HBM peak.  The weight-gradient contraction and reduction that the measured call also contains are not '
                     'in the floor.  North-star target 0.40: needs >= 2 x these bytes per step at the same step time '
                     '(32-pair tiles, a batch of >= 8192 pairs).' % (cfg.T - 1, K1_STREAM_US[which], 'backward' if which == 'bwd' else 'forward')}


def roofline_bytes(timer, spans):
    """For steps whose heavy calls are streaming kernels (cfg4: BatchNorm / conv chain): the call with the
    largest device time among those that state their algorithmic HBM bytes (ops._call(nbytes=...))."""
    cand = {t: spans[t] for t in getattr(timer, 'nbytes', {}) if t in spans}
    if not cand:
        return None
    tag, (n_launch, tot_ms) = max(cand.items(), key=lambda kv: kv[1][1])
    nbytes = timer.nbytes[tag]
    gbs = nbytes / (tot_ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernel': tag, 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': None, 'launches': n_launch,
            'launch_ms': round(tot_ms / n_launch, 4), 'algorithmic_bytes_per_launch': nbytes // n_launch}


# Whole-step algorithmic work (SURVEY 8d "whole-step context", counted for what the step's arithmetic NEEDS, not for
# what the reference re-executes): forward FLOPs per frame of the plug-ins (torch flop counter on the reference's
# modules) -- every modality encoded once, decoded once per pass that scores it (multimodal pass + its own
# unimodal pass, in both modes: 4 decodes, dgts.py:119-129) --, the sweeps' 3.31 GFLOP per sequence; backward = 2 x.
# Bytes: inputs and targets read once (fp32 frames), the sweeps' SURVEY figure; activations between fused kernels are
# not algorithmic.
PLUGIN_FWD_MFLOP = {'cfg3': {'enc': 9.80 + 9.21, 'dec': 12.06 + 11.01, 'decodes': 4}}


def roofline_step(cfg, b_dim, ms_per_step):
    if cfg.name not in ('cfg2', 'cfg3'):
        return None
    p_pass = 1 + cfg.M
    sweep_fwd = p_pass * 27 * (cfg.T - 1) * gtf_flops(cfg.D, cfg.H) * b_dim                 # P * 27 * (T - 1) * F_GTF
    m_p = [cfg.M] + [1] * cfg.M if cfg.M > 1 else [1]
    sweep_bytes_fwd = sum(6 * m + 14 for m in m_p) * cfg.T * cfg.D * 4 * b_dim
    flops, nbytes = 3 * sweep_fwd, 3 * sweep_bytes_fwd
    parts = {'sweeps_tflop': round(3 * sweep_fwd / 1e12, 3)}
    if cfg.name in PLUGIN_FWD_MFLOP:
        pl = PLUGIN_FWD_MFLOP[cfg.name]
        conv_fwd = (pl['enc'] + pl['decodes'] * pl['dec']) * 1e6 * cfg.T * b_dim
        flops += 3 * conv_fwd
        frame_bytes = sum(int(__import__('math').prod(d)) if isinstance(d, tuple) else 1 for d in cfg.dims) * 4
        nbytes += 2 * frame_bytes * cfg.T * b_dim                                             # inputs + targets, read once
        parts['plugins_tflop'] = round(3 * conv_fwd / 1e12, 3)
    peak = cfg.peak
    t_mfma, t_hbm = flops / (peak * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
    t_roof = max(t_mfma, t_hbm)
    out = {'bound': 'mfma' if t_mfma >= t_hbm else 'hbm', 'algorithmic_flops': flops, 'algorithmic_bytes': nbytes,
           'roof_times_ms': {'mfma': round(t_mfma * 1e3, 4), 'hbm': round(t_hbm * 1e3, 4)},
           'achieved_tflops': round(flops / (ms_per_step * 1e-3) / 1e12, 2), 'peak_tflops': peak,
           'frac': round(t_roof * 1e3 / ms_per_step, 4), 'parts': parts,
           'note': 'whole step: algorithmic flops and bytes (sweeps + plug-in conv chain, backward = 2 x forward) '
                   'over ms_per_step; frac = T_roof / T_step'}
    # what the step MOVES, measured (every kernel's FETCH_SIZE / WRITE_SIZE: tools/pmc_step_traffic.sh): the step as a whole
    # against the HBM peak -- its kernels are the conv chain's activations and the K-particle sweeps' parks, not its flops
    tr = step_traffic(cfg, b_dim)
    if tr is not None:
        gbs = tr['bytes_per_step'] / (ms_per_step * 1e-3) / 1e9
        out['traffic'] = tr['bytes_per_step']
        out['traffic_gbs'] = round(gbs, 1)
        out['traffic_frac_of_hbm_peak'] = round(gbs / HBM_PEAK_GBS, 4)
        out['traffic_over_algorithmic'] = round(tr['bytes_per_step'] / nbytes, 2)
        out['traffic_source'] = tr['source']
    return out


def step_traffic(cfg, b_dim):
    """Measured HBM-side bytes of one step (profiles/r05t_step_traffic.json), for the shape they were measured at."""
    if cfg.name != 'cfg3' or b_dim != cfg.B:
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r05t_step_traffic.json')
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


GRAPH_QUEUES_ENV, GRAPH_QUEUES = 'DEBUG_HIP_FORCE_GRAPH_QUEUES', '5'


ANNEAL = False        # tools: a KLD multiplier that changes with every call of the replayed step


def run(cfg, args, world, rank, device, graph):
    """Time args.steps steps of cfg on this rank; returns the result dict (rank 0) or None."""
    import torch
    import torch.distributed as dist
    from mdmm import models, ops
    from mdmm.harness import GradBucket, GraphedElboStep, elbo_step
    from mdmm.noise import PhiloxNoise

    torch.manual_seed(0)                    # identical weights on every rank
    model = cfg.model(models, device)
    model.noise = PhiloxNoise(seed=1000 + rank)
    bucket = GradBucket(model.parameters())
    if getattr(args, 'torch_adam', False):      # (A/B: the framework's fused multi-tensor Adam, three launches)
        optimizer = torch.optim.Adam(model.parameters(), lr=cfg.lr, capturable=graph, fused=True)
    else:                                       # torch.optim.Adam's update as one launch over the flat bucket
        from mdmm.harness import FlatAdam
        optimizer = FlatAdam(bucket, lr=cfg.lr)
    b_dim = args.batch or cfg.B
    inputs, targets, mask, lengths = cfg.batch(cfg.T, b_dim, 1234 + rank, device)
    n_points_global = sum(lengths) * world
    kw = dict(targets=targets, n_points_global=n_points_global, train_particles=TRAIN_PARTICLES)

    def eager_step():
        return elbo_step(model, optimizer, bucket, inputs, mask, lengths, 1.0, cfg.rec, **kw)

    execution = 'eager'
    step = eager_step
    c_capture = None
    if graph:       # same step, captured once into two HIP graphs (collective in between, eager)
        try:
            c0, warm = model.noise.counter, 3
            graphed = GraphedElboStep(model, optimizer, bucket, inputs, mask, lengths, 1.0, cfg.rec, warmup=warm, **kw)

            calls = [0]

            def step():     # as a trainer drives it (trainer.py:226-244): this batch's KLD multiplier and number of
                calls[0] += 1                                                # time-points go to the device, then the replay
                kld = 1.0 - 1e-3 * (calls[0] % 7) if ANNEAL else 1.0         # (ANNEAL: tools/bench_one_extra.py)
                graphed.schedule(kld_mult=kld, n_points=n_points_global)
                return graphed()
            step.g_step, step.loss = graphed.g_step, graphed.loss
            c_capture = model.noise.counter - (model.noise.counter - c0) // (warm + 1)   # host stream id the capture starts at
            execution = 'hipgraph, schedule scalars on the device, executor default queues'
            if os.environ.get(GRAPH_QUEUES_ENV):
                execution = execution.replace('executor default queues', '%s executor queues' % os.environ[GRAPH_QUEUES_ENV])
            execution += ', DEBUG_CLR_GRAPH_PACKET_CAPTURE=%s (runtime workaround, mdmm/__init__.py)' % os.environ.get(
                'DEBUG_CLR_GRAPH_PACKET_CAPTURE')
        except Exception as exc:        # noqa: BLE001 -- never lose the run to a capture problem
            print('bench: HIP-graph capture failed (%r); running the step eagerly' % (exc,),
                  file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            bucket.release()
            execution = 'eager (graph capture failed)'

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    if not execution.startswith('hipgraph'):
        ops.TIMER = ops.KernelTimer()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # The execution mode `value` was timed in, checked against the plain one: one more replay of the captured
    # step against the same step run eagerly on the same weights and the same Philox stream.
    replay_check = None
    if c_capture is not None and execution.startswith('hipgraph'):
        noise = model.noise
        d0 = noise.device_counter(device).clone()
        step.g_step.replay()
        torch.cuda.synchronize()
        loss_r, flat_r = float(step.loss), bucket.flat.clone()
        noise.counter = c_capture
        noise.device_counter(device).copy_(d0)
        bucket.release()
        l_e = model.step(inputs, mask, 1.0, cfg.rec, targets=targets, lengths=lengths, train_particles=TRAIN_PARTICLES)
        (l_e / n_points_global).backward()
        bucket.check_views()
        torch.cuda.synchronize()
        loss_e, flat_e = float(l_e), bucket.flat
        g_rel = float((flat_r - flat_e).norm() / (flat_e.norm() + 1e-30))
        l_rel = abs(loss_r - loss_e) / abs(loss_e)
        replay_check = {'ok': bool(l_rel < 1e-5 and g_rel < 1e-4), 'loss_rel': l_rel, 'grad_l2_rel': g_rel}
        bucket.release()
        del flat_r
    n_probe = args.steps
    timing_note = 'HIP events on the launch stream around every library call of the timed steps'
    loss_val = float(loss)
    if timer is None and graph:
        # the eager probe steps below allocate a step's worth of memory of their own: the graphs' private pool goes first
        # (cfg5 at its per-GPU batch: 133 GB of pool + as much again did not fit the 288 GB)
        import gc
        graphed = step = loss = None
        gc.collect()
        torch.cuda.empty_cache()
    if timer is None:
        # Graph replay leaves no place for events between nodes: re-run the same step eagerly
        # (not part of `value`) with HIP events around every library call.  Every rank takes the
        # steps (they hold the gradient all-reduce); rank 0 keeps the timings.
        n_probe = min(args.steps, 3)
        if rank == 0:
            ops.TIMER = ops.KernelTimer()
        for _ in range(n_probe):
            eager_step()
        torch.cuda.synchronize()
        timer, ops.TIMER = ops.TIMER, None
        timing_note = ('HIP events around every library call of %d eager re-runs of the same step, '
                       'right after the graph-replayed timed region' % n_probe)
    # N > 1: what the one collective of a step costs by itself -- HIP events around all_reduce(flat gradient) behind a
    # barrier, outside the timed region (in the timed step it sits between the two graphs, after a host wait)
    allreduce_ms = None
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for e0, e1 in ev:
            e0.record()
            dist.all_reduce(bucket.flat)
            e1.record()
        torch.cuda.synchronize()
        allreduce_ms = round(sorted(e0.elapsed_time(e1) for e0, e1 in ev)[2], 4)
        bucket.release()
        dist.barrier()
    if rank != 0:
        return None
    spans = timer.summary()
    # an event pair around a call also measures what the launch path adds between the two records: calibrate on
    # empty pairs and take it off every call (it is what made 171 small column-sum launches look like 3 ms)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for e0, e1 in pairs:
        e0.record(); e1.record()
    torch.cuda.synchronize()
    pair_ms = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)[100]
    rf = roofline_of(cfg, spans, b_dim)
    rf_k1 = roofline_of(cfg, spans, b_dim, want_k1=True)
    if rf is None or cfg is Cfg4:   # (cfg4: the sweeps' model does not apply -- its recurrences are latency chains;
        rf = roofline_bytes(timer, spans) or rf      #  what fills its step is the streaming BatchNorm / conv chain)
    # the conv family (the largest of the cfg3 step): its weight-gradient call with the most device time, algorithmic bytes
    # (both activation sides read once, + the BatchNorm adjoint's gradient where the kernel carries its sums) over its
    # HIP-event time against the HBM peak
    conv_spans = {t: v for t, v in spans.items() if t.startswith('conv_wgrad')}
    rf_conv = roofline_bytes(timer, conv_spans) if conv_spans else None
    # the audio plug-ins' launches (cfg5; csrc/audio_chain.hip): the one with the most device time, its algorithmic bytes
    # (frames, activations and gradients it has to read / write once; target rows nobody scores are not read) over its time
    audio_spans = {t: v for t, v in spans.items() if t.startswith('audio_')}
    rf_audio = roofline_bytes(timer, audio_spans) if audio_spans else None
    for r in (rf, rf_k1, rf_conv, rf_audio):
        if r is not None:
            r['timing'] = timing_note
    out_cfg = {'workload': cfg.workload % b_dim, 'global_batch': world * b_dim, 'seq_len': cfg.T,
               'parallelism': 'dp%d' % world, 'loss': round(loss_val, 3), 'execution': execution,
               'rccl_ranks': dist.get_world_size() if world > 1 and dist.is_initialized() else 1,
               # BatchNorm statistics of the conv plug-ins (common.py:80-84): over this rank's frames only -- the graph-
               # replayed step cannot hold the synchronising collective (harness.GraphedElboStep); 10,240 frames per rank
               'bn': 'per-rank' if any('net.1' in k for k, _ in model.named_parameters()) else 'none',
               'optimizer': type(optimizer).__name__ + (' (torch.optim.Adam\'s update as one launch over the flat gradient bucket, mdmm_adam_flat)'
                                                         if type(optimizer).__name__ == 'FlatAdam' else ' (fused, capturable)'),
               'allreduce': None if allreduce_ms is None else
               {'ms': allreduce_ms, 'bytes': int(bucket.flat.numel() * 4), 'timing': 'HIP events, median of 5, outside the timed region',
                'host_wait_before': True}}
    if replay_check is not None:
        out_cfg['replay_matches_eager'] = replay_check
    return {
        'metric': 'sequences/sec (ELBO step)', 'value': round(world * b_dim * args.steps / elapsed, 2),
        'unit': 'sequences/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': cfg.dtype, 'data': 'synthetic',
        'config': out_cfg,
        'roofline': rf,
        'roofline_k1': rf_k1,
        'roofline_conv': rf_conv,
        'roofline_audio': rf_audio,
        'roofline_step': roofline_step(cfg, b_dim, 1e3 * elapsed / args.steps),
        # library calls by device time per step: HIP-event spans minus the calibrated cost of an empty event pair
        # per call (eager probe steps; per-KERNEL device times: profiles/*_kernel_stats.md from rocprofv3)
        'calls_ms_per_step': {t_: round(max(v[1] - v[0] * pair_ms, 0.0) / n_probe, 4) for t_, v in
                              sorted(spans.items(), key=lambda kv: -(kv[1][1] - kv[1][0] * pair_ms))[:16]},
        'event_pair_overhead_us': round(pair_ms * 1e3, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='cfg3')
    ap.add_argument('--batch', type=int, default=0, help='sequences per GPU (default: the config\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the cfg2 / cfg4 / cfg5 / cfg3_f32 lines that ride along')
    ap.add_argument('--eager', action='store_true', help='no HIP-graph replay of the step')
    ap.add_argument('--torch-adam', action='store_true', help='torch.optim.Adam(fused=True) instead of harness.FlatAdam')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.steps is None:
        args.steps = 5 if cfg is Cfg3 else 10
    if args.warmup is None:
        args.warmup = 2 if cfg is Cfg3 else 3

    # Round 2-3 asked the HIP graph executor for five streams (DEBUG_HIP_FORCE_GRAPH_QUEUES=5: 46.3 -> 44.3 ms then).  With
    # the runtime's graph packet capture switched off (the replay workaround above) the default is as fast
    # (30.24 vs 30.25 ms, profiles/r04k_packet_capture.txt; 6 / 8 / 12 queues: 30.14 / 30.14 / 30.16 and no crash any more),
    # so nothing is forced; an exported value is honoured and named in config.execution.
    ours = False
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        # start the ranks ourselves, BEFORE anything in this process touches the GPU
        port = os.environ.get('MASTER_PORT', str(29500 + os.getpid() % 2000))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
        def launch(env, timeout):
            # own process group: a hung attempt is ended with every rank it started
            p = subprocess.Popen(cmd, env=env, start_new_session=True)
            try:
                return p.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                return -1

        # (a healthy run takes two minutes: a hang counts as a failure of the first attempt)
        rc = launch(None, 900 if ours else None)
        if rc != 0 and ours:        # never lose the run to the executor setting: once more with its default
            print('bench: ranks failed with %s=%s; retrying with the runtime default' % (GRAPH_QUEUES_ENV, GRAPH_QUEUES),
                  file=sys.stderr, flush=True)
            env = dict(os.environ)
            env.pop(GRAPH_QUEUES_ENV)
            env['MDMM_BENCH_DEFAULT_QUEUES'] = '1'      # (the ranks would set it again)
            env['MASTER_PORT'] = str(int(port) + 1)
            cmd[cmd.index('--master-port') + 1] = env['MASTER_PORT']
            rc = launch(env, None)
        raise SystemExit(rc)
    world = int(env_world or '1')
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks'
                         % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the MDMM hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    # both configurations replay the step from HIP graphs (every kernel of the cfg3 step is the
    # library's own or a capturable torch op since the conv pyramids left MIOpen); --eager opts out
    out = run(cfg, args, world, rank, device, graph=not args.eager)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.empty_cache()
            try:
                out['cpu_baseline'], out['elbo_delta'] = cpu_baseline(cfg, device)
            except Exception as exc:            # noqa: BLE001 -- the timed line is printed whatever the host leg does
                out['cpu_baseline'] = {'value': None, 'unit': 'sequences/s', 'cores': 0, 'kind': 'port', 'sample': 'failed',
                                       'error': '%s: %s' % (type(exc).__name__, str(exc)[:300])}
                print('bench: cpu_baseline failed: %r' % (exc,), file=sys.stderr, flush=True)
        if world == 1 and cfg is Cfg3 and not args.no_extra:
            out['extra'] = {}

            def ride_along(key, make):
                """One ride-along measurement.  Whatever goes wrong in it (an out-of-memory in an fp32 sibling, a library
                search that fails) is recorded under its key: the line of the timed configuration is printed regardless."""
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                try:
                    out['extra'][key] = make()
                except Exception as exc:        # noqa: BLE001
                    torch.cuda.synchronize()
                    out['extra'][key] = {'error': '%s: %s' % (type(exc).__name__, str(exc)[:400])}
                    print('bench: extra %r failed: %r' % (key, exc), file=sys.stderr, flush=True)
                gc.collect()
                torch.cuda.empty_cache()

            def timed(c, steps, warmup, graph, keys, batch=0):
                a_ = argparse.Namespace(**vars(args))
                a_.steps, a_.warmup, a_.batch = steps, warmup, batch
                r_ = run(c, a_, 1, 0, device, graph=graph)
                return {k: r_[k] for k in keys if k in r_}

            ride_along('cfg2', lambda: timed(Cfg2, 10, 3, True, ('value', 'unit', 'ms_per_step', 'dtype', 'config', 'roofline',
                                                                 'roofline_k1', 'roofline_step')))
            ride_along('cfg4', lambda: timed(Cfg4, 5, 2, not args.eager, ('value', 'unit', 'ms_per_step', 'dtype', 'config',
                                                                          'roofline', 'calls_ms_per_step')))
            # BASELINE configs[4] at its per-GPU size (512 sequences, T = 128; video + audio plug-ins)
            ride_along('cfg5', lambda: timed(Cfg5, 5, 2, not args.eager, ('value', 'unit', 'ms_per_step', 'dtype', 'config', 'roofline',
                                                                          'roofline_k1', 'roofline_conv', 'roofline_audio',
                                                                          'calls_ms_per_step')))

            # the same cfg3 step with fp32 operands everywhere (the mode whose parity tests hold 1e-5): library
            # convolutions, own fp32-operand sweeps; eager (the library's convolutions are not captured)
            def cfg3_f32():
                r = timed(Cfg3F32, 3, 3, False, ('value', 'unit', 'ms_per_step', 'dtype', 'config', 'roofline'))
                # ... and with the convolutions on the own fp32-operand path too (no library kernel in the step)
                a_ = argparse.Namespace(**vars(args))
                a_.steps, a_.warmup, a_.batch = 3, 2, 0
                rown = run(Cfg3F32Own, a_, 1, 0, device, graph=False)
                r['own_convolutions'] = {'value': rown['value'], 'ms_per_step': rown['ms_per_step'],
                                         'loss': rown['config'].get('loss')}
                return r
            ride_along('cfg3_f32', cfg3_f32)
            # cfg4 / cfg5 with fp32 operands too (their BASELINE strings state no dtype): eager, two timed steps each behind
            # TWO warm-up steps (with one, the library's search for the cfg5 image convolutions' backward kernels was still
            # running in the timed steps: 3,951 ms per step recorded where the steady state is 632, profiles/r06n_cfg5_f32_spans.txt).
            # cfg5 at HALF its per-GPU batch: fp32 activations of 512 sequences x 128 steps are 211 GB of live tensors in an
            # eager step, and a third step on the allocator's fragments did not fit the 288 GB (profiles/r06v: out of memory);
            # sequences/s is what the line reports, the batch is in its config.
            ride_along('cfg4_f32', lambda: timed(Cfg4F32, 2, 2, False, ('value', 'unit', 'ms_per_step', 'dtype', 'config')))
            ride_along('cfg5_f32', lambda: timed(Cfg5F32, 2, 2, False, ('value', 'unit', 'ms_per_step', 'dtype', 'config'),
                                                 batch=Cfg5.B // 2))

            # the callers either side of the step: on-device batch preparation and the evaluation body
            def callers(which):
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
                import bench_callers
                return getattr(bench_callers, which)(device)
            ride_along('batch_prep', lambda: callers('measure_batch_prep'))
            ride_along('eval', lambda: callers('measure_eval'))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
