"""GPU box: time one ELBO step of the other BASELINE configs (shape-only, synthetic, reduced
batch where noted).  Not the contract bench (bench.py = cfg2); evidence for DESIGN.md."""
import json, os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import models, ops
from mdmm.harness import GradBucket, elbo_step
from mdmm.noise import PhiloxNoise

dev = torch.device('cuda:0')
C = models.common


def timed(step, n=3, warm=1):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    probe = getattr(step, 'eager', None)        # graph replay: time it, then per-call events on eager re-runs
    if probe is None:
        ops.TIMER = ops.KernelTimer()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    if probe is not None:
        ops.TIMER = ops.KernelTimer()
        for _ in range(n):
            probe()
        torch.cuda.synchronize()
    spans, ops.TIMER = ops.TIMER.summary(), None
    return dt, {k: round(v[1] / n, 3) for k, v in sorted(spans.items(), key=lambda kv: -kv[1][1])[:8]}


def weizmann(kind, B, T=40):
    torch.manual_seed(0)
    mods, dims = ['video', 'mask', 'action'], [(3, 64, 64), (1, 64, 64), 10]
    dists = ['Bernoulli', 'Bernoulli', 'Categorical']
    if kind == 'dmm':
        m = models.MultiDMM(mods, dims, dists,
                            encoders={'video': C.ImageEncoder(256, n_channels=3), 'mask': C.ImageEncoder(256, n_channels=1)},
                            decoders={'video': C.ImageDecoder(256, n_channels=3), 'mask': C.ImageDecoder(256, n_channels=1)},
                            h_dim=256, z_dim=256, device=dev)
    else:
        m = models.MultiDKS(mods, dims, dists,
                            encoders={'video': C.ImageEncoder(256, gauss_out=False, n_channels=3),
                                      'mask': C.ImageEncoder(256, gauss_out=False, n_channels=1)},
                            decoders={'video': C.ImageDecoder(256, n_channels=3), 'mask': C.ImageDecoder(256, n_channels=1)},
                            h_dim=256, z_dim=256, feat_to_z=True, rnn_dir='bwd', rnn_skip=True, device=dev)
    m.noise = PhiloxNoise(seed=1)
    if os.environ.get('SWEEP_BF16', '1') == '1':     # the headline mode of bench.py (bf16 operands)
        m.sweep_dtype = torch.bfloat16
        m.conv_dtype = torch.bfloat16
        m.act_dtype = torch.bfloat16 if os.environ.get('ACT_BF16', '1') == '1' else torch.float32
    if os.environ.get('AMP') == '1':
        m.plugin_dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1234)
    tg = {'video': torch.rand(T, B, 3, 64, 64, generator=g).to(dev),
          'mask': (torch.rand(T, B, 1, 64, 64, generator=g) < 0.5).float().to(dev),
          'action': torch.randint(0, 10, (1, B, 1), generator=g).float().expand(T, B, 1).contiguous().to(dev)}
    x = {k: v.clone() for k, v in tg.items()}
    burst = T // 5
    for k in x:
        st = torch.randint(0, T - burst + 1, (B,), generator=g)
        for b in range(B):
            x[k][st[b]:st[b] + burst, b] = float('nan')
    mask = torch.ones(T, B, 1, dtype=torch.bool, device=dev)
    graph = os.environ.get('GRAPH', '1') == '1'
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, capturable=graph, fused=True)
    bucket = GradBucket(m.parameters())
    rec = {'video': 1.0, 'mask': 1.0, 'action': 10.0}
    eager = lambda: elbo_step(m, opt, bucket, x, mask, [T] * B, 1.0, rec, targets=tg)     # noqa: E731
    if not graph:
        return eager
    from mdmm.harness import GraphedElboStep
    step = GraphedElboStep(m, opt, bucket, x, mask, [T] * B, 1.0, rec, targets=tg)
    step.eager = eager          # per-call HIP-event timing needs eager launches (timed())
    return step


def spirals_cfg1(B=25, T=100, graph=True):
    """cfg1: the reference's own CPU-runnable case (Spirals, z=5, h=20, B=25, T=100, 25 particles)."""
    from mdmm.harness import GraphedElboStep
    torch.manual_seed(0)
    m = models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=20, z_dim=5, device=dev)
    m.noise = PhiloxNoise(seed=1)
    g = torch.Generator().manual_seed(1234)
    tg = {k: torch.randn(T, B, 1, generator=g).to(dev) for k in ('spiral-x', 'spiral-y')}
    x = {k: v.clone() for k, v in tg.items()}
    for k in x:
        st = torch.randint(0, T - 10 + 1, (B,), generator=g)
        for b in range(B):
            x[k][st[b]:st[b] + 10, b] = float('nan')
    mask = torch.ones(T, B, 1, dtype=torch.bool, device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=graph)
    bucket = GradBucket(m.parameters())
    rec = {'spiral-x': .5, 'spiral-y': .5}
    if graph:
        return GraphedElboStep(m, opt, bucket, x, mask, [T] * B, 1.0, rec, targets=tg, train_particles=25)
    return lambda: elbo_step(m, opt, bucket, x, mask, [T] * B, 1.0, rec, targets=tg)


def vidtimit(B, T=128):
    """cfg5 shapes: video (3,64,64) + audio (10,1281), both Bernoulli, each (t,b,modality) missing
    with p = 0.5, ragged lengths U{64..128} sorted descending, z = h = 256."""
    torch.manual_seed(0)
    mods, dims = ['video', 'audio'], [(3, 64, 64), (10, 1281)]
    m = models.MultiDMM(mods, dims, ['Bernoulli', 'Bernoulli'],
                        encoders={'video': C.ImageEncoder(256), 'audio': C.AudioEncoder(256)},
                        decoders={'video': C.ImageDecoder(256), 'audio': C.AudioDecoder(256)},
                        h_dim=256, z_dim=256, device=dev)
    m.noise = PhiloxNoise(seed=1)
    if os.environ.get('SWEEP_BF16', '1') == '1':     # as bench.py's cfg3: bf16 operands / bf16-stored conv activations
        m.sweep_dtype = torch.bfloat16
        m.conv_dtype = torch.bfloat16
        m.act_dtype = torch.bfloat16 if os.environ.get('ACT_BF16', '1') == '1' else torch.float32
    g = torch.Generator().manual_seed(1234)
    lengths = sorted(torch.randint(T // 2, T + 1, (B,), generator=g).tolist(), reverse=True)
    lengths[0] = T
    tg = {'video': torch.rand(T, B, 3, 64, 64, generator=g).to(dev),
          'audio': torch.rand(T, B, 10, 1281, generator=g).to(dev)}
    mask = torch.zeros(T, B, 1, dtype=torch.bool)
    for b, n in enumerate(lengths):
        mask[:n, b] = True
    for k in tg:
        tg[k][~mask.squeeze(-1).to(dev)] = float('nan')
    x = {k: v.clone() for k, v in tg.items()}
    for k in x:
        x[k][(torch.rand(T, B, generator=g) < 0.5).to(dev)] = float('nan')
    mask = mask.to(dev)
    graph = os.environ.get('GRAPH', '1') == '1'
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, capturable=graph, fused=True)
    bucket = GradBucket(m.parameters())
    rec = {'video': 1.0, 'audio': 1.0}
    eager = lambda: elbo_step(m, opt, bucket, x, mask, lengths, 1.0, rec, targets=tg)     # noqa: E731
    if not graph:
        return eager
    from mdmm.harness import GraphedElboStep
    step = GraphedElboStep(m, opt, bucket, x, mask, lengths, 1.0, rec, targets=tg)
    step.eager = eager
    return step


out = {}
if int(os.environ.get('B5', 0)) > 0:
    B = int(os.environ['B5'])
    try:
        dt, top = timed(vidtimit(B), n=2)
        res = {'batch': B, 's_per_step': round(dt, 4), 'seq_per_s': round(B / dt, 2), 'top_kernels_ms': top,
               'max_mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 2)}
    except Exception as e:   # noqa
        res = {'error': repr(e)[:300]}
    print('cfg5 vidTIMIT-shaped DMM z256 T128', json.dumps(res), flush=True)
    torch.cuda.empty_cache()
if os.environ.get('CFG1', '1') != '0':
    step = spirals_cfg1()
    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print('cfg1 Spirals z5 h20 T100 (graph replay)', json.dumps({'batch': 25, 's_per_step': round(dt, 5), 'seq_per_s': round(25 / dt, 1)}), flush=True)
    del step
for name, kind, B in (('cfg3 Weizmann DMM z256 T40', 'dmm', int(os.environ.get('B3', 32))),
                      ('cfg4 Weizmann DKS b-skip z256 T40', 'dks', int(os.environ.get('B4', 64)))):
    if B <= 0:
        continue
    try:
        dt, top = timed(weizmann(kind, B))
        out[name] = {'batch': B, 's_per_step': round(dt, 4), 'seq_per_s': round(B / dt, 2), 'top_kernels_ms': top,
                     'max_mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 2)}
    except Exception as e:   # noqa
        out[name] = {'error': repr(e)[:300]}
    print(name, json.dumps(out[name]), flush=True)
    torch.cuda.empty_cache()
