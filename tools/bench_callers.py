"""GPU box: the callers either side of the ELBO step, timed (round-4 review, missing #5):

  batch_prep  seq_collate_dict + burst_delete (datasets/multiseq.py:341-434, trainer.py:231-235) on the device, at the
              vidTIMIT-shaped per-GPU batch (512 items, video (len,3,64,64) + audio (len,10,1281), len 64..128) and at
              4,096 Weizmann-shaped items (mask (len,1,64,64) + action (len,1), len 20..40): wall time of the call
              (host packing + one host-to-device copy per modality + the kernels) and HIP-event time of the device
              passes with their bytes against the 8 TB/s HBM peak; the deletion with the device generator and with the
              reference's numpy draw order (rng='numpy': B x M host draws) side by side.
  eval        Trainer.evaluate's body (trainer.py:278-312) at cfg3 size: evaluation forward with flt_particles = 200
              (trainer.py:358-361) + compute_weizmann_metrics (weizmann.py:116-166) + seq_decoll_dict of the
              reconstructions (multiseq.py:388-403), sequences/s.

usage: python tools/bench_callers.py [batch_prep] [eval]      (bench.py imports measure_* for its `extra` entries)"""
import os
import sys
import time

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))

HBM_PEAK_GBS = 8000.0


def _items(n, lo, hi, shapes, seed):
    import numpy as np
    rs = np.random.RandomState(seed)
    lens = rs.randint(lo, hi + 1, n)
    data = []
    for i, ln in enumerate(lens):
        d = {m: rs.rand(int(ln), *shape).astype(np.float32) for m, shape in shapes.items()}
        d['length'], d['id'] = int(ln), i
        data.append(d)
    return data


def _timed(fn, reps=3):
    """(best wall ms, best HIP-event ms on the current stream) of fn()"""
    import torch
    wall, devt = [], []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        wall.append(1e3 * (time.perf_counter() - t0))
        devt.append(e0.elapsed_time(e1))
        del out
    return min(wall), min(devt)


def measure_batch_prep(device):
    import numpy as np
    import torch
    from mdmm import batch
    out = {}
    cases = [('vidTIMIT-shaped, 512 items (cfg5 per GPU)', 512, 64, 128, {'video': (3, 64, 64), 'audio': (10, 1281)}),
             ('Weizmann-shaped, 4096 items', 4096, 20, 40, {'mask': (1, 64, 64), 'action': (1,)})]
    for name, n, lo, hi, shapes in cases:
        data = _items(n, lo, hi, shapes, 7)
        in_bytes = sum(d[m].nbytes for d in data for m in shapes)
        w_col, d_col = _timed(lambda: batch.seq_collate_dict(data, device=device), reps=2)
        b, mask, lengths, order, _ = batch.seq_collate_dict(data, device=device)
        out_bytes = sum(v.numel() * 4 for v in b.values())
        gen = torch.Generator(device=device).manual_seed(1)
        w_del, d_del = _timed(lambda: batch.burst_delete(b, 0.2, lengths, generator=gen))
        np.random.seed(3)
        w_np, d_np = _timed(lambda: batch.burst_delete(b, 0.2, lengths, rng='numpy'))
        # the padded merge reads the packed items once and writes the batch once; the deletion reads and writes the batch
        out[name] = {
            'items': n, 'T': int(max(lengths)), 'item_bytes': in_bytes, 'batch_bytes': out_bytes,
            'seq_collate_dict': {'wall_ms': round(w_col, 3), 'stream_ms': round(d_col, 3),
                                 'host_gbs': round(in_bytes / w_col / 1e6, 1),
                                 'note': 'host threads pack the items into a pinned staging buffer piece by piece, every piece '
                                         'copied to the device as it is packed (batch._to_device_packed), then mdmm_collate_pad; '
                                         'host_gbs = item bytes over the wall time'},
            'burst_delete[device generator]': {'wall_ms': round(w_del, 3), 'stream_ms': round(d_del, 3),
                                               'hbm_gbs': round(2 * out_bytes / d_del / 1e6, 1),
                                               'hbm_frac': round(2 * out_bytes / d_del / 1e6 / HBM_PEAK_GBS, 4)},
            'burst_delete[numpy order]': {'wall_ms': round(w_np, 3), 'stream_ms': round(d_np, 3),
                                          'note': 'B x M host draws in the reference\'s call order (bit-exact against golden G10)'},
        }
        del b, mask, data
        torch.cuda.empty_cache()
    return out


def measure_eval(device, reps=2):
    import torch
    import bench
    from mdmm import batch, metrics, models
    from mdmm.noise import PhiloxNoise
    cfg = bench.Cfg3
    torch.manual_seed(0)
    m = cfg.model(models, device).eval()
    m.noise = PhiloxNoise(seed=5)
    x, tg, mask, lengths = cfg.batch(cfg.T, cfg.B, 99, device)
    order = list(range(cfg.B))
    parts = {}

    def body():
        with torch.no_grad():
            t0 = time.perf_counter()
            infer, prior, recon = m(x, lengths=lengths, sample=False, flt_particles=200)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            met = metrics.compute_weizmann_metrics(m, infer, prior, recon, tg, mask, lengths, order, cfg.rec)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            dec = batch.seq_decoll_dict(recon, lengths, order)
            t3 = time.perf_counter()
        parts.update(forward_ms=1e3 * (t1 - t0), metrics_ms=1e3 * (t2 - t1), decollate_ms=1e3 * (t3 - t2))
        return met, len(dec)

    body()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        body()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, dict(parts))
    dt, p = best
    return {'workload': 'cfg3 evaluation batch (B = %d, T = %d): forward(sample=False, flt_particles=200) + '
                        'compute_weizmann_metrics + seq_decoll_dict of the reconstructions' % (cfg.B, cfg.T),
            'value': round(cfg.B / dt, 1), 'unit': 'sequences/s', 'ms_per_batch': round(1e3 * dt, 2),
            'forward_ms': round(p['forward_ms'], 2), 'metrics_ms': round(p['metrics_ms'], 2),
            'decollate_ms': round(p['decollate_ms'], 2),
            'note': 'decollate = one device-to-host copy of the de-padded reconstructions (1.3 GB fp32) + numpy views'}


if __name__ == '__main__':
    import json
    import torch
    dev = torch.device('cuda:0')
    which = sys.argv[1:] or ['batch_prep', 'eval']
    res = {}
    if 'batch_prep' in which:
        res['batch_prep'] = measure_batch_prep(dev)
    if 'eval' in which:
        res['eval'] = measure_eval(dev)
    print(json.dumps(res, indent=1))
