"""Build container only (needs /root/reference): step time of the CPU oracle (oracle/mdmm_oracle.py) against
the unmodified reference on the same shapes, weights, inputs and host cores -- the fidelity figure
BASELINE.md section 3 asks for next to bench.py's `cpu_baseline` (kind "port").
usage: python tools/oracle_vs_reference_time.py [cfg3|cfg2] [B] [steps]"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
_orig = torch.Tensor.__rsub__
torch.Tensor.__rsub__ = lambda self, other: (~self if self.dtype is torch.bool and not torch.is_tensor(other) and other == 1
                                             else _orig(self, other))
sys.path.insert(0, '/root/reference')
import models as ref_models          # noqa: E402  (the reference package)
import bench                         # noqa: E402
from oracle import mdmm_oracle as orc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
cfg = bench.CONFIGS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
x, tg, mask, lengths = cfg.batch(cfg.T, B, 1234, 'cpu')
torch.manual_seed(0)
o = cfg.oracle(orc)
if name == 'cfg2':
    ref = ref_models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32, device=torch.device('cpu'))
else:
    C = ref_models.common
    ref = ref_models.MultiDMM(cfg.mods, cfg.dims, cfg.dists,
                              encoders={'video': C.ImageEncoder(256, n_channels=3), 'mask': C.ImageEncoder(256, n_channels=1)},
                              decoders={'video': C.ImageDecoder(256, n_channels=3), 'mask': C.ImageDecoder(256, n_channels=1)},
                              h_dim=256, z_dim=256, device=torch.device('cpu'))
ref.load_state_dict(o.state_dict())


def timed(model):
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    def one():
        loss = model.step(x, mask, 1.0, cfg.rec, targets=tg, lengths=lengths)
        (loss / sum(lengths)).backward(); opt.step(); opt.zero_grad()
        return float(loss)
    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    return (time.perf_counter() - t0) / steps


t_ref, t_orc = timed(ref), timed(o)
print('%s B=%d threads=%d: reference %.2f s/step (%.2f seq/s), oracle %.2f s/step (%.2f seq/s), oracle/reference time ratio %.2f'
      % (name, B, torch.get_num_threads(), t_ref, B / t_ref, t_orc, B / t_orc, t_orc / t_ref))
