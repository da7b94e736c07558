"""GPU box: bench.py with one product switch forced off by a monkeypatch (same-box A/B of a change that has no
environment switch).  usage: python tools/bench_patched.py <patch> [bench.py arguments]
  frames_f32   MultiDGTS._frames_store -> fp32 (cleaned frames as before mdmm_nan_to_zero_bf16)
  no_consume   ops.nll_bernoulli_logits ignores consume (the Bernoulli loss's separate backward kernel)
  no_colsum_a  the Linear heads' bias gradient by the column-sum kernel (mdmm_gemm_t.colsum_a not asked for)
  none         nothing patched"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
which = sys.argv[1]
sys.argv = ['bench.py'] + sys.argv[2:]
import bench          # noqa: E402  (spawns nothing at import; the package import order is bench.py's)
import torch          # noqa: E402
from mdmm import ops  # noqa: E402
from mdmm.models import dgts  # noqa: E402
for w in which.split('+'):
    if w == 'frames_f32':
        dgts.MultiDGTS._frames_store = lambda self, enc, x: torch.float32
    elif w == 'no_consume':
        ops.scaled_grad_ok = lambda logits: False
    elif w == 'no_colsum_a':
        _g = ops._gemm_bf16
        ops._gemm_bf16 = lambda *a, **k: ((_g(*a, **{kk: vv for kk, vv in k.items() if kk != 'colsum_a'}), None)
                                          if k.get('colsum_a') else _g(*a, **k))
    elif w != 'none':
        sys.exit('unknown patch ' + w)
bench.main()
