"""GPU box: who keeps the parameters' AccumulateGrad nodes alive between steps?  (PyTorch warns "The AccumulateGrad node's
stream does not match ..." when a node made while one stream was current receives a gradient produced on another.)
Control: a stock nn.Linear through the same sequence (side-stream step, capture, default-stream step).
usage: python tools/accgrad_probe.py"""
import gc
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multimodal-dmm_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402


def acc_node(p):
    return p.expand_as(p).grad_fn.next_functions[0][0]


def tag(params, name):
    for p in params:
        acc_node(p).metadata['tag'] = name


def alive(params):
    return sum(1 for p in params if 'tag' in acc_node(p).metadata), len(params)


def control():
    dev = torch.device('cuda:0')
    lin = nn.Linear(64, 64).to(dev)
    x = torch.randn(8, 64, device=dev)
    ps = list(lin.parameters())
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        lin(x).sum().backward()
        tag(ps, 'side')
    torch.cuda.current_stream().wait_stream(s)
    gc.collect()
    print('control: nodes made on the side stream still alive after the step:', alive(ps))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        lin(x).sum().backward()
        torch.cuda.synchronize()
    print('control: default-stream step warns:', any('AccumulateGrad' in str(i.message) for i in w))


def ours():
    import bench
    from mdmm import harness, ops  # noqa: F401
    dev = torch.device('cuda:0')
    from mdmm import models
    cfg = bench.Cfg2
    torch.manual_seed(0)
    model = cfg.model(models, dev)
    inputs, targets, mask, lengths = cfg.batch(cfg.T, 64, 1, dev)
    ps = [p for p in model.parameters() if p.requires_grad]
    loss = model.step(inputs, mask, 1.0, cfg.rec, targets=targets, lengths=lengths)
    loss.backward()
    tag(ps, 'first')
    del loss
    gc.collect()
    torch.cuda.synchronize()
    print('ours: nodes of an eager default-stream step alive after the step (loss deleted):', alive(ps))
    ops.clear_caches()
    gc.collect()
    print('ours: ... after ops.clear_caches():', alive(ps))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        with torch.cuda.stream(s):
            loss = model.step(inputs, mask, 1.0, cfg.rec, targets=targets, lengths=lengths)
            loss.backward()
        torch.cuda.synchronize()
    print('ours: a side-stream step after it warns:', any('AccumulateGrad' in str(i.message) for i in w))


if __name__ == '__main__':
    control()
    ours()
