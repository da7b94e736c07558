"""GPU box: dump the captured cfg2 step graph (hipGraphDebugDotPrint) and report, for the long
kernels, which other long kernels they depend on (are the side streams really independent?).
usage: python tools/graph_deps.py  -> gpurun_out/step_graph.dot + a summary"""
import os, re, sys, collections
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from bench import synth_batch
from mdmm import models, ops
from mdmm.harness import GradBucket
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
inputs, targets, mask, lengths = synth_batch(100, 1024, 1234, dev)
rec = {'spiral-x': .5, 'spiral-y': .5}
torch.manual_seed(0)
m = models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32, device=dev)
m.noise = PhiloxNoise(seed=1)
bucket = GradBucket(m.parameters())
def fwd_bwd():
    loss = m.step(inputs, mask, 1.0, rec, targets=targets, lengths=lengths, train_particles=25)
    (loss / 102400).backward()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        bucket.release(); fwd_bwd(); bucket.check_views()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
ops.clear_caches(m.parameters())
g = torch.cuda.CUDAGraph(); g.enable_debug_mode()
bucket.release()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    fwd_bwd(); bucket.check_views()
out = os.path.join(R, 'gpurun_out'); os.makedirs(out, exist_ok=True)
dot = os.path.join(out, 'step_graph.dot')
g.debug_dump(dot)
txt = open(dot).read()
label = {}
for mm in re.finditer(r'"?(\w+)"?\s*\[([^\]]*)\]', txt):
    lab = re.search(r'label="([^"]*)"', mm.group(2))
    if lab: label[mm.group(1)] = lab.group(1)
edges = collections.defaultdict(list)
for mm in re.finditer(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt):
    edges[mm.group(2)].append(mm.group(1))
print('%d nodes, %d edges' % (len(label), sum(len(v) for v in edges.values())))
big = {n: l for n, l in label.items() if re.search(r'sweep_mfma|sweep_bwd_kernel|sweep_fwd_kernel', l)}
def ancestors(n):
    seen, todo = set(), [n]
    while todo:
        for p in edges.get(todo.pop(), []):
            if p not in seen: seen.add(p); todo.append(p)
    return seen
def short(l): return re.sub(r'\\n.*', '', l)[:70]
for n, l in big.items():
    anc = ancestors(n)
    print('%s  %s\n    after: %s' % (n, short(l), '; '.join(sorted(short(big[a])[10:60] + '#' + a[-4:] for a in anc if a in big))))
