"""GPU box: the Linear layers around the audio stacks alone at the cfg5 per-GPU size (HIP events, best of 5):
z_to_feat (256 -> 2576 + ReLU) on 131,072 rows and the two encoder heads (2576 -> 256) on 65,536 rows, forward,
input gradient, weight gradient -- bytes each has to move and the rate it reaches."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
import mdmm
from mdmm import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def report(name, ms, nbytes):
    print('%-44s %8.3f ms  %6.2f GB  %5.2f TB/s' % (name, ms, nbytes / 1e9, nbytes / ms / 1e9))


ops.TIMER = None
with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
    # decoder: z (M, 256) fp32 -> feat (M, 2576) bf16
    M = 131072
    lin = torch.nn.Linear(256, 2576).to(dev)
    z = torch.randn(M, 256, device=dev, requires_grad=True)
    y = ops.plug_linear(lin, z, act_out=True, relu=True)
    gy = torch.randn_like(y)
    report('z_to_feat forward (M=131072, 256 -> 2576)', timed(lambda: ops.plug_linear(lin, z, act_out=True, relu=True)), M * (256 * 4 + 2576 * 2))

    def fb():
        yy = ops.plug_linear(lin, z, act_out=True, relu=True)
        yy.backward(gy)
        z.grad = None; lin.weight.grad = None; lin.bias.grad = None
    t = timed(fb)
    report('z_to_feat forward + backward', t, M * (256 * 4 + 2576 * 2) * 2 + M * 2576 * 2 * 2 + M * 256 * 4)
    # encoder heads: feats (M2, 2576) bf16 -> (M2, 256) fp32, two of them
    M2 = 65536
    h1, h2 = torch.nn.Linear(2576, 256).to(dev), torch.nn.Linear(2576, 256).to(dev)
    f = torch.randn(M2, 2576, device=dev).to(torch.bfloat16).requires_grad_(True)
    g1, g2 = torch.randn(M2, 256, device=dev), torch.randn(M2, 256, device=dev)
    report('two heads forward (M=65536, 2576 -> 256)', timed(lambda: (ops.plug_linear(h1, f), ops.plug_linear(h2, f))), M2 * 2576 * 2 + 2 * M2 * 256 * 4)

    def fb2():
        a, b = ops.plug_linear(h1, f), ops.plug_linear(h2, f)
        torch.autograd.backward([a, b], [g1, g2])
        f.grad = None
        for p in list(h1.parameters()) + list(h2.parameters()):
            p.grad = None
    report('two heads forward + backward', timed(fb2), M2 * 2576 * 2 * 3 + 4 * M2 * 256 * 4)
ops.TIMER = timer = ops.KernelTimer()
with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
    fb(); fb2()
torch.cuda.synchronize()
for k, (n, ms) in sorted(timer.summary().items(), key=lambda kv: -kv[1][1]):
    print('%-36s %3d calls %8.3f ms' % (k, n, ms))

# ---- the same layers as they run since the feature rows are padded to audio.FEAT_PAD = 2816 (head GEMM kernels)
from mdmm import audio
print('padded rows (2816): the head kernels')
ops.TIMER = None
with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
    def fbp():
        w, b = audio.pad_linear_out(lin)
        yy = ops._LinearTilesFn.apply(z, w, b, torch.bfloat16, True)
        yy.backward(gyp)
        z.grad = None; lin.weight.grad = None; lin.bias.grad = None
    gyp = torch.randn(M, audio.FEAT_PAD, device=dev).to(torch.bfloat16)
    report('z_to_feat forward + backward (padded)', timed(fbp), M * (256 * 4 + 2816 * 2) * 2 + M * 2816 * 2 * 2 + M * 256 * 4)
    fpad = torch.randn(M2, audio.FEAT_PAD, device=dev).to(torch.bfloat16).requires_grad_(True)

    def fb2p():
        a = ops.linear_tiles(fpad, audio.pad_linear_in(h1), h1.bias)
        b = ops.linear_tiles(fpad, audio.pad_linear_in(h2), h2.bias)
        torch.autograd.backward([a, b], [g1, g2])
        fpad.grad = None
        for p in list(h1.parameters()) + list(h2.parameters()):
            p.grad = None
    report('two heads forward + backward (padded)', timed(fb2p), M2 * 2816 * 2 * 3 + 4 * M2 * 256 * 4)
ops.TIMER = timer = ops.KernelTimer()
with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
    fbp(); fb2p()
torch.cuda.synchronize()
for k, (n, ms) in sorted(timer.summary().items(), key=lambda kv: -kv[1][1]):
    print('%-36s %3d calls %8.3f ms' % (k, n, ms))
