"""GPU box: the audio plug-ins' nodes alone at the cfg5 per-GPU size (mdmm.audio on csrc/audio_chain.hip): HIP events per
node and direction, bytes each launch must move, achieved rate.  usage: python tools/time_audio.py [B=512] [T=128] [reps=5]
Under `rocprofv3 --kernel-trace --stats` the per-kernel durations are those of kernels that run ALONE on the chip."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
import mdmm
from mdmm import ops, audio
from mdmm.models import common as C

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device('cuda:0')
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
lengths = sorted(torch.randint(T // 2, T + 1, (B,), generator=g).tolist(), reverse=True)
mask = torch.zeros(T, B)
for b, n in enumerate(lengths):
    mask[:n, b] = 1
mask = mask.to(dev)
target = torch.rand(T, B, 10, 1281, device=dev)
target[mask == 0] = float('nan')
frames = target.clone()
frames[torch.rand(T, B, device=dev) < 0.5] = float('nan')
rows, passes = T * B, 2
dec = C.AudioDecoder(256).to(dev).train()
enc = C.AudioEncoder(256).to(dev).train()
z = torch.randn(passes * rows, 256, device=dev, requires_grad=True)


def timed(fn, n=reps):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def dec_fwd(bwd):
    total = ops.LossSum(dev)
    with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
        dec.nll(z, audio.decoder_plan(dec), target, mask, 1.0, total, passes, [0.5, 0.5], True)
    loss = total.total()
    if bwd:
        loss.backward()
        z.grad = None
        for p in dec.parameters():
            p.grad = None


def enc_fwd(bwd):
    with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
        mean, std, seen = enc.encode_frames(frames.flatten(0, 1), audio.encoder_plan(enc))
    if bwd:
        (mean.sum() + std.sum()).backward()
        for p in enc.parameters():
            p.grad = None


seen_rows = float(mask.sum())
print('B = %d, T = %d: %d rows (%.0f unmasked), decoder batch %d frames' % (B, T, rows, seen_rows, passes * rows))
t_f, t_fb = timed(lambda: dec_fwd(False)), timed(lambda: dec_fwd(True))
# bytes the decoder node must move: z_to_feat output 2576 bf16 written + read, two 5 KB layers written + read, target once per row
dec_bytes_f = passes * rows * (2576 * 2 * 2 + 2568 * 2 * 2 + 2564 * 2 * 2) + seen_rows * 12810 * 4
print('decoder nll forward %.3f ms (%.2f TB/s of %.2f GB), forward + backward %.3f ms' % (t_f, dec_bytes_f / t_f / 1e9, dec_bytes_f / 1e9, t_fb))
t_f, t_fb = timed(lambda: enc_fwd(False)), timed(lambda: enc_fwd(True))
enc_bytes_f = rows * (12810 * 4 + (2564 + 2568 + 2576) * 2 * 2)
print('encoder forward %.3f ms (%.2f TB/s of %.2f GB), forward + backward %.3f ms' % (t_f, enc_bytes_f / t_f / 1e9, enc_bytes_f / 1e9, t_fb))
if os.environ.get('SPANS') == '1':       # per library call, HIP events (kernels alone on the chip)
    ops.TIMER = timer = ops.KernelTimer()
    for _ in range(3):
        dec_fwd(True); enc_fwd(True)
    torch.cuda.synchronize()
    ops.TIMER = None
    for k, (n, ms) in sorted(timer.summary().items(), key=lambda kv: -kv[1][1]):
        if k.startswith('audio'):
            print('%-28s %3d calls %8.3f ms each' % (k, n, ms / n))
