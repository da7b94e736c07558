"""GPU box: what a plain streaming pass reaches on this part (the ceiling the conv chain's TB/s figures are read against):
torch copy / read-only sum / fill of 2 GiB buffers, HIP events, best of 5."""
import torch
dev = torch.device('cuda:0')
n = 2 * 1024 ** 3 // 4
x = torch.randn(n, device=dev); y = torch.empty_like(x)
xb = x.to(torch.bfloat16); yb = torch.empty_like(xb)
def best(fn, nbytes, rep=5):
    ts = []
    for _ in range(rep):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return nbytes / min(ts) / 1e9
print('copy fp32   (read + write)  %.2f TB/s' % best(lambda: y.copy_(x), 2 * n * 4))
print('copy bf16   (read + write)  %.2f TB/s' % best(lambda: yb.copy_(xb), 2 * n * 2))
print('add  fp32   (2 reads + write) %.2f TB/s' % best(lambda: torch.add(x, y, out=y), 3 * n * 4))
print('sum  fp32   (read)          %.2f TB/s' % best(lambda: x.sum(), n * 4))
print('fill fp32   (write)         %.2f TB/s' % best(lambda: y.fill_(1.0), n * 4))
