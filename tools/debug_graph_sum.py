import torch
dev = torch.device('cuda:0')
for shape in ((234, 1792), (240, 768), (234, 1024), (10240, 1792)):
    G = torch.randn(*shape, device=dev)
    s = torch.ones(1, device=dev)
    junk = [torch.randn(1 << 22, device=dev) * 1e3 for _ in range(4)]; del junk
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            y = (G * s).sum(0)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        t = torch.empty(1 << 20, device=dev).fill_(float('nan'))      # another intermediate in the pool
        del t
        G2 = G * s
        y = G2.sum(0)
        z = G2.sum(0, dtype=torch.float32)[5:100]
    for r in range(3):
        s.fill_(r + 1.0)
        g.replay(); torch.cuda.synchronize()
        ref = (G * s).sum(0)
        print(shape, 'replay', r, 'rel err', float((y - ref).norm() / ref.norm()), 'finite', bool(torch.isfinite(y).all()))
