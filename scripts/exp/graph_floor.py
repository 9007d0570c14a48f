"""Per-node floor of a hipGraph replay: N trivial dependent kernels in one graph."""
import time
import torch

dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
big = torch.zeros(12874, 512, device=dev)
for n_nodes, t in ((200, x), (400, x), (200, big)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            t.add_(1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n_nodes):
                t.add_(1.0)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"nodes={n_nodes} numel={t.numel()}: {dt*1e6:.1f} us/replay, {dt*1e6/n_nodes:.2f} us/node")
    # eager back-to-back
    t0 = time.perf_counter()
    for _ in range(reps):
        for _ in range(n_nodes):
            t.add_(1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"   eager: {dt*1e6:.1f} us, {dt*1e6/n_nodes:.2f} us/launch")
