import os, sys, torch
sys.path.insert(0, "/root/repo")
from phc_gnn_amd import lib
l = lib.load(); dev = torch.device("cuda:0")
N, n, k, h = 12874, 4, 32, 128
x = torch.randn(N, h, device=dev); dy = torch.randn(N, h, device=dev)
A = torch.randn(n, n, n, device=dev); W = torch.randn(n, k, k, device=dev)
dW = torch.empty_like(W); dA = torch.empty_like(A); db = torch.empty(h, device=dev)
wsb = l.phc_phm_linear_bwd_ws_bytes(N, n, k, k); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
sp = lib.stream_ptr(dev)
f = lambda: l.phc_phm_linear_bwd(x.data_ptr(), A.data_ptr(), W.data_ptr(), dy.data_ptr(), N, n, k, k, None, dW.data_ptr(), dA.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, sp)
for _ in range(5): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): f()
b.record(); torch.cuda.synchronize()
print(os.environ.get("PHC_DH_BLOCKS", "512"), "dW/dA/db path: %.1f us" % (a.elapsed_time(b) * 10), "ws MB %.1f" % (wsb / 1e6))
