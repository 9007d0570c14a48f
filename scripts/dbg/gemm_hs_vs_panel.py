"""x[N,128] . H[128,128] on prebuilt operands: the H-stationary kernel against the LDS-panel kernel over N (run twice:
PHC_GEMM_NO_HS=1 for the second).  Prints microseconds per launch (HIP events around 20 launches, after 5 warm-up)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phc_gnn_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for K in ([int(a) for a in sys.argv[1:]] or (128, 256)):
    n, k = 4, K // 4
    A = torch.randn(n, n, n, device=dev, generator=g)
    W = torch.randn(n, k, k, device=dev, generator=g) * 0.1
    b = torch.randn(K, device=dev, generator=g)
    pre = ops.phm_build_operands([(A, W)])[0]
    for N in (16384, 32768, 65536, 262144, 1048576, 4194304, 10000000):
        if K == 256 and N > 4194304:
            continue
        x = torch.randn(N, K, device=dev, generator=g)
        for dx in (False, True):
            for _ in range(5):
                y = ops.phm_linear_pre(x, pre, b, dx=dx)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = ops.phm_linear_pre(x, pre, b, dx=dx)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            print(f"K={K} N={N:9d} {'dx ' if dx else 'fwd'} {us:9.1f} us  {2.0 * N * K * K / us / 1e6:7.1f} TFLOP/s  "
                  f"{'panel' if os.environ.get('PHC_GEMM_NO_HS') else 'hs'}", flush=True)
        del x, y
