import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phc_gnn_amd import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for N in (30001, 70001):
    K = 128; n = 4; k = K // n
    A = torch.randn(n, n, n, device=dev, generator=g); W = torch.randn(n, k, k, device=dev, generator=g) * 0.1
    b = torch.randn(K, device=dev, generator=g)
    pre = ops.phm_build_operands([(A, W)])[0]
    x = torch.randn(N, K, device=dev, generator=g)
    z0 = ops.phm_linear_pre(x, pre, b)
    z, acc = ops.gemm_pre_stats(x, pre, b)
    print(N, "z", (z - z0).abs().max().item())
    m0 = ops.bn_moments(z0)
    m1 = ops.stat_acc_read(acc, K, N)
    print(" sum", (m0[:K] - m1[:K]).abs().max().item(), m0[:K].abs().max().item(), " M2", (m0[K:] - m1[K:]).abs().max().item(), m0[K:].abs().max().item())
    ref_sum = z0.double().sum(0); ref_m2 = ((z0.double() - z0.double().mean(0)) ** 2).sum(0)
    print(" vs fp64: sum", (m1[:K].double() - ref_sum).abs().max().item(), "M2", (m1[K:].double() - ref_m2).abs().max().item())
    counts = torch.tensor([float(N)], device=dev)
    rm = [torch.zeros(k, device=dev) for _ in range(n)]; rv = [torch.ones(k, device=dev) for _ in range(n)]
    nbt = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(n)]
    w = [torch.rand(k, device=dev, generator=g) + 0.5 for _ in range(n)]; bb = [torch.randn(k, device=dev, generator=g) for _ in range(n)]
    a0, st0 = ops.bn_fwd_from_moments(z0, m0.view(1, -1), counts, N, w, bb, rm, rv, nbt, 0.1, 1e-5, 1)
    st1 = ops.bn_finalize_from_moments(m1.view(1, -1), counts, N, w, bb, rm, rv, nbt, 0.1, 1e-5, K)
    print(" save_stat", [(st0[i * K:(i + 1) * K] - st1[i * K:(i + 1) * K]).abs().max().item() for i in range(6)])
    y0 = ops.phm_linear_pre(a0, pre, b)
    y1 = ops.gemm_pre_xf(z, st1, 1, pre, b)
    print(" y", (y0 - y1).abs().max().item(), y0.abs().max().item())
