"""In-kernel phase times of the register-resident dH kernel (-DPHC_KTIMING variant build):
    python -m phc_gnn_amd.build --variant ktiming -DPHC_KTIMING
    PHC_LIB_PATH=scripts/exp/variants/libphc_hip_ktiming.so python scripts/dbg/dh_time.py [N] [layers]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phc_gnn_amd, torch
from phc_gnn_amd import ops, lib as L
from phc.hypercomplex.layers import PHMLinear
N = int(sys.argv[1]) if len(sys.argv) > 1 else 13824
J = int(sys.argv[2]) if len(sys.argv) > 2 else 9
dev = torch.device("cuda:0")
lins = [PHMLinear(128, 128, phm_dim=4).to(dev) for _ in range(J)]
xs = [torch.randn(N, 128, device=dev) for _ in range(J)]
L.load()
dbg = C.CDLL(L.LIB_PATH)
names = {1: "accumulators zeroed", 2: "row loops done", 3: "exchange + slab stores issued", 4: "end"}
for it in range(5):
    with ops.deferred_param_grads():
        tot = 0.0
        for lin, x in zip(lins, xs):
            tot = tot + ops.phm_linear(x, lin.phm_rule, lin.W, lin.b).square().sum()
        tot.backward()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    dbg.phc_debug_dh_timestamps(buf)
    if it >= 2:
        for base, who in ((0, "first wg"), (32, "last wg")):
            t0 = buf[base]
            print(it, who, " ".join(f"{names[i]}={(buf[base + i] - t0) / 100.0:.2f}" for i in sorted(names)), flush=True)
