import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phc_gnn_amd, torch
from phc_gnn_amd import ops, lib as L
from tests.test_gpu_downstream import _make
dn = _make(4, 128, (128, 32), 1, "relu", seed=7).to("cuda:0")
x = torch.randn(544, 128, device="cuda:0")
lib = L.load()
names = {0: "start", 1: "staged", 2: "gemm1", 3: "published1", 4: "rendezvous1", 5: "norm1 done", 6: "applied1", 7: "gemm2", 8: "published2", 9: "rendezvous2", 10: "norm2 done", 11: "end"}
with ops.prebuilt_H(ops.phm_pairs(dn), ops.dense_weights(dn)):
    for it in range(5):
        y = dn(x)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 64)()
        rc = C.CDLL(L.LIB_PATH).phc_debug_dn_timestamps(buf)  # same handle: dlopen of an already loaded path
        for base, who in ((0, "wg0"), (32, "wgLast")):
            t0 = buf[base]
            print(it, who, " ".join(f"{names[i]}={(buf[base + i] - t0) / 100.0:.2f}us" for i in range(1, 12)), flush=True)
