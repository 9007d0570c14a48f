"""In-kernel phase clocks of the head's one-launch forward + backward (downstream2_fb_kernel; variant build -DPHC_DN_TIMING:
python -m phc_gnn_amd.build --variant dntime -DPHC_DN_TIMING, then PHC_LIB_PATH=scripts/exp/variants/libphc_hip_dntime.so
python scripts/dbg/dn_fb_time.py [B d0 d1 d2 n]).  Thread 0 of the first and last workgroup stamp wall_clock64() (100 MHz); printed:
microseconds since kernel entry, the forward part's phases, then the backward part's."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phc_gnn_amd, torch
from phc_gnn_amd import ops, lib as L
from tests.test_gpu_downstream import _make
B, d0, d1, d2, n = (int(v) for v in (sys.argv[1:6] + ["544", "128", "128", "32", "4"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
dn = _make(n, d0, (d1, d2), 1, "relu", seed=7).to(dev)
x = torch.randn(B, d0, device=dev).requires_grad_()
y = (torch.rand(B, 1, device=dev) < 0.1).float()
y[B - 32:] = float("nan")
seed = ops.register_unit_seed(torch.ones((), device=dev))
lib = L.load()
fwd = {12: "H1 issued", 13: "norm cols", 14: "x in LDS", 1: "staged", 2: "gemm1", 3: "published1", 4: "collected1", 5: "norm1 done", 6: "applied1", 7: "gemm2", 8: "published2", 9: "collected2",
       10: "norm2 done", 11: "logits"}
bwd = {16: "bwd part", 17: "staged", 18: "small", 19: "gated2", 20: "published2", 21: "collected2", 22: "dz2", 23: "gemm(dz2.H2T)", 24: "gated1", 25: "published1",
       26: "collected1", 27: "dz1", 28: "gemm(dz1.H1T)", 29: "end"}
with ops.prebuilt_H(ops.phm_pairs(dn), ops.dense_weights(dn)):
    for it in range(6):
        with ops.expect_loss(y, "bce"):
            logits = dn(x)
        loss = ops.masked_bce_with_logits(logits, y)
        loss.backward(seed)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 64)()
        rc = C.CDLL(L.LIB_PATH).phc_debug_dn_timestamps(buf)
        if it < 3:
            continue
        for base, who in ((0, "wg0"), (32, "wgLast")):
            print(it, who, "|", " ".join(f"{fwd[i]}={(buf[base + i] - buf[base]) / 100.0:.2f}" for i in fwd if buf[base + i]), "||",
                  " ".join(f"{bwd[i]}={(buf[base + i] - buf[base]) / 100.0:.2f}" for i in sorted(bwd) if buf[base + i]), flush=True)
