import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phc_gnn_amd, torch
from phc_gnn_amd import ops, lib as L
from phc.hypercomplex.layers import PHMLinear
N = int(sys.argv[1]) if len(sys.argv) > 1 else 13824
lin = PHMLinear(128, 128, phm_dim=4).to("cuda:0")
x = torch.randn(N, 128, device="cuda:0")
lib = L.load()
dbg = C.CDLL(L.LIB_PATH)
names = {1: "panel+sync", 2: "x half0 staged", 4: "x half1 staged (half0 MFMAs done)", 6: "MFMAs done", 7: "epilogue stores issued", 8: "end"}
for mode in ("plain", "stats"):
    with ops.prebuilt_H([(lin.phm_rule, lin.W)], []):
        for it in range(4):
            req = ops.BNStatsRequest(None) if mode == "stats" else None
            with torch.no_grad():
                y = ops.phm_linear(x, lin.phm_rule, lin.W, lin.b, bn_stats=req)
            torch.cuda.synchronize()
            buf = (C.c_ulonglong * 64)()
            dbg.phc_debug_stream_timestamps(buf)
            if it >= 2:
                for base, who in ((0, "first wg"), (32, "last wg")):
                    t0 = buf[base]
                    print(mode, it, who, " ".join(f"{names[i]}={(buf[base + i] - t0) / 100.0:.2f}" for i in sorted(names)), flush=True)
