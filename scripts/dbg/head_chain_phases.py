"""Phase clocks of the head-chain launches (csrc/head_chain.hip built with -DPHC_HC_TIMING):

    python -m phc_gnn_amd.build --variant hct -DPHC_HC_TIMING
    PHC_LIB_PATH=scripts/exp/variants/libphc_hip_hct.so python scripts/dbg/head_chain_phases.py [B]

Prints, for the first and the last workgroup of the grid, the time of every stamp relative to the launch's first stamp (100 MHz wall
clock -> us): stamp 8 s + 1 = stage s begins, + 2 = slab in LDS, + 3 = GEMM done, + 4 = partial statistics published, + 5 = collected,
+ 6 = post-op done; 40 = all stages done, 41 = end."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from phc_gnn_amd import lib, ops  # noqa: E402
from tests.test_gpu_downstream import _make  # noqa: E402

DEV = "cuda:0"


def stamps():
    L = lib.load()
    L.phc_debug_hc_timestamps.restype = C.c_int
    L.phc_debug_hc_timestamps.argtypes = [C.c_void_p]
    buf = (C.c_uint64 * 128)()
    torch.cuda.synchronize()
    assert L.phc_debug_hc_timestamps(buf) == 0
    return list(buf)


def show(tag, ts):
    for base, who in ((0, "first wg"), (64, "last wg ")):
        t0 = ts[base]
        pts = [(i, (ts[base + i] - t0) / 100.0) for i in range(64) if ts[base + i] >= t0 and ts[base + i] != 0]
        print(f"{tag} {who}: " + "  ".join(f"{i}:{t:.1f}" for i, t in pts if t < 1e4))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    d_in = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dn = _make(4, d_in, (768, 256), 128, "relu", seed=3).to(DEV)
    g = torch.Generator().manual_seed(1)
    x, dy = torch.randn(B, d_in, generator=g).to(DEV), torch.randn(B, 128, generator=g).to(DEV)
    for it in range(4):
        xd = x.clone().requires_grad_()
        with ops.prebuilt_H(ops.phm_pairs(dn), ops.dense_weights(dn)):
            y = dn(xd)
            assert type(y.grad_fn).__name__ == "_HeadChainFnBackward"
            f = stamps()
            y.backward(dy)
            b = stamps()
    show("fwd", f)
    show("bwd", b)


if __name__ == "__main__":
    main()
