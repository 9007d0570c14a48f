import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phc_gnn_amd, torch
from phc_gnn_amd import ops
from oracle import phc_oracle as O
from tests.test_gpu_downstream import _make
import torch.nn.functional as F
for (n, d_in, hidden, target, act, B) in [(4, 64, (64, 32), 2, "swish", 100), (4, 64, (64, 32), 2, "relu", 100), (4, 128, (128, 32), 2, "relu", 100), (4, 64, (64, 32), 1, "relu", 100), (4,128,(64,32),1,"relu",100), (4,64,(128,32),1,"relu",100)]:
    dn = _make(n, d_in, hidden, target, act, seed=7)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, d_in, generator=g)
    p = {k: v.detach().clone() for k, v in dn.state_dict().items()}
    z1r = O.phm_linear(x, p["affine.0.phm_rule"], p["affine.0.W"], p["affine.0.b"])
    a1 = O.activation(act, O._norm_from_state(z1r, p, "norm.0.", n, True))
    z2r = O.phm_linear(a1, p["affine.1.phm_rule"], p["affine.1.W"], p["affine.1.b"])
    a2 = O.activation(act, O._norm_from_state(z2r, p, "norm.1.", n, True))
    z3r = O.phm_linear(a2, p["affine.2.phm_rule"], p["affine.2.W"], p["affine.2.b"])
    yr = F.linear(z3r, p["real_trafo.affine.weight"], p["real_trafo.affine.bias"])
    dn.to("cuda:0")
    xd = x.to("cuda:0").requires_grad_()
    with ops.prebuilt_H(ops.phm_pairs(dn), ops.dense_weights(dn)):
        y = dn(xd)
        _, z1, z2, z3, s1, s2, _ = y.grad_fn.saved_tensors
    def e(a, b): return float((a.cpu() - b).abs().max() / b.abs().max())
    print((n, d_in, hidden, target, act, B), "z1", e(z1, z1r), "z2", e(z2, z2r), "z3", e(z3, z3r), "y", e(y.detach(), yr), flush=True)
