"""Every workgroup's phase clocks of the aggregating fused forward / the fused backward inside a molhiv_b512 step (eager):
PYTHONPATH=. python scripts/dbg/mlp_all_wg_phases.py [bwd]   (phc_debug_mlp_all_stamps; the last launch of the pass is shown)"""
import sys
import numpy as np
import torch
from phc_gnn_amd import harness, ops, lib as L
dev = torch.device("cuda:0")
lib = L.load()
bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
model = harness.build_model("molhiv_b512", dev, seed=0).train()
data = harness.to_device(harness.make_batch("molhiv_b512", seed=1), dev)
G = 448
st = torch.zeros(8 * G, dtype=torch.int64, device=dev)
res = []
for it in range(8):
    with ops.stat_accumulators(dev), ops.prebuilt_H(ops.phm_pairs(model), ops.dense_weights(model), ops.gate_specs(model),
                                                   encode=ops.atom_encode_spec(model, data)), ops.deferred_param_grads():
        if not bwd:
            st.zero_()
            lib.phc_debug_mlp_all_stamps(L.ptr(st))
        logits = model(data)
        torch.cuda.synchronize()
        if not bwd:
            lib.phc_debug_mlp_all_stamps(None)
        loss = logits.float().mean()
        if bwd:
            st.zero_()
            lib.phc_debug_mlp_all_stamps(L.ptr(st))
        loss.backward()
        torch.cuda.synchronize()
        lib.phc_debug_mlp_all_stamps(None)
    res.append(st.clone().cpu().numpy().reshape(G, 8))
r = res[-1].astype(np.float64)
live = r[:, 0] > 0
r = r[live]
t0 = r[:, 0].min()
r = (r - t0) / 100.0
ids = np.nonzero(live)[0]
names = ["entry", "P0 done", "GEMM1", "arrive", "release", "XF", "GEMM2", "end"] if not bwd else ["entry", "gate0", "arrive1", "release1", "P1 done", "arrive2", "release2", "end"]
print("workgroups:", len(r))
for k, nm in enumerate(names):
    c = r[:, k]
    print(f"{nm:9s} min {c.min():6.2f}  p50 {np.median(c):6.2f}  p90 {np.percentile(c, 90):6.2f}  max {c.max():6.2f}   (us since the first entry)")
ak = 3 if not bwd else 2
order = np.argsort(-r[:, ak])[:10]
print("latest arrivals (block, xcd, entry, first phase dur, arrive):")
for o in order:
    print(f"  block {ids[o]:4d} xcd {ids[o] % 8}  entry {r[o,0]:5.2f}  phase1 {r[o,1]-r[o,0]:5.2f}  arrive {r[o,ak]:6.2f}")
d = r[:, 1] - r[:, 0]
print(f"first phase duration: min {d.min():.2f} p50 {np.median(d):.2f} p90 {np.percentile(d,90):.2f} max {d.max():.2f}")
first = ids < 256
for nm, m in (("blocks < 256", first), ("blocks >= 256", ~first)):
    if m.any():
        print(f"{nm}: n {m.sum()}  entry p50 {np.median(r[m,0]):.2f}  phase1 p50 {np.median(d[m]):.2f} max {d[m].max():.2f}  arrive p50 {np.median(r[m,ak]):.2f} p90 {np.percentile(r[m,ak],90):.2f} max {r[m,ak].max():.2f}  end p50 {np.median(r[m,7]):.2f} max {r[m,7].max():.2f}")
