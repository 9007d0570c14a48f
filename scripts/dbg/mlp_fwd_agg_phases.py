"""Phase clocks of the aggregating fused forward (last layer of a molhiv_b512 forward pass): PYTHONPATH=. python scripts/dbg/mlp_fwd_agg_phases.py"""
import torch
from phc_gnn_amd import harness, ops, lib as L
dev = torch.device("cuda:0")
lib = L.load()
model = harness.build_model("molhiv_b512", dev, seed=0).train()
data = harness.to_device(harness.make_batch("molhiv_b512", seed=1), dev)
stamps = torch.zeros(32, dtype=torch.int64, device=dev)
names = ["P0 (gather / rows -> LDS)", "GEMM1 + z1 out", "stats1 issued", "barrier", "totals, XF, a in LDS", "GEMM2 + y2 out", "stats2 issued"]
for agg in (True, False):
    ops.MLP_FWD_FUSED_AGG = agg
    rows = []
    for i in range(12):
        lib.phc_debug_mlp_bwd_stamps(L.ptr(stamps))
        with ops.stat_accumulators(dev), ops.prebuilt_H(ops.phm_pairs(model), ops.dense_weights(model), ops.gate_specs(model),
                                                       encode=ops.atom_encode_spec(model, data)):
            with torch.no_grad():
                pass
            logits = model(data)
        torch.cuda.synchronize()
        lib.phc_debug_mlp_bwd_stamps(None)
        rows.append(stamps.clone().cpu())
    rows = rows[2:]
    for which, off in (("wg0", 0), ("wgLast", 16)):
        med = []
        for j in range(1, 8):
            v = sorted(float(r[off + j] - r[off + j - 1]) / 100.0 for r in rows)
            med.append(v[len(v) // 2])
        if agg:
            sub = [sorted(float(r[off + j] - r[off]) / 100.0 for r in rows)[len(rows) // 2] for j in (8, 9, 10, 12, 13, 14, 15, 11)]
            print("   since entry: prologue %.2f | barrier %.2f | loads A issued %.2f | A arrived %.2f | A summed %.2f | B arrived %.2f | B summed %.2f | stored %.2f" % tuple(sub), flush=True)
        print(f"agg={agg}", which, " | ".join(f"{nm} {m:.2f}" for nm, m in zip(names, med)), f"| total {sum(med):.2f} us")
