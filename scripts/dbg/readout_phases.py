"""Phase clocks of the one-launch read-out (phc_readout_fused_fwd + phc_debug_readout_stamps) inside a real training step of a bench
workload: per workgroup (= graph) entry -> rows found -> norm constants -> tiles done -> end, against the graph's size.
    python scripts/dbg/readout_phases.py [workload]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import phc_gnn_amd  # noqa
import torch
from phc_gnn_amd import harness, lib as L

DEV = "cuda:0"
wl = sys.argv[1] if len(sys.argv) > 1 else "molhiv_b512"
model = harness.build_model(wl, DEV, seed=3).train()
gb = harness.make_batch(wl, seed=11)
cfg = harness.StepConfig(regression=bool(harness.WORKLOADS[wl].get("regression")), optimizer=False, weight_decay=1e-3)
data = harness.to_device(gb, DEV)
for _ in range(3):
    harness.train_step(model, data, None, cfg)
torch.cuda.synchronize()
B = int(data.num_graphs)
stamps = torch.zeros(8 * (B + 64), dtype=torch.int64, device=DEV)
lib = L.load()
L.check(lib.phc_debug_readout_stamps(L.ptr(stamps)), "stamps")
harness.train_step(model, data, None, cfg)
torch.cuda.synchronize()
L.check(lib.phc_debug_readout_stamps(None), "stamps off")
w = stamps.cpu().view(-1, 8)[:B].double()
w = w[w[:, 0] != 0]
t0 = w[:, 0].min()
us = lambda c: (w[:, c] - w[:, 0]) * 0.01
rows = w[:, 5]
print(f"{wl}: {len(w)} graphs, rows per graph mean {rows.mean():.1f} max {int(rows.max())}")
print(f"search: first level resolved {((w[:, 6] - w[:, 0]) * 0.01).median():.2f} us after entry, second {((w[:, 7] - w[:, 0]) * 0.01).median():.2f}")
print(f"entry spread {((w[:, 0] - t0) * 0.01).max():.2f} us | rows found {us(1).median():.2f} | norm constants {us(2).median():.2f} | tiles done med {us(3).median():.2f} "
      f"max {us(3).max():.2f} | end med {us(4).median():.2f} max {us(4).max():.2f} | last end since first entry {((w[:, 4] - t0) * 0.01).max():.2f}")
for lo, hi in ((0, 16), (17, 32), (33, 48), (49, 96), (97, 1000)):
    sel = (rows >= lo) & (rows <= hi)
    if sel.any():
        print(f"  graphs of {lo}-{hi} rows: {int(sel.sum())}, tiles phase med {(us(3) - us(2))[sel].median():.2f} us, end med {us(4)[sel].median():.2f}")
