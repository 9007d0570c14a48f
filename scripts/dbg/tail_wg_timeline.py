"""Entry / exit clocks of EVERY workgroup of the step tail's first launch inside the replayed training step
(-DPHC_TAIL_STAMPS variant build):
    python -m phc_gnn_amd.build --variant tailstamps -DPHC_TAIL_STAMPS
    PHC_LIB_PATH=scripts/exp/variants/libphc_hip_tailstamps.so python scripts/dbg/tail_wg_timeline.py [workload]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from phc_gnn_amd import graphed, harness, lib as L
from phc_gnn_amd.optim import ClipAdam
wl = sys.argv[1] if len(sys.argv) > 1 else "molhiv_b512"
dev = torch.device("cuda:0")
model = harness.build_model(wl, dev, seed=1).train()
w = harness.WORKLOADS[wl]
cfg = harness.StepConfig(regression=bool(w.get("regression")), optimizer=True, weight_decay=1e-3)
batches = [harness.make_batch(wl, seed=100 + i) for i in range(4)]
n_max, e_max = graphed.bucket_shape(batches)
padded = [graphed.padded_to_device(graphed.pad_batch(b, n_max, e_max), dev) for b in batches]
step = graphed.GraphedTrainStep(model, ClipAdam(model.parameters(), lr=1e-3), cfg, padded[0], warmup=2)
dbg = C.CDLL(L.LIB_PATH)
with step.on_stream():
    for it in range(12):
        step.step(padded[it % 4])
torch.cuda.synchronize()
buf = (C.c_ulonglong * (4096 * 4))()
assert dbg.phc_debug_tail_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4)
live = a[:, 1] > 0
a = a[live]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
fam = (a[:, 2] & 15).astype(int)
fam = np.where(fam >= 2, 2, fam)
job = ((a[:, 2] >> 4) & 255).astype(int)
tasks = ((a[:, 2] >> 12) & 15).astype(int)
hw = a[:, 3] & 0xffffffff
xcc = (a[:, 3] >> 32) & 15
cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (xcc << 7)               # CU_ID | SE_ID | XCC_ID
print(f"{wl}: {len(a)} workgroups, launch span {en.max():.1f} us, distinct CUs {len(set(cu.tolist()))}")
for f, name in ((0, "atom tables"), (1, "bond tables"), (2, "dH")):
    m = fam == f
    if not m.any():
        continue
    d = en[m] - st[m]
    print(f"  {name:12s} n={m.sum():4d}  start {st[m].min():6.1f} .. {st[m].max():6.1f}   end {en[m].min():6.1f} .. {en[m].max():6.1f}   "
          f"duration min {d.min():5.1f} median {np.median(d):5.1f} max {d.max():5.1f}")
m = fam == 2
late = m & (st > 5.0)
print(f"  dH workgroups that start later than 5 us: {late.sum()} (median start {np.median(st[late]) if late.any() else 0:.1f} us)")
for j in sorted(set(job[m].tolist())):
    mj = m & (job == j)
    d = en[mj] - st[mj]
    print(f"    job {j:2d}: n={mj.sum():3d} duration median {np.median(d):5.1f} max {d.max():5.1f}  end max {en[mj].max():5.1f}  multi-task wgs {(tasks[mj] > 1).sum()}")
# slot occupancy: how many workgroups are alive at each 2 us tick, by family
for tck in np.arange(0, en.max() + 2, 4.0):
    alive = (st <= tck) & (en > tck)
    print(f"  t={tck:5.1f} us  alive: atom {(alive & (fam == 0)).sum():3d}  bond {(alive & (fam == 1)).sum():3d}  dH {(alive & (fam == 2)).sum():3d}")
per_cu = {}
for c, s_, e_ in zip(cu.tolist(), st.tolist(), en.tolist()):
    per_cu.setdefault(c, []).append((s_, e_))
cnt = np.array([len(v) for v in per_cu.values()])
print(f"  workgroups per CU: min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()}; CUs with > 2: {(cnt > 2).sum()}")
