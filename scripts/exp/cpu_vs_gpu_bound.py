"""Is the replayed step GPU-bound or bound by the CPU's graph-launch rate?  Times issuing K steps (no sync) vs total."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phc_gnn_amd  # noqa
import torch
from phc_gnn_amd import harness, graphed
dev = torch.device("cuda:0")
wl = "molhiv_b512"
m = harness.build_model(wl, dev, seed=0); m.train()
from phc_gnn_amd.optim import ClipAdam
o = ClipAdam(m.parameters(), lr=1e-3)
hb = [harness.make_batch(wl, seed=i) for i in range(8)]
n_max, e_max = graphed.bucket_shape(hb)
padded = [graphed.padded_to_device(graphed.pad_batch(g, n_max, e_max), dev) for g in hb]
g = graphed.GraphedTrainStep(m, o, harness.StepConfig(), padded[0])
K = 300
with g.on_stream():
    for i in range(20): g.step(padded[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): g.step(padded[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"issue: {(t1-t0)/K*1e6:.1f} us/step   total: {(t2-t0)/K*1e6:.1f} us/step   (CPU-bound if issue ~= total)")
# replay only, per-step sync: pure GPU time of one replay incl. launch latency
with g.on_stream():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100):
        g.step(None); torch.cuda.synchronize()
    t1 = time.perf_counter()
print(f"replay + sync each step: {(t1-t0)/100*1e6:.1f} us")
