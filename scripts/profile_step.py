"""Host+device profile of the training step with torch.profiler (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from phc_gnn_amd import harness

wl = sys.argv[1] if len(sys.argv) > 1 else "molhiv_b512"
dev = torch.device("cuda:0")
model = harness.build_model(wl, dev)
model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
cfg = harness.StepConfig()
batches = [harness.to_device(harness.make_batch(wl, seed=i), dev) for i in range(4)]
for b in batches:
    b.all_labeled = True
for i in range(5):
    harness.train_step(model, batches[i % 4], opt, cfg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    harness.train_step(model, batches[i % 4], opt, cfg)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 100)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(5):
        harness.train_step(model, batches[i % 4], opt, cfg)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
