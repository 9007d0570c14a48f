"""Host-side profile of the reference's unchanged loop body on the mirror model (bench.py --mode refloop is Python-bound):
PYTHONPATH=. python scripts/dbg/refloop_profile.py [steps]   -> cProfile, cumulative and own time"""
import cProfile
import pstats
import sys
import time
import torch
import torch.nn.functional as F
from phc_gnn_amd import harness
from phc.hypercomplex.regularization import phm_weight_regularization

dev = torch.device("cuda:0")
model = harness.build_model("molhiv_b512", dev, seed=0).train()
batches = [harness.to_device(harness.make_batch("molhiv_b512", seed=s), dev) for s in range(4)]
opt = torch.optim.Adam(model.parameters(), weight_decay=0.0)
tot = [0.0]


def body(data):
    mask = ~torch.isnan(data.y)
    opt.zero_grad()
    logits = model(data)
    y = data.y.to(torch.float)
    loss = F.binary_cross_entropy_with_logits(input=logits[mask], target=y[mask])
    reg_loss = torch.tensor(0.0, device=dev)
    reg_loss += 1e-3 * 0.1 * phm_weight_regularization(model, device=dev, p=2)
    loss += reg_loss.squeeze()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=2.0, norm_type=2)
    opt.step()
    tot[0] += loss.item() * data.num_graphs


for i in range(10):
    body(batches[i % 4])
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
t0 = time.perf_counter()
for i in range(n):
    body(batches[i % 4])
torch.cuda.synchronize()
print(f"plain: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    body(batches[i % 4])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(35)
