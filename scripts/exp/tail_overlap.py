"""Do the step's tail kernels (PHM parameter gradients vs table gradients) overlap when they run concurrently?
Hypothesis test for merging them into one heterogeneous launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phc_gnn_amd  # noqa
import torch
from phc_gnn_amd import harness, ops
dev = torch.device("cuda:0")
wl = "molhiv_b512"
m = harness.build_model(wl, dev, seed=0); m.train()
data = harness.to_device(harness.make_batch(wl, seed=0), dev)
# capture the jobs of one backward pass
jobs = {}
orig_p, orig_t, orig_a = ops._flush_param_grads, ops._flush_table_grads, ops._flush_atom_grads
ops.STEP_TAIL_MERGED = False                       # separate launches: each family can be timed on its own
ops._flush_param_grads = lambda js: jobs.__setitem__("p", js)
ops._flush_table_grads = lambda js: jobs.__setitem__("t", js)
ops._flush_atom_grads = lambda aj: jobs.__setitem__("a", aj)
with ops.prebuilt_H(ops.phm_pairs(m), ops.dense_weights(m)), ops.deferred_param_grads():
    loss = harness.loss_fn(m(data), data)
    loss.backward()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run(which, reps=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if which in ("p", "both"):
            with torch.cuda.stream(sa): orig_p(jobs["p"])
        if which in ("t", "both"):
            with torch.cuda.stream(sb): orig_t(jobs["t"])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for w in ("p", "t", "both", "p", "t", "both"):
    print(w, f"{run(w):.1f} us")

# ---- GPU time via captured graphs: sequential vs forked branches
def capture(kind):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        orig_p(jobs["p"]); orig_t(jobs["t"])        # warm-up allocations
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            if kind == "seq":
                orig_p(jobs["p"]); orig_t(jobs["t"])
            elif kind == "p":
                orig_p(jobs["p"])
            elif kind == "t":
                orig_t(jobs["t"])
            else:
                side = torch.cuda.Stream()
                side.wait_stream(s)
                with torch.cuda.stream(side):
                    orig_t(jobs["t"])
                orig_p(jobs["p"])
                s.wait_stream(side)
    return g, s
for kind in (os.environ.get("TAIL_KINDS", "p,t,seq,fork").split(",")):
    g, s = capture(kind)
    rs = torch.cuda.Stream()
    with torch.cuda.stream(rs):
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): g.replay()
        torch.cuda.synchronize()
    print(kind, f"{(time.perf_counter() - t0) / 300 * 1e6:.1f} us per replay")
