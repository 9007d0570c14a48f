import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phc_gnn_amd import harness, graphed
dev = torch.device("cuda:0")
wl = "molhiv_b32"
cfg = harness.StepConfig()
gbs = [harness.make_batch(wl, seed=s) for s in (3, 4, 5)]
n_max, e_max = graphed.bucket_shape(gbs)
padded = [graphed.padded_to_device(graphed.pad_batch(g, n_max, e_max), dev) for g in gbs]
m = harness.build_model(wl, dev, seed=0); m.train()
o = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True, capturable=True)
torch.cuda.memory._record_memory_history(context="alloc", stacks="python", max_entries=200000)
orig_exit = torch.cuda.graph.__exit__
MARK = {}
def exit_(self, *a):
    r = orig_exit(self, *a)
    MARK["n_events_at_capture_end"] = len(torch.cuda.memory._snapshot()["device_traces"][0])
    return r
orig_enter = torch.cuda.graph.__enter__
def enter(self):
    r = orig_enter(self)
    MARK["n_events_at_capture_begin"] = len(torch.cuda.memory._snapshot()["device_traces"][0])
    return r
torch.cuda.graph.__enter__ = enter
torch.cuda.graph.__exit__ = exit_
step = graphed.GraphedTrainStep(m, o, cfg, padded[0], warmup=2)
torch.cuda.synchronize()
snap = torch.cuda.memory._snapshot()
segs = [(s["address"], s["address"] + s["total_size"], s.get("segment_pool_id"), s.get("stream")) for s in snap["segments"]]
ev = snap["device_traces"][0]
b, e = MARK["n_events_at_capture_begin"], MARK["n_events_at_capture_end"]
print("events", len(ev), "capture window", b, e, "capture stream", step.stream.cuda_stream)
stats = {}
examples = {}
for x in ev[b:e]:
    if x["action"] != "alloc":
        continue
    owner = [s for s in segs if s[0] <= x["addr"] < s[1]]
    pid = str(owner[0][2]) if owner else "none"
    key = (pid, x["stream"])
    stats[key] = stats.get(key, 0) + 1
    if pid != "(1, 0)" and key not in examples:
        fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in x.get("frames", [])[:8]]
        examples[key] = (x["size"], fr)
for k, v in sorted(stats.items()):
    print("alloc during capture: pool", k[0], "stream", k[1], "count", v)
for k, v in examples.items():
    print("EXAMPLE non-private alloc", k, v)
