import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phc_gnn_amd import graphed, harness
from phc_gnn_amd.optim import ClipAdam
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
DEV = "cuda:0"
wl = "molhiv_b32"
gbs = [harness.make_batch(wl, seed=s) for s in (11, 12, 13, 14)]
n_max, e_max = graphed.bucket_shape(gbs)
padded = [graphed.padded_to_device(graphed.pad_batch(g, n_max, e_max), DEV) for g in gbs]
def fresh(seed):
    m = harness.build_model(wl, DEV, seed=seed); m.train()
    return m
m_t = fresh(3); o_t = torch.optim.Adam(m_t.parameters(), lr=1e-3, fused=True, capturable=True)
m_e = fresh(3); o_e = ClipAdam(m_e.parameters(), lr=1e-3)
base = {k: v.clone() for k, v in m_t.state_dict().items()}
s_t = graphed.GraphedTrainStep(m_t, o_t, harness.StepConfig(), padded[0], warmup=1)
s_e = graphed.GraphedTrainStep(m_e, o_e, harness.StepConfig(), padded[0], warmup=1)
for m in (m_t, m_e):
    m.load_state_dict(base)
for st in o_t.state.values():
    [v.zero_() for v in st.values() if torch.is_tensor(v)]
o_e.step_count.zero_(); o_e.exp_avg_flat.zero_(); o_e.exp_avg_sq_flat.zero_()
for i, p in enumerate(padded):
    le, lt = s_e.step(p), s_t.step(p)
    torch.cuda.synchronize()
    rows = []
    for (k, a), b in zip(m_e.state_dict().items(), m_t.state_dict().values()):
        if not a.is_floating_point(): continue
        d = float((a - b).abs().max()); e = d / max(float(b.abs().max()), 1e-30)
        rows.append((e, d, k))
    rows.sort(reverse=True)
    print(f"step {i} loss {float(le):.7f} {float(lt):.7f}; worst:", [(f"{e:.1e}", f"{d:.1e}", k) for e, d, k in rows[:4]],
          " pooling:", [(f"{e:.1e}", k) for e, d, k in rows if k.startswith("pooling")][:6])
