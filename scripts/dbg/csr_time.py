import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phc_gnn_amd, torch
from phc_gnn_amd import lib as L
from phc_gnn_amd.graph import EdgeStructure
N, E = 10_000_000, 50_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
ei = torch.randint(0, N, (2, E), generator=g, device=dev)
for it in range(3):
    st = EdgeStructure(ei, N); st.csr; st.csc
torch.cuda.synchronize()
L.prof_reset(); L.prof_enable(True)
for it in range(3):
    st = EdgeStructure(ei, N); st.csr; st.csc
torch.cuda.synchronize()
L.prof_enable(False)
for k, v in L.prof_read().items():
    print(k, round(v["total_ms"] / 3, 3), "ms per build pair", v["launches"] // 3, "launches")
