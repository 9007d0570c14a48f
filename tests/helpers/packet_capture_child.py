"""Child process of tests/test_gpu_packet_capture.py: runs with DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 (the flag is read when the
HIP runtime initialises, so it cannot be flipped inside the pytest process) and checks that GraphedTrainStep's double-buffered
replay loop gives bit-identical results with and without host synchronisation between steps."""
import contextlib
import os
import sys

assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import phc_gnn_amd  # noqa: E402,F401
import torch  # noqa: E402
from phc_gnn_amd import graphed, harness  # noqa: E402


def main(where: str, steps: int) -> None:
    dev = torch.device("cuda:0")
    wl = "molhiv_b512"
    gbs = [harness.make_batch(wl, seed=s) for s in range(5)]          # odd count: every batch visits both buffer sets
    n_max, e_max = graphed.bucket_shape(gbs)
    padded = [graphed.padded_to_device(graphed.pad_batch(g, n_max, e_max), dev) for g in gbs]

    def fresh():
        m = harness.build_model(wl, dev, seed=2)
        m.train()
        return m, torch.optim.Adam(m.parameters(), lr=1e-3, fused=True, capturable=True)

    m, o = fresh()
    step = graphed.GraphedTrainStep(m, o, harness.StepConfig(), padded[0], warmup=1)
    assert step.double_buffer and len(step.sets) == 2, "packet capture on => two static input sets"
    base = {k: v.clone() for k, v in fresh()[0].state_dict().items()}
    results = []
    for synced in (True, False):
        m.load_state_dict(base)
        for st in o.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        torch.cuda.synchronize()
        losses = torch.zeros(steps, device=dev)
        with (step.on_stream() if where == "on_stream" else contextlib.nullcontext()):
            for i in range(steps):
                loss = step.step(padded[i % len(padded)])
                losses[i].copy_(loss.detach().reshape(()), non_blocking=True)      # stream-ordered read of this step's loss
                if synced:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
        results.append((losses.cpu(), {k: v.clone() for k, v in m.state_dict().items()}))
    assert torch.equal(results[0][0], results[1][0]), (results[0][0], results[1][0])
    bad = [k for k in base if not torch.equal(results[0][1][k], results[1][1][k])]
    assert not bad, bad
    phc_gnn_amd.check_indices()
    print("PACKET_CAPTURE_OK", where, steps, float(results[0][0][-1]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
