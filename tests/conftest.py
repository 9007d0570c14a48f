import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import phc_gnn_amd  # noqa: E402,F401  (HIP runtime flags must be in the environment before the first torch.cuda call)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        # the CPU oracle runs on small tensors: on a many-core GPU host torch's default intra-op pool (one thread per core)
        # is hundreds of times slower than a handful of threads (measured: 91 s instead of 0.5 s per 512-graph step at 256)
        import torch
        torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    except Exception:  # pragma: no cover
        pass


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when collected on a box without a GPU
    and no -m filter was given."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _no_step_block_left_open():
    """A training-mode ``model(data)`` on a ROCm device opens the step's contexts by itself (phc_gnn_amd.ops.auto_step_forward) and the
    backward's end closes them; a test that never calls ``backward`` leaves the block to the next forward.  Tests must not see each
    other's: close whatever is still open."""
    yield
    import sys
    ops = sys.modules.get("phc_gnn_amd.ops")
    if ops is not None:
        ops.close_auto_step(flush=False)
