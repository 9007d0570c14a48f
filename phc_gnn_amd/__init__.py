"""MI355X (gfx950) engine for the PHM-linear + neighbour-aggregation hot path of phc-gnn."""
import os as _os

# HIP 7.0 / ROCm 7.2: the "graph packet capture" fast path of hipGraphLaunch loses the ordering between a replay
# in flight and work queued behind it as soon as an event is recorded on the launch stream (every RCCL collective
# and every Stream.wait_stream does that) or the launch goes to the null stream: the next batch's copy then
# overwrites the static buffers of the running step (scripts/dbg/debug_pipelined.py reproduces the memory fault).
# The regular enqueue path is correct and, for this step, equally fast (measured 1.43 ms/step either way).
# The flag is read when the HIP runtime initialises, i.e. at the first torch.cuda call — import this package first.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def check_indices(device=None) -> None:
    """Raise IndexError if an engine kernel met an out-of-range node id / feature index since the last call (the value
    was read as 0, never dereferenced).  One 4-byte device read; see include/phc_hip.h ``phc_set_error_word``."""
    from .lib import check_indices as _check
    _check(device)
