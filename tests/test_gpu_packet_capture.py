"""hipGraphLaunch's packet-capture fast path (DEBUG_CLR_GRAPH_PACKET_CAPTURE=1, what bench.py turns on for the single-process run)
with the double-buffered static inputs of GraphedTrainStep: unsynchronised replays == synchronised replays, bit for bit.
The flag is read at HIP initialisation, hence the child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("where", ["on_stream", "default_stream"])
def test_double_buffered_replays_with_packet_capture_do_not_race(where):
    env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE="1")
    out = subprocess.run([sys.executable, os.path.join(HERE, "helpers", "packet_capture_child.py"), where, "60"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "PACKET_CAPTURE_OK" in out.stdout


def test_packet_capture_is_refused_for_data_parallel_steps():
    """A collective queued behind a replay is not protected by double buffering: the constructor says so."""
    code = ("import os; os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE']='1'\n"
            "import sys; sys.path.insert(0, %r)\n"
            "import phc_gnn_amd, torch\n"
            "from phc_gnn_amd import graphed, harness\n"
            "dev = torch.device('cuda:0'); wl = 'molhiv_b32'\n"
            "g = harness.make_batch(wl, seed=0); n, e = graphed.bucket_shape([g])\n"
            "p = graphed.padded_to_device(graphed.pad_batch(g, n, e), dev)\n"
            "m = harness.build_model(wl, dev, seed=0); o = torch.optim.Adam(m.parameters(), fused=True, capturable=True)\n"
            "try:\n"
            "    graphed.GraphedTrainStep(m, o, harness.StepConfig(), p, grad_sync=object())\n"
            "except RuntimeError as ex:\n"
            "    assert 'packet capture' in str(ex); print('REFUSED')\n" % os.path.dirname(HERE))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "REFUSED" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
