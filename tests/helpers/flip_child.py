"""Child process of tests/test_gpu_dispatch_surface.py::test_every_library_switch_holds_the_oracle_in_a_fresh_process: the library's
environment switches are read once per process, so every flip gets a process of its own.
    python -m tests.helpers.flip_child NAME=VALUE workload [workload ...]"""
import os
import sys


def main():
    tag = sys.argv[1]
    name, value = tag.split("=", 1)
    assert os.environ.get(name) == value, (name, os.environ.get(name))
    import torch
    assert torch.cuda.is_available()
    from tests.test_gpu_dispatch_surface import check_step_against_oracle
    from phc_gnn_amd import lib
    for wl in sys.argv[2:]:
        check_step_against_oracle(wl, "none", tag=tag)
    lib.check_indices(torch.device("cuda:0"))
    print("FLIP-OK", tag, flush=True)


if __name__ == "__main__":
    main()
