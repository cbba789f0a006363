import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a test never starts a multi-minute build of a kernel-instance library (engine/instances.py): what __graft_entry__.build() built is
    # what runs; a model without a prebuilt exact library takes the generic instances
    os.environ.setdefault("PMT_JIT", "0")
    # the oracle is PyTorch on the CPU: on a many-core host its default intra-op pool (one thread per hardware thread: 256 on the
    # MI355X hosts, where a job owns 16) is oversubscribed several times over and every small op becomes a thread rendezvous
    import torch
    torch.set_num_threads(min(torch.get_num_threads(), 16))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
