import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the tests put graphs of every size through the wide kernels (production keeps them for batches that fill the GPU)
from g_adaptivity_amd import graph as _graph_mod   # noqa: E402
_graph_mod.WIDE_MIN_NODES = 0


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is torch code: on a many-core host torch's default thread count oversubscribes its index_add_ / gather
    # loops (measured on the 256-core GPU box: 6 meshes/s at 128 threads, 27 at 16).  Cap it unless the caller chose a count.
    if 'OMP_NUM_THREADS' not in os.environ:
        import torch
        if torch.get_num_threads() > 16:
            torch.set_num_threads(16)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
