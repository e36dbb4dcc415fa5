import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from g_adaptivity_amd import graph as _graph_mod   # noqa: E402

# Kernel choice.  graph.WIDE_MIN_NODES decides which forward kernel a mesh-ordered hidden-64 batch gets:
# production keeps the wide kernel for batches that fill the GPU (>= 24 576 nodes) and runs the tiled, LDS-windowed forward
# below that.  EVERY -m gpu test runs under both settings - 'wide-any-size' (limit 0: graphs of every size go through the wide
# kernels, which is what exercises them on test-sized inputs) and 'production-dispatch' (the value users get) - unless it is
# marked `one_dispatch` (tests that never build a graph, or that choose the limit themselves per case).
PRODUCTION_WIDE_MIN_NODES = _graph_mod.WIDE_MIN_NODES
DISPATCHES = {'wide-any-size': 0, 'production-dispatch': PRODUCTION_WIDE_MIN_NODES}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "one_dispatch: independent of graph.WIDE_MIN_NODES (or sets it per case): not run once per kernel choice")
    # The CPU oracle is torch code: on a many-core host torch's default thread count oversubscribes its index_add_ / gather
    # loops (measured on the 256-core GPU box: 6 meshes/s at 128 threads, 27 at 16).  Cap it unless the caller chose a count.
    if 'OMP_NUM_THREADS' not in os.environ:
        import torch
        if torch.get_num_threads() > 16:
            torch.set_num_threads(16)


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker('gpu') is not None and metafunc.definition.get_closest_marker('one_dispatch') is None:
        metafunc.parametrize('_kernel_dispatch', list(DISPATCHES), indirect=True)


@pytest.fixture(autouse=True)
def _kernel_dispatch(request):
    """Sets graph.WIDE_MIN_NODES for one test: parametrised over DISPATCHES for -m gpu tests (pytest_generate_tests), the
    wide-any-size setting for everything else (CPU tests, `one_dispatch` GPU tests)."""
    name = getattr(request, 'param', 'wide-any-size')
    keep = _graph_mod.WIDE_MIN_NODES
    _graph_mod.WIDE_MIN_NODES = DISPATCHES[name]
    try:
        yield name
    finally:
        _graph_mod.WIDE_MIN_NODES = keep


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
