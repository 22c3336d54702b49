import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the host thread pool inside the container's CPU quota (the oracle legs' OpenMP teams included): a pool sized by the
    # machine gets the whole session frozen by the kernel's bandwidth control again and again (rampvo_amd/hostenv.py)
    import warnings
    from rampvo_amd import hostenv
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hostenv.fit_host_threads()


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _quiesce_gpu_between_tests(request):
    """GPU tests build and drop trackers (hipGraphs, helper threads, side streams) in quick succession: drain the device
    and collect the dead objects at a quiescent point, so that no graph is destroyed while another one is captured"""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
