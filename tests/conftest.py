import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_python.npz"))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from exllamav3_amd import ext
    ext.init(0)
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _default_kernel_selection(request):
    """GPU tests start from the library's default kernel selection (tests that pin a generation / variant do so explicitly)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        from exllamav3_amd import ext
        ext.set_gemm3_min_rows(5)
        ext.set_gemv_gen4(True)
    yield
