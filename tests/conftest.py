import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this process")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _default_noise_key():
    """Every test starts from the default Philox key and offset (settings.set_seed(0)): a test that re-keys the on-device noise
    stream must not change the draws -- and with them the statistical assertions -- of whichever test runs next."""
    from dgps_with_iwvi_amd import settings
    settings.set_seed(0)
    yield
    settings.set_seed(0)
