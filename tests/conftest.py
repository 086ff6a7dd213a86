import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    # whole-suite A/B runs: E2EFT_TEST_PERSISTENT_GRID=8 sends every eligible SMALL problem of the tests through the persistent kernel,
    # E2EFT_TEST_PERSISTENT=0 keeps everything on igemm2 (the library itself never reads the environment: e2eft_set_option)
    from diffusion_e2e_ft_amd import _lib
    if "E2EFT_TEST_PERSISTENT_GRID" in os.environ:
        _lib.set_option(_lib.OPT_PERSISTENT_GRID, int(os.environ["E2EFT_TEST_PERSISTENT_GRID"]))
    if "E2EFT_TEST_PERSISTENT" in os.environ:
        _lib.set_option(_lib.OPT_PERSISTENT, int(os.environ["E2EFT_TEST_PERSISTENT"]))
    return torch.device("cuda:0")
